#!/bin/bash
# round 6, call R: k_ln_qkv8<true, true>: the split q, k | v kernel on 32-position workgroups (B = 1 at T 1000: 256 workgroups): parity, timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06r; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2000 python -m pytest tests -q -m gpu -s -x -k "small_ or split or registry or forward_vs or golden or fwd or inference or cfg1 or tps or ipa_table or headline or row_owner or stress or attention or flash" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-300 | tail -12
B1=tetrapeptide_fwdsim_crop4_T1000_B1
for rep in 1 2; do timeout 300 python scripts/kbench.py $B1 3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; done
for rep in 1 2 3; do timeout 300 python bench.py --workload $B1 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B1', d['value'], d['ms_per_step'])" | tee -a $O/bench.txt; done
