#!/bin/bash
# round 6, call E: the embedding tail's products as bf16 pairs (option embed_split 1) against the fp32 MFMA form (0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06e; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -s -k "fold or headline or registry or B16 or multi_block or inference_headline" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit\|embedding" | cut -c1-250 | tail -12
run_k() { echo "== $1 $2 $3" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 $3 2>&1 | grep -v parity | grep -v amdgpu | grep "per NFE\|embed\|final" | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline $3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
C2=tetrapeptide_fwdsim_crop4_T1000_B16; AT=atlas_crop256_T250_B1
for rep in 1 2; do run_k product $C2; run_k product $C2 embed_split=0; run_k product $AT; run_k product $AT embed_split=0; done
for rep in 1 2 3; do
  run_b product $C2; run_b product $C2 "--option embed_split=0"; run_b product $C2 "--option mlp_tail=1"
  run_b product $AT; run_b product $AT "--option embed_split=0"
done
