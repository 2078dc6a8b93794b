#!/bin/bash
# round 5, call S: small launches split a panel over workgroups (k_mlp8<., 3>, k_ln_qkv8<true>; option small_split)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05s; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "small_launches_split or small_split_in or row_owner_mlp_paths or abi" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit\|split vs" | tail -14
for rep in 1 2; do
  for a in "tetrapeptide_fwdsim_crop4_T1000_B1 small_split=0" "tetrapeptide_fwdsim_crop4_T1000_B1 small_split=1"; do
    set -- $a
    echo "== $a" | tee -a $O/kbench.txt
    timeout 300 python scripts/kbench.py $1 3 $2 2>&1 | grep -v "amdgpu.ids\|parity" | head -14 | tee -a $O/kbench.txt
  done
done
for rep in 1 2; do
  for a in "tetrapeptide_fwdsim_crop4_T1000_B1 --option small_split=0" "tetrapeptide_fwdsim_crop4_T1000_B1" "tetrapeptide_tps_crop4_T100_B32 --option small_split=0" "tetrapeptide_tps_crop4_T100_B32"; do
    timeout 300 python bench.py --workload $a --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done
