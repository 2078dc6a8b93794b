#!/bin/bash
# round 6, call A: gate-folded k_mlp_rows (option mlp_fold) -- new parity tests, per-kernel timings and end-to-end A/B against mlp_fold=0
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06a; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "fold or headline or B16 or flash_proj or row_owner or small_launches or native" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -8
for rep in 1 2; do for f in 0 1; do
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
    echo "== mlp_fold=$f $wl" | tee -a $O/kbench.txt
    timeout 300 python scripts/kbench.py $wl 3 mlp_fold=$f 2>&1 | grep -v parity | head -8 | tee -a $O/kbench.txt
  done
done; done
for rep in 1 2; do for f in 0 1; do
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
    timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --option mlp_fold=$f 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mlp_fold=$f $wl', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done; done
