#!/bin/bash
# round 6, call I: the folded row-owner MLP with f16 hidden activations and the GELU on packed-f16 VALU ops (mlp_fold 2 against 1): parity, per-launch and end-to-end A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06i; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -s -k "gate_fold or headline_kernel_mix" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit\|vs unfolded\|fold16" | cut -c1-330 | tail -30
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $1 3 $2 2>&1 | grep -v parity | grep -v amdgpu | head -8 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
C2=tetrapeptide_fwdsim_crop4_T1000_B16; AT=atlas_crop256_T250_B1
for rep in 1 2; do
  run_k $C2 mlp_fold=1; run_k $C2 mlp_fold=2; run_k $AT mlp_fold=1; run_k $AT mlp_fold=2
done
for rep in 1 2 3; do
  run_b $C2 "--option mlp_fold=1"; run_b $C2 "--option mlp_fold=2"
  run_b $AT "--option mlp_fold=1"; run_b $AT "--option mlp_fold=2"
done
