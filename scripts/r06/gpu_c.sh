#!/bin/bash
# round 6, call C: full GPU suite; FinalLayer-as-tail A/B (option mlp_tail); k_flash_proj8 without its two round-6 changes (NOPRIO, NOEARLY);
# 64-row form experiments on ATLAS (HWPRIO, EARLY64)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06c; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -12
bash scripts/micro/flash_variants.sh NOPRIO NOEARLY HWPRIO EARLY64 > $O/build.log 2>&1; tail -2 $O/build.log
run_k() { echo "== $1 $2 $3" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 $3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline $3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
C2=tetrapeptide_fwdsim_crop4_T1000_B16; AT=atlas_crop256_T250_B1
for rep in 1 2; do
  unset MDGEN_AMD_LIB
  run_k product $C2; run_k product $C2 mlp_tail=0; run_k product $AT; run_k product $AT mlp_tail=0
  for v in NOPRIO NOEARLY; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_k $v $C2; done
  for v in HWPRIO EARLY64; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_k $v $AT; done
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB
  run_b product $C2; run_b product $C2 "--option mlp_tail=0"; run_b product $AT; run_b product $AT "--option mlp_tail=0"
  for v in NOPRIO NOEARLY; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_b $v $C2; done
  for v in HWPRIO EARLY64; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_b $v $AT; done
done
