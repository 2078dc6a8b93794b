#!/bin/bash
# round 6, call G: ATLAS with the 128-query fused attention form on the temporal axis (flash_proj_form=8) and with per-head-group priority
# turns between a CU's two workgroups in the 64-query form (BPRIO build); tests: TPS registry case, hand-over stress, what failed before
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06g; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s -k "registry or stress or headline_regime or rccl" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-250 | tail -8
bash scripts/micro/flash_variants.sh BPRIO > $O/build.log 2>&1; tail -1 $O/build.log
run_k() { echo "== $1 $2 $3" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 $3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline $3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
AT=atlas_crop256_T250_B1
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $AT; run_k product $AT flash_proj_form=8
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_BPRIO.so; run_k BPRIO $AT
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $AT; run_b product $AT "--option flash_proj_form=8"
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_BPRIO.so; run_b BPRIO $AT
done
