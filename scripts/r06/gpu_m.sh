#!/bin/bash
# round 6, call M: the eight-wave k_ln_qkv_attn4 at the headline (EIGHT build: every launch, one workgroup per CU) against the product
# (two four-wave workgroups per CU), and with six k-steps of weights in flight at B = 1 / the shard (DEEP build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06m; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=ATTN4 bash scripts/micro/flash_variants.sh EIGHT DEEP > $O/build.log 2>&1; tail -1 $O/build.log
C2=tetrapeptide_fwdsim_crop4_T1000_B16; B1=tetrapeptide_fwdsim_crop4_T1000_B1; TP=tetrapeptide_tps_crop4_T100_B32
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 2>&1 | grep -v parity | grep -v amdgpu | head -6 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $C2; run_k product $B1; run_k product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_EIGHT.so; run_k EIGHT $C2
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_DEEP.so; run_k DEEP $B1; run_k DEEP $TP
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $C2; run_b product $B1; run_b product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_EIGHT.so; run_b EIGHT $C2
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_DEEP.so; run_b DEEP $B1; run_b DEEP $TP
done
