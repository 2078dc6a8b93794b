#!/bin/bash
# round 6, call U: weight-ring depth of the small launches' one-tile-per-wave GEMMs (32-row forms, k_mlp8's out-projection phase): 4 (product) / 8 / 12
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06u; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=SMALL bash scripts/micro/flash_variants.sh PF8 PF12 > $O/build.log 2>&1; tail -1 $O/build.log
B1=tetrapeptide_fwdsim_crop4_T1000_B1; TP=tetrapeptide_tps_crop4_T100_B32
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $B1; run_k product $TP
  for v in PF8 PF12; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_k $v $B1; run_k $v $TP; done
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $B1; run_b product $TP
  for v in PF8 PF12; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run_b $v $B1; run_b $v $TP; done
done
