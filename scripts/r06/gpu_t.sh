#!/bin/bash
# round 6, call T: phase stamps of k_mlp8 at B = 1 (split form) and at cfg-3's shard (unsplit)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06t; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=MLP8 bash scripts/micro/flash_variants.sh STAMPS > $O/build.log 2>&1; tail -1 $O/build.log
for wl in tetrapeptide_fwdsim_crop4_T1000_B1 tetrapeptide_tps_crop4_T100_B32; do
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r06/mlp8_stamps.py $wl 2>&1 | grep -v amdgpu | tail -16 | tee -a $O/out.txt
done
