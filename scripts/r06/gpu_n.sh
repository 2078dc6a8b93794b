#!/bin/bash
# round 6, call N: phase stamps of k_ln_qkv_attn4<true> held in SGPRs (a build that spills like the product: 11 registers against 10)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06n; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS > $O/build.log 2>&1; tail -1 $O/build.log
for wl in tetrapeptide_fwdsim_crop4_T1000_B1 tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B16; do
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/attn4_stamps.py $wl 2>&1 | grep -v amdgpu | tail -12 | tee -a $O/out.txt
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/kbench.py $wl 3 2>&1 | grep "attn_L" | tee -a $O/out.txt
done
