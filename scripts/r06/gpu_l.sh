#!/bin/bash
# round 6, call L: where a small launch of the L = 4 sub-layer kernel spends its 41 us (phase stamps at B = 1 and at cfg-3's shard, four- and eight-wave forms)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06l; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS STAMPS4 > $O/build.log 2>&1; tail -1 $O/build.log
for wl in tetrapeptide_fwdsim_crop4_T1000_B1 tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B16; do
for v in STAMPS STAMPS4; do
  echo "== $v $wl" | tee -a $O/out.txt
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so timeout 300 python scripts/r05/attn4_stamps.py $wl 2>&1 | grep -v amdgpu | tail -12 | tee -a $O/out.txt
done; done
