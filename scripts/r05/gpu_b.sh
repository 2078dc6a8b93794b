#!/bin/bash
# round 5, call B: the remaining new parity tests; k_flash_proj phase stamps (STAMPS experiment build, built on the box) at both
# register budgets; the same kernels without their K / V loads (NOLOAD: timing only) to see what the load latency costs at 2 waves/SIMD
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "flash_proj or panel_kernels_257 or two_stream_views" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|vs separate\|exit\|B 8\|B 7\|two 313" | tail -30
bash scripts/micro/flash_variants.sh STAMPS NOLOAD > $O/build.log 2>&1; tail -2 $O/build.log
for o in "flash_proj_occ=2" "flash_proj_occ=3"; do
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/fproj_stamps.py tetrapeptide_fwdsim_crop4_T1000_B16 $o 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a $O/stamps.txt
done
MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/fproj_stamps.py atlas_crop256_T250_B1 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a $O/stamps.txt
MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/micro/flash_stamps.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/stamps.txt
for o in "flash_proj=0" "flash_proj=1 flash_proj_occ=2" "flash_proj=1 flash_proj_occ=3"; do
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_NOLOAD.so timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 3 $o 2>&1 | grep "S=3\|flash" | tee -a $O/noload.txt
done
