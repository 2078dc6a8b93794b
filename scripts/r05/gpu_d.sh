#!/bin/bash
# round 5, call D: k_flash_proj -- per-job phase stamps (which part of the early jobs is slow), residual epilogue with all rows up front
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05d; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_proj" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -6
bash scripts/micro/flash_variants.sh STAMPS > $O/build.log 2>&1; tail -1 $O/build.log
for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
for o in "flash_proj_epilogue=0" "flash_proj_epilogue=1"; do
  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/fproj_stamps.py $wl $o 2>&1 | grep -v amdgpu.ids | tail -9 | tee -a $O/stamps.txt
done; done
for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
  for o in "flash_proj_epilogue=0" "flash_proj_epilogue=1"; do
    timeout 300 python scripts/kbench.py $wl 3 $o 2>&1 | grep "S=3\|flash" | tee -a $O/kbench.txt
  done
done
for o in "flash_proj_epilogue=0" "flash_proj_epilogue=1" ; do
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
    timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', '$o', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done
