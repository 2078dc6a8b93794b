#!/bin/bash
# round 5, call P/R: k_flash_proj12 (twelve waves per workgroup = three waves per SIMD) against forms 4 and 8
# (call P measured a six-wave form this way; call Q showed it ran one workgroup per CU -- scripts/micro/occ_probe.hip)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05r; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_proj" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -6
for rep in 1 2; do
  for a in "tetrapeptide_fwdsim_crop4_T1000_B16 flash_proj_form=4" "tetrapeptide_fwdsim_crop4_T1000_B16 flash_proj_form=12" "tetrapeptide_fwdsim_crop4_T1000_B16 flash_proj_form=8" "atlas_crop256_T250_B1 flash_proj_form=12" "atlas_crop256_T250_B1"; do
    set -- $a
    echo "== $a" | tee -a $O/kbench.txt
    timeout 300 python scripts/kbench.py $1 3 $2 2>&1 | grep "flash_proj" | tee -a $O/kbench.txt
  done
done
for rep in 1 2; do
  for a in "tetrapeptide_fwdsim_crop4_T1000_B16 --option flash_proj_form=12" "tetrapeptide_fwdsim_crop4_T1000_B16" "atlas_crop256_T250_B1 --option flash_proj_form=12" "atlas_crop256_T250_B1"; do
    timeout 300 python bench.py --workload $a --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done
bash scripts/micro/flash_variants.sh STAMPS > $O/build.log 2>&1; tail -1 $O/build.log
export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so
timeout 300 python scripts/r05/fproj12_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps12.txt
