#!/bin/bash
# round 5, call O: k_flash_proj with three K / V tile slots and late bias loads (product) against the two-slot form (SHALLOW build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05o; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_proj or forward_cfg4_full or full_size_properties or forward_vs_oracle_shapes or attention_fixed_anchor" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -6
bash scripts/micro/flash_variants.sh SHALLOW > $O/build.log 2>&1; tail -1 $O/build.log
for rep in 1 2; do for v in product SHALLOW; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  for a in "tetrapeptide_fwdsim_crop4_T1000_B16 flash_proj_form=4" "tetrapeptide_fwdsim_crop4_T1000_B16 flash_proj_form=8" "atlas_crop256_T250_B1"; do
    set -- $a
    echo "== $v $a" | tee -a $O/kbench.txt
    timeout 300 python scripts/kbench.py $1 3 $2 2>&1 | grep "flash_proj" | tee -a $O/kbench.txt
  done
done; done
for v in product SHALLOW product SHALLOW; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  for a in "tetrapeptide_fwdsim_crop4_T1000_B16 --option flash_proj_form=4" "tetrapeptide_fwdsim_crop4_T1000_B16 --option flash_proj_form=8" "atlas_crop256_T250_B1"; do
    timeout 300 python bench.py --workload $a --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v $a', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done
