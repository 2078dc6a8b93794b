#!/bin/bash
# round 6, call B: the new parity tests in full; k_flash_proj8 experiment builds (TRUNC, PRIO1, PRIO2, EARLY) against the product
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06b; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "fold or headline or B16 or flash_proj or row_owner or small_launches or native" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -8
bash scripts/micro/flash_variants.sh TRUNC PRIO1 PRIO2 EARLY > $O/build.log 2>&1; tail -2 $O/build.log
# accuracy of the truncating pack: the five fused-kernel shapes and the headline inference golden, product vs TRUNC
for v in product TRUNC; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "flash_proj_kernel or inference_headline_regime or headline_kernel_mix" > $O/acc_$v.log 2>&1
  echo "== $v" >> $O/acc.txt; grep -E "fused 128'|S=49|S=10|one Euler|passed|failed" $O/acc_$v.log | grep -v amdgpu | cut -c1-300 >> $O/acc.txt
done
for rep in 1 2; do for v in product TRUNC PRIO1 PRIO2 EARLY; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 3 2>&1 | grep "flash_proj\|per NFE" | tee -a $O/kbench.txt
done; done
for rep in 1 2; do for v in product TRUNC PRIO1 PRIO2 EARLY; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done; done
