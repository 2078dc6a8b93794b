#!/bin/bash
# round 5, call K: the loop's first K / V tiles requested by flash_prefetch: parity, A/B against the NOFIRSTPF experiment build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05k; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_proj or attention_fixed_anchor or forward_vs_oracle_shapes or forward_vs_reference_golden" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -6
bash scripts/micro/flash_variants.sh NOFIRSTPF > $O/build.log 2>&1; tail -1 $O/build.log
for rep in 1 2; do for v in product NOFIRSTPF; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1 tetrapeptide_fwdsim_crop4_T1000_B1; do
    echo "== $v $wl" | tee -a $O/kbench.txt
    timeout 300 python scripts/kbench.py $wl 3 2>&1 | grep "flash" | tee -a $O/kbench.txt
  done
done; done
unset MDGEN_AMD_LIB
for v in product NOFIRSTPF product NOFIRSTPF; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
    timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v $wl', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
  done
done
