#!/bin/bash
# round 5, call G: k_ln_qkv_attn4 -- weight-ring depth of the K / V GEMMs (experiment builds, values identical), per-kernel timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05g; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh K3V3 K3V4 K4V4 > $O/build.log 2>&1; tail -3 $O/build.log
for v in product K3V3 K3V4 K4V4 product; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 3 2>&1 | grep "parity\|S=3\|attn_L" | tee -a $O/kbench.txt
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  bench', d['value'], d['ms_per_step'])" | tee -a $O/kbench.txt
done
