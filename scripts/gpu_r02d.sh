#!/bin/bash
# round-2 GPU call D: what bounds the attention kernel -- product vs no-K/V-loads variants; fallback evidence
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
for v in PRODUCT NOLOAD PRODUCT; do
  if [ $v = PRODUCT ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=$R/scripts/micro/dev_libs/libmdgen_amd_$v.so; fi
  echo "== $v"; timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -E "parity|flash_T|mlp  |ln_qkv_T"
done > $O/flash_variants.txt 2>&1
unset MDGEN_AMD_LIB
timeout 300 python scripts/kbench.py atlas_crop256_T250_B1 2 2>&1 | grep -E "flash|mlp  " >> $O/flash_variants.txt
cat $O/flash_variants.txt
timeout 600 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "attention_fixed or golden or cfg4_full_size_vs or shapes" 2>&1 | grep -v amdgpu.ids | tail -12
echo "== NOFALLBACK variant on the same test (expected: non-finite at the overflowing scales)"
MDGEN_AMD_LIB=$R/scripts/micro/dev_libs/libmdgen_amd_NOFALLBACK.so timeout 300 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "attention_fixed" 2>&1 | grep -v amdgpu.ids | grep -E "attention loops|assert|passed|failed" | head
