#!/bin/bash
# GPU: per-kernel time of k16_linear_wide in the training step for each experiment library built by
#   KFILE=k_wide16 KPFX=WIDE bash scripts/micro/flash_variants.sh NOLOAD NOMMA NOSTORE
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib="$R/gpurun_out/dev_libs/libmdgen_amd_$v.so"
  rm -rf $O/wv_$v
  (cd /tmp && MDGEN_AMD_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wv_$v -o k -- python $R/scripts/train_bench.py 1 250 256 1 16 > $O/wv_$v.log 2>&1 < /dev/null)
  f=$(find $O/wv_$v -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "== variant '${v:-product}'"; [ -n "$f" ] && grep -E "k16_linear_wide|k16_dw_wide" "$f" | cut -d, -f1-4 | cut -c1-140
done
