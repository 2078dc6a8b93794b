#!/bin/bash
# round 4: ablation builds of the training attention kernels (scripts/micro/flash_variants.sh KFILE=k_attn16 KPFX=ATTN16 ...),
# per-kernel averages from rocprofv3 --stats -> gpurun_out/r04a/summary.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r04a; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
: > $O/summary.txt
for v in product "$@"; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=$R/gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o kt -- python $R/scripts/r04/attn16_run.py 5 > $O/run_$v.log 2>&1 < /dev/null)
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v  $(tail -1 $O/run_$v.log)" >> $O/summary.txt
  python - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k16_attn" in r["Name"]:
        print(f"   {r['Name'][10:36]:28s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:7.1f} us  min {float(r['MinNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
PY
  rm -rf $O/prof_$v
done
cat $O/summary.txt
