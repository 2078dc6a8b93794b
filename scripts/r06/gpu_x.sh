#!/bin/bash
# round 6, call X: what k16_attn_bwd_seq's time is made of: ablation builds (fill loads / pass loops / result stores left out)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06x; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_attn16 KPFX=ATTN16 bash scripts/micro/flash_variants.sh "$@" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_attention_kernels_unit" 2>&1 | tail -4 | tee $O/pytest.log
: > $O/summary.txt
for v in product "$@"; do
  if [ $v = product ]; then unset MDGEN_AMD_LIB; else export MDGEN_AMD_LIB=$R/gpurun_out/dev_libs/libmdgen_amd_$v.so; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o kt -- python $R/scripts/r04/attn16_run.py 5 16 > $O/run_$v.log 2>&1 < /dev/null)
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v" >> $O/summary.txt
  python - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k16_attn" in r["Name"]:
        print(f"   {r['Name'][:40]:40s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:7.1f} us  min {float(r['MinNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
PY
  rm -rf $O/prof_$v
done
cat $O/summary.txt
