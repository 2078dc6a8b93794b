#!/bin/bash
# round 6, call W: the training attention kernels alone at the ATLAS per-GPU shape (scripts/r04/attn16_run.py): sequence-resident forms
# (precision 16) against the chunked kernels (160), rocprofv3 averages; unit test first
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06w; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_attention_kernels_unit" 2>&1 | tail -8 | tee $O/pytest.log
: > $O/summary.txt
for prec in 16 160; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$prec -o kt -- python $R/scripts/r04/attn16_run.py 5 $prec > $O/run_$prec.log 2>&1 < /dev/null)
  f=$(find $O/prof_$prec -name "*kernel_stats.csv" | head -1)
  echo "== precision $prec  $(tail -1 $O/run_$prec.log)" >> $O/summary.txt
  python - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k16_attn" in r["Name"]:
        print(f"   {r['Name'][:40]:40s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:7.1f} us  min {float(r['MinNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
PY
  rm -rf $O/prof_$prec
done
cat $O/summary.txt
