#!/bin/bash
# SQ counter passes over the training step (ATLAS 256 x 250, train_precision 16) for kernels matching a regex, grouped by grid.
# usage: scripts/pmc_train_sq.sh "<regex>" "CTR1 CTR2 ..." ["CTR ..." more passes]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp PYTHONUNBUFFERED=1
RE="$1"; shift
i=0
for ctrs in "$@"; do
  i=$((i+1)); rm -rf $O/sq$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "$RE" --output-format csv -d $O/sq$i -o pmc -- python $R/scripts/train_bench.py 1 250 256 1 16 > $O/sq$i.log 2>&1 < /dev/null)
  python - "$O/sq$i" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:28], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -kv[0][1])[:8]:
    print(k, "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(d.items())))
PY
done
