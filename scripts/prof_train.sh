#!/bin/bash
# rocprofv3 kernel stats of the training step at ATLAS 256 x 250 per GPU, for one train_precision (default 16).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; P=${1:-16}; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train$P -o ktrace -- python $R/scripts/train_bench.py 1 250 256 2 $P > $O/rocprof_train$P.log 2>&1 < /dev/null)
f=$(find $O/prof_train$P -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats_train$P.csv; head -28 "$f" | cut -c1-150; else echo "no stats file"; tail -5 $O/rocprof_train$P.log; fi
