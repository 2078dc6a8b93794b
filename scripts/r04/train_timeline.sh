#!/bin/bash
# round 4: kernel timeline of one training step at cfg-5's per-GPU size (bf16 operands) -> gpurun_out/r04t/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r04t; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o kt -- python $R/scripts/train_bench.py 1 250 256 3 16 > $O/run.log 2>&1 < /dev/null)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
tail -2 $O/run.log
python scripts/r04/train_timeline.py "$f" list > $O/timeline.txt 2>&1
head -45 $O/timeline.txt
rm -rf $O/prof
