#!/bin/bash
# round 6: one training step (cfg-5 per-GPU size, bf16 operands) as a full rocprofv3 kernel timeline (every launch, per queue) -> gpurun_out/r06tl/timeline.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06tl; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o kt -- python $R/scripts/train_bench.py 1 250 256 3 16 "$@" > $O/run.log 2>&1 < /dev/null)
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
tail -1 $O/run.log
python scripts/r04/train_timeline.py "$f" list > $O/timeline.txt 2>&1
head -4 $O/timeline.txt
rm -rf $O/prof
