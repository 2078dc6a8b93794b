#!/bin/bash
# round 4: weight gradients on a second stream (option train_streams): identical gradients; step time 1 vs 2
mkdir -p gpurun_out/r04k; O=gpurun_out/r04k; rm -f $O/bench.log
timeout 600 python scripts/r04/train_streams_check.py > $O/check.log 2>&1; echo "check exit $?" >> $O/check.log
tail -3 $O/check.log
for ns in 1 2 1 2; do
  timeout 300 python scripts/train_bench.py 1 250 256 5 16 train_streams=$ns 2>&1 | tail -1 >> $O/bench.log
done
cat $O/bench.log
