#!/bin/bash
# round 5: shader clock and package power while the headline rollout runs (rocm-smi sampled once a second beside bench.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05clk; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python bench.py --steps 40 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > $O/bench.log 2>&1 &
BP=$!
: > $O/clocks.txt
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')" >> $O/clocks.txt
  sleep 1
done
wait $BP
tail -1 $O/bench.log | cut -c1-160 >> $O/clocks.txt
cat $O/clocks.txt
