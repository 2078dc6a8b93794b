#!/bin/bash
# round 6, call S: cfg-3's shard on two / four sub-batch streams now that the L = 4 kernel has a 32-row form (r04 found one stream best)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06s; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TP=tetrapeptide_tps_crop4_T100_B32
run_b() { timeout 300 python bench.py --workload $1 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  run_b $TP ""; run_b $TP "--option streams=2"; run_b $TP "--option streams=4"
done
