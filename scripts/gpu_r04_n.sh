#!/bin/bash
# round 4: rocprofv3 kernel stats at the small-N shapes (cfg-3's shard, B = 1), defaults of the final build, single stream, no graph
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r04n; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for w in tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o ktrace -- python $R/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph > $O/rocprof_$w.log 2>&1)
  find $O/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_'$w'.csv; head -9 {} | cut -c1-150'
  rm -rf $O/prof_$w
done
