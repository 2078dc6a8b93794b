#!/bin/bash
# round 5, the tracked evidence of the FINAL build from ONE box and one call: the default bench line (-> profiles/r05_bench_cfg2.json)
# and the rocprofv3 kernel stats of cfg-2 (single stream), ATLAS, the TPS shard and B = 1 (-> profiles/r05_kernel_stats_*.csv).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05stats; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
cut -c1-330 $O/bench_cfg2.json; echo
prof() {  # name workload
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o ktrace -- python $R/bench.py --workload $2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_$1.log 2>&1)
  find $O/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_'$1'.csv; head -6 {} | cut -c1-110'
  rm -rf $O/prof_$1
}
prof cfg2 tetrapeptide_fwdsim_crop4_T1000_B16
prof atlas atlas_crop256_T250_B1
prof tps_B32 tetrapeptide_tps_crop4_T100_B32
prof B1_T1000 tetrapeptide_fwdsim_crop4_T1000_B1
