#!/bin/bash
# Round-4 measurement call: smoke, the default bench line (cfg-2 with roofline, cpu_baseline and the extra legs), rocprofv3 kernel
# stats (cfg-2 single stream, ATLAS), PMC HBM traffic (cfg-2, ATLAS), chain-kernel stamps.  Outputs land in gpurun_out/r04final;
# the summaries are copied to profiles/r04_*.  (The parity suite is scripts/gpu_r04_full.sh.)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r04final; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.log
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_cfg2.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atlas -o ktrace -- python $R/bench.py --workload atlas_crop256_T250_B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_atlas.log 2>&1)
bash scripts/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
timeout 300 python scripts/r04/chain_stamps.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/chain_stamps.txt
tail -2 $O/smoke.log; cut -c1-400 $O/bench_cfg2.json
find $O/prof_cfg2 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_cfg2.csv; head -9 {} | cut -c1-150'
find $O/prof_atlas -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_atlas.csv; head -10 {} | cut -c1-150'
tail -24 $O/pmc_traffic.txt; cat $O/chain_stamps.txt
rm -rf $O/prof_cfg2 $O/prof_atlas gpurun_out/pmc_FETCH_SIZE_* gpurun_out/pmc_WRITE_SIZE_*
