cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -E "final_euler|^  mlp" 
rm -rf $O/prof1 $O/prof_atlas
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/rocprof1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atlas -o ktrace -- python $R/bench.py --workload atlas_crop256_T250_B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/rocprof_atlas.log 2>&1)
bash scripts/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
for wl in atlas_crop256_T250_B1 tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_$wl.json
done
timeout 900 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2_fp32.json
head -9 $O/prof1/ktrace_kernel_stats.csv | cut -c1-120
for f in $O/bench_atlas*.json $O/bench_tetrapeptide_tps*.json $O/bench_tetrapeptide_fwdsim_crop4_T1000_B1.json $O/bench_cfg2_fp32.json; do cut -c1-110 $f; done
