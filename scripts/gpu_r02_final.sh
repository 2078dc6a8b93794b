#!/bin/bash
# Round-2 measurement round: parity suite, smoke, bench lines (cfg-2 default with roofline + cpu_baseline; cfg-4, cfg-3,
# B = 1, fp32 mode), rocprofv3 kernel stats (cfg-2 single stream, cfg-4), PMC HBM traffic (cfg-2, cfg-4), SQ counters of
# the two dominant kernels, in-kernel stamps.  Everything lands in gpurun_out/; copy the summaries to profiles/r02_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -120 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.log
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
for wl in atlas_crop256_T250_B1 tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_$wl.json
done
timeout 900 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2_fp32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/rocprof1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atlas -o ktrace -- python $R/bench.py --workload atlas_crop256_T250_B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/rocprof_atlas.log 2>&1)
bash scripts/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
bash scripts/pmc.sh "k_flash|k_mlp" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" > $O/pmc_sq.txt 2>&1
timeout 300 python scripts/trace_mlp.py 2>&1 | grep -v amdgpu.ids > $O/mlp_trace.txt
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench_cfg2.json
for f in $O/bench_atlas*.json $O/bench_tetrapeptide_tps*.json $O/bench_tetrapeptide_fwdsim_crop4_T1000_B1.json $O/bench_cfg2_fp32.json; do cut -c1-160 $f; done
head -12 $O/prof1/ktrace_kernel_stats.csv | cut -c1-150
cat $O/pmc_traffic.txt | tail -20
