#!/bin/bash
# Round-3 measurement round: parity suite, smoke, the default bench line (cfg-2 with roofline, cpu_baseline and the extra legs),
# fp32-mode line, rocprofv3 kernel stats (cfg-2 single stream, cfg-4, training step in both operand precisions), PMC HBM
# traffic (cfg-2, cfg-4), SQ counters of the dominant kernels, in-kernel stamps of k_mlp_rows.  Everything lands in
# gpurun_out/; copy the summaries to profiles/r03_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -150 > $O/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/r03_smoke.log
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2.json
timeout 600 python bench.py --option mlp_path=0 --no-cpu-baseline --no-extra 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2_panel_mlp.json
timeout 600 python bench.py --option fuse_proj=1 --no-cpu-baseline --no-extra 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2_fuse_proj.json
timeout 900 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2_fp32.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof1 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/r03_rocprof1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_atlas -o ktrace -- python $R/bench.py --workload atlas_crop256_T250_B1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/r03_rocprof_atlas.log 2>&1)
for p in 16 32; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_prof_train$p -o ktrace -- python $R/scripts/train_bench.py 1 250 256 2 $p > $O/r03_rocprof_train$p.log 2>&1)
done
bash scripts/pmc_traffic.sh > $O/r03_pmc_traffic.txt 2>&1
bash scripts/pmc.sh "k_flash|k_mlp_rows" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" > $O/r03_pmc_sq.txt 2>&1
timeout 300 python scripts/r03/mlp_rows_variants.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_mlp_rows_stamps.txt
tail -3 $O/r03_pytest_gpu.log; tail -2 $O/r03_smoke.log; cut -c1-300 $O/r03_bench_cfg2.json
for f in $O/r03_bench_cfg2_panel_mlp.json $O/r03_bench_cfg2_fuse_proj.json $O/r03_bench_cfg2_fp32.json; do cut -c1-160 $f; done
head -12 $O/r03_prof1/ktrace_kernel_stats.csv | cut -c1-150
tail -24 $O/r03_pmc_traffic.txt; cat $O/r03_mlp_rows_stamps.txt
