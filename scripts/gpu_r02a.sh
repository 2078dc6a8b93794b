#!/bin/bash
# round-2 GPU call A: full parity suite, bench lines (cfg-2 / cfg-4 / cfg-3 / B=1), SQ PMC counters for the two
# dominant kernels, rocprof kernel stats, instruction-rate micro-benchmark.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
(hipcc --offload-arch=gfx950 -O3 -w -o /tmp/issue_rate scripts/micro/issue_rate.hip && timeout 120 /tmp/issue_rate) > $O/issue_rate.txt 2>&1 &
timeout 1500 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -150 > $O/pytest_gpu.log
wait
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.log
timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids > $O/bench.log
timeout 600 python bench.py --steps 3 --warmup 1 --workload atlas_crop256_T250_B1 2>&1 | grep -v amdgpu.ids > $O/bench_atlas.log
timeout 300 python bench.py --steps 3 --warmup 1 --workload tetrapeptide_tps_crop4_T100_B32 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/bench_tps.log
timeout 300 python bench.py --steps 3 --warmup 1 --workload tetrapeptide_fwdsim_crop4_T1000_B1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/bench_b1.log
# PMC passes (counters only, no tracing domains)
(cd /tmp && rocprofv3 -L > $O/counters_all.txt 2>&1; grep -o "SQ_[A-Z0-9_]*" $O/counters_all.txt | sort -u > $O/counters_sq.txt)
WL=tetrapeptide_fwdsim_crop4_T1000_B16 bash scripts/pmc.sh "k_flash|k_mlp|k_ln_qkv" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
  "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" > $O/pmc_sq.txt 2>&1
# kernel stats: cfg-2 single stream, cfg-4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $O/rocprof1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_atlas -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --workload atlas_crop256_T250_B1 > $O/rocprof_atlas.log 2>&1)
tail -5 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench.log | cut -c1-400; tail -1 $O/bench_atlas.log | cut -c1-300; tail -1 $O/bench_tps.log | cut -c1-300; tail -1 $O/bench_b1.log | cut -c1-300
cat $O/issue_rate.txt
