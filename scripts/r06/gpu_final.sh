#!/bin/bash
# Round-6 measurement call on the final build: smoke, the full GPU suite, the default bench line (cfg-2 with roofline, cpu_baseline and
# the extra legs), rocprofv3 kernel stats (cfg-2 single stream, ATLAS, the two small-N shapes), PMC HBM traffic (cfg-2, ATLAS), one SQ
# counter pass for the big kernels.  Outputs land in gpurun_out/r06final; the summaries are copied to profiles/r06_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06final; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $O/pytest_gpu.log
grep "passed\|failed" $O/pytest_gpu.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
prof() {  # name workload
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o ktrace -- python $R/bench.py --workload $2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_$1.log 2>&1)
  find $O/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_'$1'.csv; head -9 {} | cut -c1-150'
  rm -rf $O/prof_$1
}
prof cfg2 tetrapeptide_fwdsim_crop4_T1000_B16
prof atlas atlas_crop256_T250_B1
prof tps_B32 tetrapeptide_tps_crop4_T100_B32
prof B1_T1000 tetrapeptide_fwdsim_crop4_T1000_B1
bash scripts/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
tail -40 $O/pmc_traffic.txt
bash scripts/pmc.sh "k_mlp_rows|k_flash_proj|k_ln_qkv_attn4|k_ln_qkv<false, false>" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" > $O/pmc_sq.txt 2>&1
tail -2 $O/smoke.log; cut -c1-700 $O/bench_cfg2.json; echo; tail -60 $O/pmc_sq.txt
rm -rf gpurun_out/pmc_FETCH_SIZE_* gpurun_out/pmc_WRITE_SIZE_* gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
