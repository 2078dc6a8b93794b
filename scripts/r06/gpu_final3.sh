#!/bin/bash
# Round-6 measurement call on the FINAL build (third session: sequence-resident training attention, deferred gated updates): smoke, the full GPU suite, the default
# bench line as the driver runs it (cfg-2 with roofline, cpu_baseline and the extra legs), rocprofv3 kernel stats (cfg-2 single stream,
# ATLAS, the two small-N shapes), PMC HBM traffic (cfg-2, ATLAS), one SQ counter pass for the big kernels, clocks / power beside the rollout.
# Outputs land in gpurun_out/r06final3; the summaries are copied to profiles/r06_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06final3; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
cut -c1-400 $O/bench_cfg2.json; echo
prof() {  # name workload
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o ktrace -- python $R/bench.py --workload $2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-graph --streams 1 > $O/rocprof_$1.log 2>&1)
  find $O/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$O'/kernel_stats_'$1'.csv; head -7 {} | cut -c1-150'
  rm -rf $O/prof_$1
}
prof cfg2 tetrapeptide_fwdsim_crop4_T1000_B16
prof atlas atlas_crop256_T250_B1
prof tps_B32 tetrapeptide_tps_crop4_T100_B32
prof B1_T1000 tetrapeptide_fwdsim_crop4_T1000_B1
# clocks / power beside the headline rollout
timeout 300 python bench.py --steps 40 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > $O/clk_bench.log 2>&1 &
BP=$!
: > $O/clocks.txt
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')" >> $O/clocks.txt
  sleep 1
done
wait $BP
tail -1 $O/clk_bench.log | cut -c1-160 >> $O/clocks.txt
tail -4 $O/clocks.txt
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $O/pytest_gpu.log
grep "passed\|failed" $O/pytest_gpu.log | tail -3
bash scripts/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
tail -12 $O/pmc_traffic.txt
bash scripts/pmc.sh "k_mlp_rows|k_flash_proj|k_ln_qkv_attn4|k_ln_qkv<false, false>" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" > $O/pmc_sq.txt 2>&1
tail -30 $O/pmc_sq.txt
rm -rf gpurun_out/pmc_FETCH_SIZE_* gpurun_out/pmc_WRITE_SIZE_* gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
# training step (cfg-5's per-GPU size, bf16 operands): step time x3, rocprofv3 kernel stats, per-queue timeline, attention kernels alone
for rep in 1 2 3; do timeout 300 python scripts/train_bench.py 1 250 256 10 16 2>&1 | tail -1; done | tee $O/train_bench.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o kt -- python $R/scripts/train_bench.py 1 250 256 3 16 > $O/rocprof_train.log 2>&1 < /dev/null)
find $O/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_train.csv
rm -rf $O/prof_train
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d $O/prof_tl -o kt -- python $R/scripts/train_bench.py 1 250 256 3 16 > $O/run_tl.log 2>&1 < /dev/null)
f=$(find $O/prof_tl -name "*kernel_trace.csv" | head -1)
python scripts/r04/train_timeline.py "$f" list 2>&1 | head -60 > $O/train_timeline.txt
rm -rf $O/prof_tl
head -6 $O/train_timeline.txt
for prec in 16 160; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a$prec -o kt -- python $R/scripts/r04/attn16_run.py 5 $prec > $O/run_a$prec.log 2>&1 < /dev/null)
  f=$(find $O/prof_a$prec -name "*kernel_stats.csv" | head -1)
  echo "== mdgen_debug_train_attention precision $prec (ATLAS per-GPU shape, both axes, 10 launches)" >> $O/train_attention_kernels.txt
  grep "k16_attn" "$f" | cut -d, -f1-4 | cut -c1-140 >> $O/train_attention_kernels.txt
  rm -rf $O/prof_a$prec
done
cat $O/train_attention_kernels.txt
