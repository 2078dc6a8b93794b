#!/bin/bash
# Round-3 re-measurement after the training-step work (the sampler's kernels are unchanged since scripts/gpu_r03_final.sh):
# parity suite, smoke, the default bench line (with the extra legs: the training step is one of them), the fp32-mode line,
# rocprofv3 kernel stats of the training step in both operand precisions, PMC HBM traffic and SQ counters of its big kernels.
# Everything lands in gpurun_out/; copy the summaries to profiles/r03_*.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider < /dev/null 2>&1 | grep -v amdgpu.ids | tail -170 > $O/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | grep -v amdgpu.ids > $O/r03_smoke.log
timeout 900 python bench.py < /dev/null 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2.json
timeout 900 python bench.py --precision fp32 --steps 1 --warmup 1 --no-cpu-baseline --no-extra < /dev/null 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_cfg2_fp32.json
for p in 16 32; do
  bash scripts/prof_train.sh $p > /dev/null 2>&1
  cp $O/kernel_stats_train$p.csv $O/r03_kernel_stats_train_cfg5_p$p.csv 2>/dev/null
done
bash scripts/pmc_train.sh "k16_linear_wdma|k16_dw_wide|k16_attn" > $O/r03_pmc_traffic_train.txt 2>&1
bash scripts/pmc_train_sq.sh "k16_linear_wdma|k16_dw_wide|k16_attn" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA" > $O/r03_pmc_sq_train.txt 2>&1
for p in 16 32; do timeout 120 python scripts/train_bench.py 1 250 256 3 $p < /dev/null 2>&1 | grep -v amdgpu.ids | tail -1; done > $O/r03_train_bench.txt
tail -3 $O/r03_pytest_gpu.log; tail -2 $O/r03_smoke.log; cut -c1-400 $O/r03_bench_cfg2.json; cut -c1-200 $O/r03_bench_cfg2_fp32.json; cat $O/r03_train_bench.txt
head -16 $O/r03_pmc_traffic_train.txt
