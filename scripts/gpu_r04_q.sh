#!/bin/bash
# round 4, last measurement call: -m gpu suite, default bench line, PMC traffic of the training kernels (final build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O
bash scripts/gpu_r04_full.sh
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_cfg2.json
cut -c1-300 $O/bench_cfg2.json
bash scripts/pmc_train.sh "k16_linear|k16_dw|k16_attn" > $O/pmc_traffic_train.txt 2>&1
head -12 $O/pmc_traffic_train.txt
rm -rf gpurun_out/pmc_train_FETCH_SIZE gpurun_out/pmc_train_WRITE_SIZE
