#!/bin/bash
# round 4: XCD-aware workgroup order of the training attention kernels: unit tests, kernel times, step time, HBM traffic
mkdir -p gpurun_out/r04p; O=gpurun_out/r04p
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_attention_kernels_unit or bf16_operand_kernels_vs_exact or gradients_bf16_operands_vs_reference or second_stream" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash scripts/r04/attn16_variants.sh 2>&1 | tail -4
for i in 1 2; do timeout 300 python scripts/train_bench.py 1 250 256 5 16 2>&1 | tail -1; done
bash scripts/pmc_train.sh "k16_attn" 2>&1 | head -4
rm -rf gpurun_out/pmc_train_FETCH_SIZE gpurun_out/pmc_train_WRITE_SIZE
