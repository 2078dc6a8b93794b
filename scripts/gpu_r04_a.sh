#!/bin/bash
# round 4, call A: the new parity tests (headline regime T = 1000, TPS vs the reference golden, cond_interval, resume, INTEGRATION stub)
# + the unchanged sampler's bench line as this round's baseline on this box
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s \
  -k "native or headline or tps_vs_oracle or fp32_mode_forward or prep_batch or resume or integration_md or rccl or inference_end_to_end or S49 or multi_block or chained or (training_attention_kernels_unit and (1000 or 1001)) or ddp_two" \
  > gpurun_out/r04a/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
tail -c 600 gpurun_out/r04a/bench.json
