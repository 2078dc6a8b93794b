#!/bin/bash
# round 4: chain kernel iteration: parity + stamps (+ NOSTORE experiment build)
mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "chain_kernel" > gpurun_out/r04f/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04f/pytest.log
grep "chain vs\|passed\|failed\|Error" gpurun_out/r04f/pytest.log | tail -12
timeout 300 true

for cp in 0 1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option chain_path=$cp > gpurun_out/r04f/bench_chain$cp.json 2> gpurun_out/r04f/bench_chain$cp.err
  python -c "
import json
d=json.load(open('gpurun_out/r04f/bench_chain$cp.json')); print('chain_path=$cp', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_call'])"
done
