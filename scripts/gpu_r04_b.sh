#!/bin/bash
# round 4, call B: the chain kernel (k_chain_l4): parity vs oracle / panel kernels, then bench A/B (chain_path 0 vs 1)
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "chain_kernel or (training_attention_kernels_unit and (1000 or 1001))" > gpurun_out/r04b/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04b/pytest.log
tail -15 gpurun_out/r04b/pytest.log
for cp in 0 1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option chain_path=$cp > gpurun_out/r04b/bench_chain$cp.json 2> gpurun_out/r04b/bench_chain$cp.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04b/bench_chain$cp.json"))
    print("chain_path=$cp", d["value"], d["ms_per_step"], d["roofline"]["by_kernel_ms_per_call"])
except Exception as e:
    print("chain_path=$cp failed", e); print(open("gpurun_out/r04b/bench_chain$cp.err").read()[-2000:])
PY
done
