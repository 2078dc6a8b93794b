#!/bin/bash
# first GPU contact: parity tests (all, verbose), smoke, short bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/dev.txt
timeout 900 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; tail -3 gpurun_out/bench.log
