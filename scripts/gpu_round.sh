#!/bin/bash
# full GPU check: parity suite, smoke, bench, rocprof kernel stats (product mode + single-stream mode)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids > gpurun_out/bench.log
R=$PWD
# (a) product mode: two sub-batches on two streams -> half-batch launches, overlapping
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/rocprof.log 2>&1)
# (b) single stream: full-batch launches, the mode bench.py's hipEvent roofline leg measures
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -o ktrace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --streams 1 > $R/gpurun_out/rocprof1.log 2>&1)
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -2; tail -2 gpurun_out/bench.log
head -4 gpurun_out/prof1/ktrace_kernel_stats.csv
