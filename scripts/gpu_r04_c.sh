#!/bin/bash
# round 4, call C: chain kernel parity (with the query-slot zero fill), phase stamps, experiment builds
mkdir -p gpurun_out/r04c
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "chain_kernel" > gpurun_out/r04c/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04c/pytest.log
grep "chain vs\|passed\|failed" gpurun_out/r04c/pytest.log | tail -12
timeout 300 python scripts/r04/chain_stamps.py 2>&1 | tail -1 | tee gpurun_out/r04c/stamps.txt
for v in NOSTORE NOEPI; do
  MDGEN_AMD_LIB=scripts/micro/dev_libs/libmdgen_amd_$v.so timeout 300 python scripts/r04/chain_stamps.py 2>&1 | tail -1 | tee -a gpurun_out/r04c/stamps.txt
done
for cp in 0 1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option chain_path=$cp > gpurun_out/r04c/bench_chain$cp.json 2> gpurun_out/r04c/bench_chain$cp.err
  python -c "
import json
d=json.load(open('gpurun_out/r04c/bench_chain$cp.json')); print('chain_path=$cp', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_call'])"
done
