#!/bin/bash
# round 4: chain kernel iteration: parity + stamps (+ NOSTORE experiment build)
mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "chain_kernel" > gpurun_out/r04f/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04f/pytest.log
grep "chain vs\|passed\|failed\|Error" gpurun_out/r04f/pytest.log | tail -12
timeout 300 python scripts/r04/chain_stamps.py 2>&1 | tail -1 | tee gpurun_out/r04f/stamps.txt
MDGEN_AMD_LIB=scripts/micro/dev_libs/libmdgen_amd_NOSTORE.so timeout 300 python scripts/r04/chain_stamps.py 2>&1 | tail -1 | tee -a gpurun_out/r04f/stamps.txt
