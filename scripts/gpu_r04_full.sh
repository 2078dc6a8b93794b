#!/bin/bash
# round 4: the whole -m gpu suite (log -> profiles/r04_pytest_gpu.log)
mkdir -p gpurun_out/r04full
timeout 2400 python -m pytest tests/ -x -q -m gpu -s > gpurun_out/r04full/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04full/pytest.log
tail -5 gpurun_out/r04full/pytest.log
