#!/bin/bash
# round 6, call Z: every training test of the GPU suite, then the step-time A/B of the options given as arguments (gpu_y.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06z; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "train or ddp or adam or rccl" 2>&1 | tail -8 | tee $O/pytest.log
bash scripts/r06/gpu_y.sh "$@"
