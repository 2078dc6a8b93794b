#!/bin/bash
# round-2 GPU call C: attention kernel v3 (fixed anchor + sticky range test, buffer loads): parity subset + kernel timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -v amdgpu.ids > $O/kbench_cfg2.log
timeout 300 python scripts/kbench.py atlas_crop256_T250_B1 2 2>&1 | grep -v amdgpu.ids > $O/kbench_atlas.log
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not training and not trainer and not fp32 and not cfg5" 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest_gpu_c.log
cat $O/kbench_cfg2.log | tail -18; cat $O/kbench_atlas.log | tail -18
tail -8 $O/pytest_gpu_c.log
