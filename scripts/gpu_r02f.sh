#!/bin/bash
# round-2 GPU call F: fp32-MFMA token embedding: timings + full suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -v amdgpu.ids | tail -16
python scripts/kbench.py atlas_crop256_T250_B1 2 2>&1 | grep -E "parity|embed|flash|mlp  "
python scripts/kbench.py tetrapeptide_tps_crop4_T100_B32 2 2>&1 | grep -E "embed|flash|mlp  "
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -12 > $O/pytest_gpu_f.log
tail -5 $O/pytest_gpu_f.log
