#!/bin/bash
# round-2 GPU call B: parity suite on the restructured attention kernel, per-kernel timings, issue-rate microbench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
(hipcc --offload-arch=gfx950 -O3 -w -o /tmp/issue_rate scripts/micro/issue_rate.hip && timeout 120 /tmp/issue_rate) > $O/issue_rate2.txt 2>&1 &
timeout 1500 python -m pytest tests -m gpu -q -rA -s -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -150 > $O/pytest_gpu_b.log
wait
timeout 300 python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -v amdgpu.ids > $O/kbench_cfg2.log
timeout 300 python scripts/kbench.py atlas_crop256_T250_B1 2 2>&1 | grep -v amdgpu.ids > $O/kbench_atlas.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > $O/bench_b.log
WL=tetrapeptide_fwdsim_crop4_T1000_B16 bash scripts/pmc.sh "k_flash" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > $O/pmc_flash_b.txt 2>&1
grep -E "passed|failed|Error" $O/pytest_gpu_b.log | tail -5
cat $O/kbench_cfg2.log | tail -18; cat $O/kbench_atlas.log | tail -18; tail -1 $O/bench_b.log | cut -c1-200
cat $O/issue_rate2.txt
