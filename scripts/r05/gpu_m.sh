#!/bin/bash
# round 5, call M: k_flash_proj8 parity (all forms), defaults A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05m; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "flash_proj or full_size_properties or forward_cfg4_full or graph_replay or dual_stream or inference_headline" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit\|128-row\|B 8\|B 7" | tail -16
run() { timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for i in 1 2; do
  run --option flash_proj_form=4
  run
  run --workload atlas_crop256_T250_B1
done
