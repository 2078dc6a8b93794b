#!/bin/bash
# round 5, call J: intra-sample split for B = 1 (option split_sample): parity (bit-identity), ATLAS A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05j; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "intra_sample or inference_end_to_end or multi_block_rollout" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -8
run() { timeout 300 python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for i in 1 2; do
run --workload atlas_crop256_T250_B1 --option split_sample=0
run --workload atlas_crop256_T250_B1 --option split_sample=1
done
run --workload atlas_crop256_T250_B1 --option split_sample=1 --option flash_proj=2
