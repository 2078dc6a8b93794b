#!/bin/bash
# round 5, call F: the co-scheduling A/B (round-4 verdict item 3): MFMA-bound MLP and VALU-bound attention side by side on a CU.
# Stream 2 starts `stream_offset` us after stream 1 (so the two sub-batches sit in different kernels); `mlp_cap` holds the four-wave
# panel MLP kernel (256 registers per wave) at one workgroup per CU, which leaves each SIMD room for one k_flash_proj wave (227).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05f; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
run
for off in 50 100 150 200 250; do run --option stream_offset=$off; done
run --option mlp_path=0
run --option mlp_path=0 --option mlp_cap=1
for off in 100 150 200 250; do run --option mlp_path=0 --option mlp_cap=1 --option stream_offset=$off; done
run --option mlp_path=0 --option stream_offset=150
run
