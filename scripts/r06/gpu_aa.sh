#!/bin/bash
# round 6, call AA: cfg-3's shard (B 32 x T 100) with the 32-row form of k_ln_qkv_attn4 at two workgroups per CU (option small_half2): bench A/B, kbench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06aa; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TP=tetrapeptide_tps_crop4_T100_B32
run_b() { timeout 300 python bench.py --workload $TP --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2 3; do run_b base ""; run_b half2 "--option small_half2=1"; done
timeout 300 python scripts/kbench.py $TP 3 2>&1 | grep -v parity | grep -v amdgpu | head -8
