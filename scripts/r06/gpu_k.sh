#!/bin/bash
# round 6, call K: k_ln_qkv_attn4<true, true>, the eight-wave form of the L = 4 sub-layer for launches of at most one workgroup per CU
# (B = 1, cfg-3's shard, the IPA stack): parity (every fixture-size L = 4 test runs through it), per-launch and end-to-end A/B against
# the four-wave form (ATTN4_FOUR build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06k; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s -x -k "small_ or split or registry or forward_vs or golden or fwd or inference_sim or cfg1 or tps or ipa_table or headline" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-300 | tail -12
KFILE=k_gemm KPFX=ATTN4 bash scripts/micro/flash_variants.sh FOUR > $O/build.log 2>&1; tail -1 $O/build.log
B1=tetrapeptide_fwdsim_crop4_T1000_B1; TP=tetrapeptide_tps_crop4_T100_B32
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $B1; run_k product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_FOUR.so; run_k FOUR $B1; run_k FOUR $TP
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $B1; run_b product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_FOUR.so; run_b FOUR $B1; run_b FOUR $TP
done
