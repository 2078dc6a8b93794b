#!/bin/bash
# round 6, call O: k_ln_qkv_attn4<true, true>, 32-row workgroups for launches of at most one of them per CU (B <= 2 at T 1000, the IPA stack):
# parity, phase stamps, per-launch and end-to-end A/B against 64-row workgroups (ATTN4_FULL build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06o; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s -x -k "small_ or split or registry or forward_vs or golden or fwd or inference_sim or cfg1 or tps or ipa_table or headline" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-300 | tail -12
KFILE=k_gemm KPFX=ATTN4 bash scripts/micro/flash_variants.sh FULL > $O/build.log 2>&1; tail -1 $O/build.log
KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS >> $O/build.log 2>&1; tail -1 $O/build.log
B1=tetrapeptide_fwdsim_crop4_T1000_B1
MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/attn4_stamps.py $B1 2>&1 | grep -v amdgpu | tail -12 | tee -a $O/stamps.txt
run_k() { echo "== $1 $2" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $B1
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_FULL.so; run_k FULL $B1
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $B1
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_FULL.so; run_b FULL $B1
done
