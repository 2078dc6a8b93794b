#!/bin/bash
# round 6, call D: the embedding-as-tail form (mlp_tail 2 against 1 / 0), new tests (dispatch registry cases, IPA table at S = 49, one-rank
# RCCL self-test), k_ln_qkv_attn4 with early residual rows (experiment build), and a full default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06d; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -s -k "registry or ipa_table or rccl or fold or headline or row_owner or cfg4_full or B16 or multi_block or rollout or graph or inference" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -12
KFILE=k_gemm KPFX=ATTN4 bash scripts/micro/flash_variants.sh EARLY > $O/build.log 2>&1; tail -1 $O/build.log
run_k() { echo "== $1 $2 $3" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 $3 2>&1 | grep -v parity | grep -v amdgpu | head -9 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline $3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
C2=tetrapeptide_fwdsim_crop4_T1000_B16; AT=atlas_crop256_T250_B1
for rep in 1 2; do
  unset MDGEN_AMD_LIB
  run_k product $C2; run_k product $C2 mlp_tail=1; run_k product $AT
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_EARLY.so; run_k ATTN4_EARLY $C2
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB
  run_b product $C2; run_b product $C2 "--option mlp_tail=1"; run_b product $C2 "--option mlp_tail=0"
  run_b product $AT; run_b product $AT "--option mlp_tail=1"
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_EARLY.so; run_b ATTN4_EARLY $C2
done
unset MDGEN_AMD_LIB
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json
