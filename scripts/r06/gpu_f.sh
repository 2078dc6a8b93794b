#!/bin/bash
# round 6, call F: small-launch forms with all rows requested at once (BURST; NOBURST experiment build for the A/B) at B = 1 and the TPS
# shard; phase stamps of the MLP launch with both tails; tests of what changed
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06f; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python scripts/r06/tail_stamps.py 2>&1 | grep -v amdgpu | tee $O/tail_stamps.txt
timeout 300 python scripts/r06/tail_stamps.py embed_split=0 2>&1 | grep -v amdgpu | tee -a $O/tail_stamps.txt
timeout 1800 python -m pytest tests -q -m gpu -s -k "small_ or row_owner or headline or registry or tps or inference or forward_vs or split or stress or fold" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | cut -c1-250 | tail -12
KFILE=k_gemm KPFX=GEMM bash scripts/micro/flash_variants.sh NOBURST > $O/build.log 2>&1; tail -1 $O/build.log
run_k() { echo "== $1 $2 $3" | tee -a $O/kbench.txt; timeout 300 python scripts/kbench.py $2 3 $3 2>&1 | grep -v parity | grep -v amdgpu | head -7 | tee -a $O/kbench.txt; }
run_b() { timeout 300 python bench.py --workload $2 --steps 8 --warmup 3 --no-extra --no-cpu-baseline --no-roofline $3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1 $2 $3', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
B1=tetrapeptide_fwdsim_crop4_T1000_B1; TP=tetrapeptide_tps_crop4_T100_B32
for rep in 1 2; do
  unset MDGEN_AMD_LIB; run_k product $B1; run_k product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_NOBURST.so; run_k NOBURST $B1; run_k NOBURST $TP
done
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; run_b product $B1; run_b product $TP
  export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_NOBURST.so; run_b NOBURST $B1; run_b NOBURST $TP
done
