#!/bin/bash
# round 5, call E: k_ln_qkv_attn4 phase stamps; stream count with the fused attention kernel; k_flash_proj at the small-N shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05e; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_proj or residue_axis_paths" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit" | tail -6
KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS > $O/build.log 2>&1; tail -1 $O/build.log
MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so timeout 300 python scripts/r05/attn4_stamps.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $O/attn4_stamps.txt
run() { timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
run --streams 1
run --streams 2
run --streams 3
run
for wl in tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  run --workload $wl
  run --workload $wl --option flash_proj=2
  run --workload $wl --option flash_proj=0
done
