#!/bin/bash
# round 6, call J: what the row-owner MLP loop's 43 cycles per MFMA are made of (experiment builds of the f16-hidden form: no accumulator
# re-arm / no accumulator reads / neither / short polynomial / no GELU at all): kbench per launch + s_memtime stamps of the main loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06j; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_rows KPFX=ROWS bash scripts/micro/flash_variants.sh NOREARM NOACCREAD NOACC NOPOLY NOGELU > $O/build.log 2>&1; tail -2 $O/build.log
C2=tetrapeptide_fwdsim_crop4_T1000_B16
run() { echo "== $1 $2" | tee -a $O/out.txt
  timeout 300 python scripts/kbench.py $C2 3 $2 2>&1 | grep "mlp@fold " | tee -a $O/out.txt
  timeout 300 python scripts/r06/tail_stamps.py $C2 $2 2>&1 | grep -A5 "plain folded" | tee -a $O/out.txt; }
for rep in 1 2; do
unset MDGEN_AMD_LIB; run product mlp_fold=1; run product mlp_fold=2
for v in NOREARM NOACCREAD NOACC NOPOLY NOGELU; do export MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_$v.so; run $v mlp_fold=2; done
done
