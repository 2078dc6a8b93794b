#!/bin/bash
# round 6, call AB: k16_linear_small with the contraction split over waves: training tests, then step time of the product against the
# NOSPLIT experiment build (KFILE=k_fp32 KPFX=LIN), three rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r06ab; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
KFILE=k_fp32 KPFX=LIN bash scripts/micro/flash_variants.sh NOSPLIT > $O/build.log 2>&1; tail -1 $O/build.log
timeout 1500 python -m pytest tests -x -q -m gpu -k "train or ddp or adam or rccl" 2>&1 | tail -4 | tee $O/pytest.log
for rep in 1 2 3; do
  unset MDGEN_AMD_LIB; echo "product $(timeout 300 python scripts/train_bench.py 1 250 256 10 16 2>&1 | tail -1)" | tee -a $O/train_ab.txt
  export MDGEN_AMD_LIB=$R/gpurun_out/dev_libs/libmdgen_amd_NOSPLIT.so; echo "NOSPLIT $(timeout 300 python scripts/train_bench.py 1 250 256 10 16 2>&1 | tail -1)" | tee -a $O/train_ab.txt
done
