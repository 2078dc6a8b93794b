#!/bin/bash
# Dev tool: build variant libraries of one kernel file (compile-time experiment switches, -DMDGEN_DEV_<FILE>_<NAME>) next
# to the product library.  Usage (from the repo root, no GPU needed):  bash scripts/micro/flash_variants.sh NOLOAD [...]
# (attention kernel, -DMDGEN_DEV_FLASH_<NAME>; run it inside the gpurun command -- hipcc is on the box)  or  KFILE=k_gemm KPFX=MLP bash scripts/micro/flash_variants.sh HALFW
# then:  MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_NOLOAD.so python scripts/kbench.py ...
# Experiment libraries are built ON the GPU box, on demand, into gpurun_out/ (scratch; never part of the snapshot a lease receives).
set -e
cd "$(dirname "$0")/../.."
python -m mdgen_amd.build >/dev/null
mkdir -p gpurun_out/dev_libs
KFILE=${KFILE:-k_flash}; KPFX=${KPFX:-FLASH}
EXTRA=""; [ $KFILE = k_flash ] && EXTRA="-fno-honor-nans"
# api.hip of every variant library reports mdgen_dev_build() = 1 (csrc/dev.h)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function \
    -Wno-pass-failed -DMDGEN_DEV_BUILD -c mdgen_amd/csrc/api.hip -o gpurun_out/dev_libs/api_dev.o
for v in "$@"; do
  o=gpurun_out/dev_libs/${KFILE}_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize $EXTRA \
      -Wno-unused-function -Wno-pass-failed -DMDGEN_DEV_BUILD -DMDGEN_DEV_${KPFX}_$v -c mdgen_amd/csrc/$KFILE.hip -o $o
  objs=$(ls mdgen_amd/build/*.o | grep -v "/$KFILE.o" | grep -v "/api.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/dev_libs/libmdgen_amd_$v.so $objs gpurun_out/dev_libs/api_dev.o $o
  echo built gpurun_out/dev_libs/libmdgen_amd_$v.so
done
