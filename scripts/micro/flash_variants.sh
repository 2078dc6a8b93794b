#!/bin/bash
# Dev tool: build variant libraries of the attention kernel (compile-time experiment switches) next to the product
# library.  Usage (from the repo root, no GPU needed):  bash scripts/micro/flash_variants.sh NOLOAD [...]
# then on the GPU box:  MDGEN_AMD_LIB=scripts/micro/dev_libs/libmdgen_amd_NOLOAD.so python scripts/kbench.py ...
set -e
cd "$(dirname "$0")/../.."
python -m mdgen_amd.build >/dev/null
mkdir -p scripts/micro/dev_libs
for v in "$@"; do
  o=scripts/micro/dev_libs/k_flash_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -fno-honor-nans \
      -Wno-unused-function -Wno-pass-failed -DMDGEN_DEV_FLASH_$v -c mdgen_amd/csrc/k_flash.hip -o $o
  objs=$(ls mdgen_amd/build/*.o | grep -v k_flash.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/dev_libs/libmdgen_amd_$v.so $objs $o
  echo built scripts/micro/dev_libs/libmdgen_amd_$v.so
done
