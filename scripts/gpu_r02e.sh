#!/bin/bash
# round-2 GPU call E: attention kernel with precomputed key-validity words: stamps, timings, full parity suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=$R/gpurun_out
(MDGEN_AMD_LIB=$R/scripts/micro/dev_libs/libmdgen_amd_STAMPS.so python scripts/micro/flash_stamps.py; MDGEN_AMD_LIB=$R/scripts/micro/dev_libs/libmdgen_amd_STAMPS.so python scripts/micro/flash_stamps.py atlas_crop256_T250_B1) 2>&1 | grep -v "amdgpu.ids\|Warning\|ret =\|return _m" > $O/flash_stamps.txt
cat $O/flash_stamps.txt
for i in 1 2; do python scripts/kbench.py tetrapeptide_fwdsim_crop4_T1000_B16 2 2>&1 | grep -E "flash_T|mlp  |ln_qkv_T"; done
python scripts/kbench.py atlas_crop256_T250_B1 2 2>&1 | grep -E "parity|flash|mlp  |ln_qkv"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "attention_fixed or poisoned or shapes or golden" 2>&1 | grep -v amdgpu.ids | tail -12 > $O/pytest_gpu_e.log
tail -5 $O/pytest_gpu_e.log
echo "== NOFALLBACK variant on the attention-path test (expected: non-finite at the overflowing scales)"
MDGEN_AMD_LIB=$R/scripts/micro/dev_libs/libmdgen_amd_NOFALLBACK.so timeout 300 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider -k "attention_fixed" 2>&1 | grep -v amdgpu.ids | grep -E "attention loops|assert|passed|failed" | head
