#!/bin/bash
# round 5, call A: the new parity tests (k_flash_proj, the 257..383-panel window, four- vs eight-wave panel kernels), then
# A/B of flash_proj (0 / occ 2 / occ 3) per kernel class (kbench, single stream, eager) and end to end (bench.py, graph, two streams)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05a; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "native_library or flash_proj or panel_kernels_257 or two_stream_views or row_owner_mlp_paths or forward_vs_reference_golden" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|vs separate\|exit" | tail -30
for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
  for o in "flash_proj=0" "flash_proj=1 flash_proj_occ=2" "flash_proj=1 flash_proj_occ=3"; do
    timeout 300 python scripts/kbench.py $wl 3 $o 2>&1 | grep -v "amdgpu.ids\|^parity" | head -12 | tee -a $O/kbench.txt
  done
done
for o in "flash_proj=0" "flash_proj=1 --option flash_proj_occ=2" "flash_proj=1 --option flash_proj_occ=3"; do
  for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
    timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-extra --no-cpu-baseline --option $o 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$wl', '$o', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_call'] if d.get('roofline') else None)" | tee -a $O/bench_ab.txt
  done
done
