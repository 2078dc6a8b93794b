#!/bin/bash
# round 5, call L: k_flash_proj8 (eight waves, 128-row panel, four query tiles per wave): parity, A/B against k_flash_proj
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$PWD; O=$R/gpurun_out/r05l; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "flash_proj or full_size_properties or forward_cfg4_full" > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
grep -v "amdgpu.ids" $O/pytest.log | grep "passed\|failed\|Error\|error\|assert\|exit\|128-row" | tail -12
for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
  for o in "flash_proj_form=4" "flash_proj_form=8" "flash_proj_form=4" "flash_proj_form=8"; do
    timeout 300 python scripts/kbench.py $wl 3 $o 2>&1 | grep "S=3\|flash" | tee -a $O/kbench.txt
  done
done
run() { timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt; }
for i in 1 2; do for wl in tetrapeptide_fwdsim_crop4_T1000_B16 atlas_crop256_T250_B1; do
  run --workload $wl --option flash_proj_form=4
  run --workload $wl --option flash_proj_form=8
done; done
