#!/bin/bash
# round 4: 32-token workgroups of k_embed at small N: parity subset + per-kernel breakdown at the small-N shapes
mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or inference or cfg" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 3 --warmup 1 2>&1 | tail -1 > $O/$w.json
  python - $O/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["config"]["workload"], d["value"], "frames/s", d["ms_per_step"], "ms")
print("  ", d["roofline"]["by_kernel_ms_per_call"])
PY
done
