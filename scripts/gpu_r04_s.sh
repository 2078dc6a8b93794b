#!/bin/bash
# round 4: k_mlp8 (eight waves per panel for launches of <= 256 panels): parity subset + small-N bench lines
mkdir -p gpurun_out/r04s; O=gpurun_out/r04s
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or mlp_paths or inference_sim or padding" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>&1 | tail -1 > $O/x.json
  python - $O/x.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
k = d["roofline"]["by_kernel_ms_per_call"]
print(d["config"]["workload"], d["value"], "frames/s", d["ms_per_step"], "ms;", {a: b for a, b in list(k.items())[:5]})
PY
done
