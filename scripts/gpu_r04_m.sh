#!/bin/bash
# round 4: small-N shapes after fuse_proj = 3 (default) and the 32-token k_embed; headline check
mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
for w in tetrapeptide_tps_crop4_T100_B32 tetrapeptide_fwdsim_crop4_T1000_B1 tetrapeptide_fwdsim_crop4_T1000_B16; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --no-roofline --steps 5 --warmup 2 2>&1 | tail -1 > $O/x.json
  python - $O/x.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["config"]["workload"], d["value"], "frames/s", d["ms_per_step"], "ms")
PY
done
bash scripts/gpu_r04_full.sh
