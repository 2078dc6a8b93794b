#!/bin/bash
# round 4: out-projection inside the PANEL MLP kernel (fuse_proj = 2): parity + A/B at cfg-2 and ATLAS
mkdir -p gpurun_out/r04j
timeout 600 python - > gpurun_out/r04j/parity.log 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden, weights_for, rel_l2
from mdgen_amd.model import LatentMDGenModel
dev = torch.device("cuda")
for name in ("fwd_full_pep", "fwd_full_atlas", "fwd_full_sim"):
    g = load_golden(name); cfg, sd = weights_for(g)
    kw = dict(x=g["x"].to(dev), t=g["t"].to(dev), mask=g["mask"].to(dev), start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
              x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))
    outs = {}
    for fp in (0, 2):
        m = LatentMDGenModel(cfg); m.load_state_dict(sd); m.set_option("fuse_proj", fp); m.set_option("mlp_path", 0)
        outs[fp] = m.forward(**kw).cpu()
    print(name, "fuse_proj 2 vs golden", rel_l2(outs[2], g["out"]), "vs fuse_proj 0", rel_l2(outs[2], outs[0]))
PY
cat gpurun_out/r04j/parity.log | grep -v amdgpu
run() {  n=$1; shift
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline "$@" > gpurun_out/r04j/$n.json 2> gpurun_out/r04j/$n.err
  python -c "
import json
d=json.load(open('gpurun_out/r04j/$n.json')); r=d['roofline'] or {}; print('$n', d['value'], d['ms_per_step'], {k:v for k,v in list((r.get('by_kernel_ms_per_call') or {}).items())[:7]})"
}
run c2_rows
run c2_panel --option mlp_path=0
run c2_panel_fp2 --option fuse_proj=2
run c2_rows_b
run c2_panel_fp2_b --option fuse_proj=2
run atlas --workload atlas_crop256_T250_B1
run atlas_fp2 --workload atlas_crop256_T250_B1 --option fuse_proj=2
