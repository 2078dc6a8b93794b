"""Dev tool (GPU): attention_path 0 vs 1 on one shape -- outputs, finiteness, per-kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
from mdgen_amd.model import LatentMDGenModel
torch.set_grad_enabled(False)
dev = torch.device("cuda")
B, T, L, n_pad = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1, 250, 256, 16))]
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
cfg = ModelConfig(crop=L, num_frames=T, num_layers=1, abs_pos_emb=False, sim_condition=True)
inp = synth_forward_inputs(cfg, B, T, L, n_pad, 11)
kw = dict(x=inp["x"].to(dev), t=inp["t"].to(dev), mask=inp["mask"].to(dev),
          start_frames=(inp["start_rot"].to(dev), inp["start_trans"].to(dev)), x_cond=inp["x_cond"].to(dev),
          x_cond_mask=inp["x_cond_mask"].to(dev), aatype=inp["aatype"].to(dev))
sd = synth_state_dict(cfg, 4)
for ax in ("mha_t", "mha_l"):
    for nm in ("q_proj", "k_proj"):
        sd[f"layers.0.{ax}.attn.{nm}.weight"] = sd[f"layers.0.{ax}.attn.{nm}.weight"] * scale
m = LatentMDGenModel(cfg)
m.load_state_dict(sd)
outs = []
for path in (0, 1):
    m.set_option("attention_path", path)
    m.forward(**kw)
    m.profile(True)
    o = m.forward(**kw).clone()
    rep = m.profile_report()
    m.profile(False)
    outs.append(o)
    print(f"path {path}: finite {bool(torch.isfinite(o).all())} nonfinite {int((~torch.isfinite(o)).sum())}",
          {k: round(v['ms'] / v['count'] * 1e3, 1) for k, v in rep.items() if 'flash' in k})
a, b = outs
ok = torch.isfinite(a) & torch.isfinite(b)
print("rel-L2 over finite entries:", float(((a - b)[ok]).norm() / b[ok].norm()))
