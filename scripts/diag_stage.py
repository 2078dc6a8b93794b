import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden, weights_for, rel_l2
from mdgen_amd.model import LatentMDGenModel
torch.set_grad_enabled(False)
dev = torch.device("cuda")
g = load_golden("fwd_full_pep")
cfg, sd = weights_for(g)
m = LatentMDGenModel(cfg); m.load_state_dict(sd)
kw = dict(x=g["x"].to(dev), t=g["t"].to(dev), mask=g["mask"].to(dev), start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
          x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))
for skip in (7, 6, 5, 3, 0):
    os.environ["MDGEN_DEBUG_SKIP"] = str(skip)
    out, tr = m.forward(return_trace=True, **kw)
    print("skip", skip, {k: int((~torch.isfinite(v)).sum()) for k, v in tr.items()}, "out", int((~torch.isfinite(out)).sum()),
          "h0 rel", f"{rel_l2(tr['h0'].cpu(), g['h0']):.2e}" if torch.isfinite(tr['h0']).all() else "nan")
