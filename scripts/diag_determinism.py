"""GPU diagnostic: run the same forward 3x with traces and report where results first differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.model import LatentMDGenModel
from mdgen_amd.rigid_utils import Rotation
torch.set_grad_enabled(False)
dev = torch.device("cuda")
cfg = ModelConfig.forward_sim(num_frames=1000, crop=4)
m = LatentMDGenModel(cfg); m.load_state_dict(synth_state_dict(cfg, 0))
for (B, T, L) in [(1, 200, 4), (4, 1000, 4), (16, 1000, 4)]:
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, L, 21, generator=gen).to(dev)
    t = torch.full((B,), 0.3, device=dev)
    mask = torch.ones(B, T, L, device=dev)
    q = torch.randn(B, L, 4, generator=gen); q = q / q.norm(dim=-1, keepdim=True)
    R = Rotation(quats=q.to(dev)).get_rot_mats()
    tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1).to(dev)
    cm = torch.zeros(B, T, L, dtype=torch.long, device=dev); cm[:, 0] = 1
    xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, 21, generator=gen).to(dev), torch.zeros((), device=dev))
    aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
    kw = dict(t=t, mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
    for skip in (0, 6, 5, 3, 7):
        os.environ["MDGEN_DEBUG_SKIP"] = str(skip)
        runs = []
        for i in range(3):
            out, tr = m.forward(x, return_trace=True, **kw)
            torch.cuda.synchronize()
            runs.append((out.clone(), {k: v.clone() for k, v in tr.items()}))
        def cmp(a, b):
            d = (a - b).abs()
            return f"{float(d.max()):.2e}/{int((d > 0).sum())}"
        keys = ["ipa_out"] + [f"h{i}" for i in range(6)]
        print(f"B{B} T{T} L{L} skip={skip}: " + " ".join(f"{k}:{cmp(runs[0][1][k], runs[1][1][k])}|{cmp(runs[1][1][k], runs[2][1][k])}" for k in keys)
              + f" out:{cmp(runs[0][0], runs[1][0])}|{cmp(runs[1][0], runs[2][0])}", flush=True)
        if skip == 0:
            d = (runs[0][1]["h1"] - runs[1][1]["h1"]).abs()
            nz = d.reshape(B, T, L, -1).amax(-1).nonzero()
            print("   first differing (b,t,l):", nz[:8].tolist(), "count", len(nz))
os.environ["MDGEN_DEBUG_SKIP"] = "0"
