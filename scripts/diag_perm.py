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
for (B, T, L) in [(2, 64, 4), (3, 100, 4), (16, 1000, 4)]:
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
    perm = torch.arange(B - 1, -1, -1, device=dev)
    kwp = dict(t=t[perm], mask=mask[perm].contiguous(), start_frames=(R[perm].contiguous(), tr_[perm].contiguous()),
               x_cond=xc[perm].contiguous(), x_cond_mask=cm[perm].contiguous(), aatype=aat[perm].contiguous())
    for skip in (0, 7, 6, 5, 3):
        os.environ["MDGEN_DEBUG_SKIP"] = str(skip)
        y1, tr1 = m.forward(x, return_trace=True, **kw)
        tr1 = {k: v.clone() for k, v in tr1.items()}; y1 = y1.clone()
        yp, trp = m.forward(x[perm].contiguous(), return_trace=True, **kwp)
        torch.cuda.synchronize()
        def c(a, b):
            d = (a - b).abs(); return f"{float(d.max()):.1e}/{int((d > 0).sum())}"
        print(f"B{B} T{T} skip={skip}: ipa {c(trp['ipa_out'], tr1['ipa_out'][perm])} " +
              " ".join(f"h{i} {c(trp[f'h{i}'], tr1[f'h{i}'][perm])}" for i in range(6)) + f" out {c(yp, y1[perm])}", flush=True)
