"""GPU diagnostic 2: one-layer model, compare workspace buffers (q/k/v operands, attention output) across runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.model import LatentMDGenModel
from mdgen_amd.rigid_utils import Rotation
torch.set_grad_enabled(False)
dev = torch.device("cuda")
cfg = ModelConfig(num_layers=1, crop=4, num_frames=1000)
m = LatentMDGenModel(cfg); m.load_state_dict(synth_state_dict(cfg, 0))
B, T, L = 16, 1000, 4
gen = torch.Generator().manual_seed(9)
x = torch.randn(B, T, L, 21, generator=gen).to(dev)
t = torch.full((B,), 0.3, device=dev)
mask = torch.ones(B, T, L, device=dev)
q = torch.randn(B, L, 4, generator=gen); q = q / q.norm(dim=-1, keepdim=True)
R = Rotation(quats=q.to(dev)).get_rot_mats()
tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1).to(dev)
cm = torch.zeros(B, T, L, dtype=torch.long, device=dev); cm[:, 0] = 1
xc = torch.zeros(B, T, L, 21, device=dev)
aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
kw = dict(t=t, mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
lay = m.workspace_layout(B, T, L, 1, False)
N = B * T * L
regions = {"qf": (lay.qf, lay.kf - lay.qf), "kf": (lay.kf, lay.vf - lay.kf), "vf": (lay.vf, lay.obuf - lay.vf),
           "obuf": (lay.obuf, N * 384 * 2), "h": (lay.h, N * 384 * 4), "mod": (lay.mod, lay.silu_t - lay.mod)}
for skip in (6, 5):
    os.environ["MDGEN_DEBUG_SKIP"] = str(skip)
    snaps = []
    for i in range(3):
        out = m.forward(x, **kw)
        torch.cuda.synchronize()
        ws = m._ws[(B, T, L, 1, 0)]
        snaps.append({k: ws[o:o + n].clone() for k, (o, n) in regions.items()})
        snaps[-1]["out"] = out.clone().view(torch.uint8).flatten()
    for k in ("qf", "kf"):
        a, b = snaps[0][k].view(torch.bfloat16).float(), snaps[1][k].view(torch.bfloat16).float()
        d = (a - b).abs()
        nz = d > 0
        if nz.any():
            rel = d[nz] / (a[nz].abs() + 1e-30)
            print(f"   {k}: {int(nz.sum())} bf16 elements differ; max abs {float(d.max()):.3e}; rel max {float(rel.max()):.3e} median {float(rel.median()):.3e}", flush=True)
    for k in list(regions) + ["out"]:
        a, b, c = snaps[0][k], snaps[1][k], snaps[2][k]
        d01 = (a != b).nonzero().flatten()
        d12 = (b != c).nonzero().flatten()
        print(f"skip={skip} {k:5s} bytes={a.numel()} diff01={len(d01)} diff12={len(d12)}",
              (f"first byte offsets {d01[:6].tolist()} last {d01[-3:].tolist()}" if len(d01) else ""), flush=True)
