"""One-off diagnostic (GPU): where do split_sample 0 / 1 rollouts differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.model import LatentMDGenModel
from mdgen_amd.rigid_utils import Rotation
from mdgen_amd.synthetic import synth_state_dict
torch.set_grad_enabled(False)
dev = torch.device("cuda")
T, L, n_pad, B = 128, 256, 0, 1
cfg = ModelConfig.forward_sim(num_frames=T, crop=L)
sd = synth_state_dict(cfg, 4)
gen = torch.Generator().manual_seed(91 + T)
zs = torch.randn(B, T, L, 21, generator=gen).to(dev)
mask = torch.ones(B, T, L).to(dev)
q = torch.randn(B, L, 4, generator=gen)
R = Rotation(quats=(q / q.norm(dim=-1, keepdim=True)).to(dev)).get_rot_mats()
tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1).to(dev)
cm = torch.zeros(B, T, L, dtype=torch.long, device=dev); cm[:, 0] = 1
xc = torch.where(cm.unsqueeze(-1).bool(), torch.randn(B, T, L, 21, generator=gen).to(dev), torch.zeros((), device=dev))
aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
kw = dict(mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
rl = lambda a, b: float((a - b).norm() / b.norm())
for S in (1, 3):
    outs = {}
    for split in (0, 1):
        for extra in ({}, {"flash_proj": 0}):
            m = LatentMDGenModel(cfg); m.load_state_dict(sd); m.set_option("split_sample", split)
            for k, v in extra.items(): m.set_option(k, v)
            outs[split, tuple(extra)] = [m.sample_euler(zs, S, use_graph=g, **kw) for g in (False, False, True)]
            torch.cuda.synchronize(); del m
    ref = outs[0, ()][0]
    for key, o in outs.items():
        d = [rl(x, ref) for x in o]
        bad = (o[0] - ref).abs().amax(dim=(0, 3))   # [T, L] map of differences
        where = bad.nonzero()
        print(f"S={S} split={key[0]} opts={key[1]}: eager/eager/graph vs one-stream eager: {d}; differing (t,l): {len(where)}"
              + (f" t range {int(where[:,0].min())}-{int(where[:,0].max())} l range {int(where[:,1].min())}-{int(where[:,1].max())}" if len(where) else ""))
