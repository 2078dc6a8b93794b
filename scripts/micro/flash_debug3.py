"""Dev tool (GPU): where does the first non-finite value appear?  forward() with traces on the rollout's own inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
torch.set_grad_enabled(False)
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 else "atlas_crop256_T250_B1"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True, num_layers=nl)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
prep = w.prep_batch(batch)
kw = prep["model_kwargs"]
for path in (1, 0):
    w.model.set_option("attention_path", path)
    out, tr = w.model.forward(x=zs, t=torch.zeros(B, device=dev), return_trace=True, **kw)
    print(f"path {path}: out nonfinite {int((~torch.isfinite(out)).sum())}")
    for k, v in tr.items():
        bad = ~torch.isfinite(v)
        msg = ""
        if bad.any():
            idx = bad.nonzero()
            msg = f" first {idx[0].tolist()} last {idx[-1].tolist()} distinct dim-1 {sorted(set(idx[:, 1].tolist()))[:8]}"
        print(f"   {k}: nonfinite {int(bad.sum())} of {v.numel()}{msg}")
