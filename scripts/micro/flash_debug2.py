"""Dev tool (GPU): finiteness of a short rollout under both attention paths."""
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
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
if len(sys.argv) > 6:
    B, T, L, n_pad = [int(v) for v in sys.argv[3:7]]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True, num_layers=nl)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
for path in (1, 0):
    w.model.set_option("attention_path", path)
    for S in (1, 2, 3):
        a, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=False)
        smp = w.last_samples
        bad = ~torch.isfinite(smp)
        where = ""
        if bad.any():
            idx = bad.nonzero()
            where = f" frames {sorted(set(idx[:, 1].tolist()))[:6]}.. residues {sorted(set(idx[:, 2].tolist()))[:6]}.. n_frames {len(set(idx[:, 1].tolist()))} n_res {len(set(idx[:, 2].tolist()))}"
        print(f"B{B} T{T} L{L} pad{n_pad} layers {nl} path {path} S {S}: atom14 nonfinite {int((~torch.isfinite(a)).sum())}  samples nonfinite {int(bad.sum())} of {smp.numel()}{where}")
