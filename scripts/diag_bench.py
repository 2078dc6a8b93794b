import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
torch.set_grad_enabled(False)
dev = torch.device("cuda")
B, T, L = 16, 1000, 4
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=True, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
batch = bench.synth_batch(B, T, L, 0, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
for trial in range(3):
    outs = []
    for i in range(4):
        a, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
        outs.append((a, w.last_samples))
    torch.cuda.synchronize()
    for i, (a, s) in enumerate(outs):
        print(f"trial {trial} call {i}: samples nonfinite {int((~torch.isfinite(s)).sum())} atom14 nonfinite {int((~torch.isfinite(a)).sum())} equal_to_first {bool(torch.equal(a, outs[0][0]))}", flush=True)
