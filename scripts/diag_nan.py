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
for k, v in batch.items():
    print(k, tuple(v.shape), "finite" if torch.isfinite(v.float()).all() else "NONFINITE", float(v.float().abs().max()))
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
prep = w.prep_batch(batch)
print("latents finite", bool(torch.isfinite(prep["latents"]).all()), float(prep["latents"].abs().max()))
for S in (1, 2, 5, 10, 20, 49):
    for g in (False, True):
        a, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=g)
        s = w.last_samples
        print(f"S={S} graph={g}: samples finite {bool(torch.isfinite(s).all())} max {float(s[torch.isfinite(s)].abs().max()):.3e} nonfinite {int((~torch.isfinite(s)).sum())}; atom14 nonfinite {int((~torch.isfinite(a)).sum())}", flush=True)
print("--- repeated graph replays (bench pattern)")
for i in range(5):
    a, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
    torch.cuda.synchronize()
    s = w.last_samples
    print(f"call {i}: samples nonfinite {int((~torch.isfinite(s)).sum())} atom14 nonfinite {int((~torch.isfinite(a)).sum())}", flush=True)
    if not torch.isfinite(a).all():
        bad = (~torch.isfinite(a)).nonzero()
        print("   first bad atom14 idx", bad[:5].tolist(), "samples at that token:", s[bad[0][0], bad[0][1], bad[0][2]].tolist())
