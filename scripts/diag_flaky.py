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
def nf(x): return int((~torch.isfinite(x)).sum())
a, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
a2, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=True)
torch.cuda.synchronize()
bad = nf(a2)
print("graph: atom14 nonfinite", nf(a), bad, "samples nonfinite", nf(w.last_samples))
if bad or nf(w.last_samples):
    s = w.last_samples
    idx = (~torch.isfinite(s)).nonzero()
    print("  bad sample idx (first 10):", idx[:10].tolist(), " distinct b:", sorted(set(idx[:, 0].tolist())), "distinct l:", sorted(set(idx[:, 2].tolist())), "count", len(idx))
    e, _ = w.inference(batch, zs=zs, num_steps=49, use_graph=False)
    torch.cuda.synchronize()
    print("  eager rerun: atom14 nonfinite", nf(e), "samples", nf(w.last_samples))
    for S in (1, 2, 3, 5, 10):
        e, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=False)
        print(f"  eager S={S}: samples nonfinite {nf(w.last_samples)}")
    prep = w.prep_batch(batch)
    kw = dict(prep["model_kwargs"]); kw["mask"] = kw["mask"].contiguous(); kw["end_frames"] = None
    for tval in (0.0, 0.5, 0.98):
        out, tr = w.model.forward(zs, torch.full((B,), tval, device=dev), return_trace=True, **kw)
        print(f"  forward t={tval}: out nonfinite {nf(out)}", {k: nf(v) for k, v in tr.items()})
    sys.exit(3)
