"""GPU: per-phase timeline of the warp-specialised MLP kernel (second panel of every workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
torch.set_grad_enabled(False)
dev = torch.device("cuda")
B, T, L, abs_pos, n_pad = bench.WORKLOADS["tetrapeptide_fwdsim_crop4_T1000_B16"]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
os.environ["MDGEN_DUAL_STREAM"] = "0"
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
nwg = 256
buf = torch.zeros(nwg * 8 * 32, dtype=torch.int64, device=dev)
w.model.phase_trace(buf)
w.inference(batch, zs=zs, num_steps=1, use_graph=False)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8, 32).astype(np.int64)
ok = t[:, 0, 25] > 0
t = t[ok]
print(f"workgroups traced {ok.sum()}")
A, Bw = t[:, :4], t[:, 4:]
def rep(name, x, labels):
    prev = x[..., 0]
    out = []
    for c in range(6):
        for j, lb in enumerate(labels):
            cur = x[..., 1 + 4 * c + j]
            out.append((f"c{c}.{lb}", (cur - prev).reshape(-1)))
            prev = cur
    agg = {}
    for nm, d in out:
        agg.setdefault(nm.split(".")[1], []).append(d)
    print(name, "panel total", (x[..., 25] - x[..., 0]).mean())
    for kname, ds in agg.items():
        d = np.stack(ds)   # [6 chunks][waves]
        print(f"   {kname:10s} per chunk mean " + " ".join(f"{v:7.0f}" for v in d.mean(1)) + f"   | sum {d.mean(1).sum():8.0f}")
rep("A (fc1+GELU)", A, ["fc1", "gelu", "waitX", "write+Y"])
rep("B (fc2+slots)", Bw, ["slot", "waitX", "waitY", "fc2"])
