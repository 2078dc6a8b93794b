"""GPU: per-phase timeline of the fused MLP kernel from in-kernel s_memtime stamps (dev loop helper).

Prints, per phase, the mean/percentiles of its duration over all waves, the spread between the 4 waves of a
workgroup at each barrier, and the workgroup lifetime for the first and second dispatch round."""
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
wl = sys.argv[1] if len(sys.argv) > 1 else "tetrapeptide_fwdsim_crop4_T1000_B16"
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.model.set_option("streams", 1)
for kv in sys.argv[2:]:   # library options, name=value (e.g. mlp_path=0 fuse_proj=0)
    k_, v_ = kv.split("=")
    w.model.set_option(k_, int(v_))
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
nwg = (B * T * L + 63) // 64
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device=dev)
w.model.phase_trace(buf)
w.inference(batch, zs=zs, num_steps=1, use_graph=False)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 4, 32).astype(np.int64)
t0 = t[..., 0].min()
rel = t[..., :28] - t0
names = ["prologue", "fc1(0)+gelu(0)"]
slots = [1, 2]
for c in range(1, 12):
    names += [f"c{c}.barrier+fc1", f"c{c}.gelu||fc2"]
    slots += [1 + 2 * c, 2 + 2 * c]
names += ["fc2(11)", "tail-barriers", "epilogue"]
slots += [24, 26, 27]
prev = 0
print(f"workgroups {nwg}; kernel span {rel[..., 27].max()} ticks (s_memtime)")
agg = {}
for nm, s in zip(names, slots):
    d = (t[..., s] - t[..., prev]).reshape(-1)
    key = nm.split(".")[-1] if "." in nm else nm
    agg.setdefault(key, []).append(d)
    prev = s
tot = 0
for k, ds in agg.items():
    d = np.concatenate(ds) if len(ds) > 1 else ds[0]
    per_wg = sum(x.mean() for x in ds)
    tot += per_wg
    print(f"  {k:14s} per-instance mean {d.mean():9.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}   sum per WG {per_wg:9.0f}")
life = (t[..., 27].max(1) - t[..., 0].min(1))
start = t[..., 0].min(1) - t0
order = np.argsort(start)
first = order[: min(512, nwg)]
second = order[min(512, nwg):]
print(f"  WG lifetime: all {life.mean():.0f}  first-round {life[first].mean():.0f}  later {life[second].mean() if len(second) else 0:.0f}; sum of phases {tot:.0f}")
print(f"  start spread first round: {start[first].min()}..{start[first].max()}   later starts: {start[second].min() if len(second) else 0}..{start[second].max() if len(second) else 0}")
hw = t[..., 28]; xcc = t[..., 29]
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | ((xcc & 0xf) << 8)
print(f"  distinct CU ids seen: {len(np.unique(cu[:, 0]))}")
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mlp_trace.npy"), t)
