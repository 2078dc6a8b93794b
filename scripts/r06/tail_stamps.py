"""GPU: phase stamps of the row-owner MLP launch that carries both tails (FinalLayer + Euler update, next step's embedding):
s_memtime deltas per wave between  start | LN prologue | pipeline fill | 22 iterations | drain | FinalLayer tail | embedding | store.
    python scripts/r06/tail_stamps.py [workload] [option=value ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
torch.set_grad_enabled(False)
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "tetrapeptide_fwdsim_crop4_T1000_B16"
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
for kv in sys.argv[1:]:
    if "=" in kv:
        w.model.set_option(kv.split("=")[0], int(kv.split("=")[1]))
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
for tail in (1, 0):
    w.model.set_option("trace_tail", tail)
    nw = (B * T * L + 31) // 32
    buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
    w.model.phase_trace(buf)
    w.inference(batch, zs=zs, num_steps=2, use_graph=False)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nw, 8).astype(np.int64)
    t = t[t[:, 0] != 0]
    if tail:   # stamps 0 1 2 3 4 6 7 5
        order, names = [0, 1, 2, 3, 4, 6, 7, 5], ["LN prologue", "fill", "22 iterations", "drain", "FinalLayer tail", "embedding", "store"]
    else:      # first trunk MLP launch (no tail): 0 1 2 3 4 5
        order, names = [0, 1, 2, 3, 4, 5], ["LN prologue", "fill", "22 iterations", "drain", "store epilogue"]
    print(f"{wl}: {'launch with both tails' if tail else 'plain folded launch'}: {len(t)} waves, lifetime mean {np.mean(t[:, 5] - t[:, 0]):.0f} ticks")
    for a, b, nm in zip(order[:-1], order[1:], names):
        d = t[:, b] - t[:, a]
        print(f"   {nm:16s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
