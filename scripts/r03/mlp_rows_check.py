"""GPU: row-owner MLP kernel (k_mlp_rows) vs the panel kernel (k_mlp): parity against the reference goldens with each,
per-kernel-class hipEvent timings at cfg-2 / cfg-4, and the in-kernel phase stamps of k_mlp_rows (dev loop helper)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from conftest import load_golden, weights_for, rel_l2
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
from mdgen_amd.model import LatentMDGenModel
torch.set_grad_enabled(False)
dev = torch.device("cuda")

for name in ("fwd_full_pep", "fwd_full_atlas", "fwd_tiny_sim"):
    g = load_golden(name)
    cfg, sd = weights_for(g)
    if cfg.embed_dim != 384:
        continue
    outs = {}
    for path in (0, 2):
        m = LatentMDGenModel(cfg); m.load_state_dict(sd); m.set_option("mlp_path", path)
        out = m.forward(x=g["x"].to(dev), t=g["t"].to(dev), mask=g["mask"].to(dev),
                        start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
                        x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))
        outs[path] = out.cpu()
        print(f"parity {name} mlp_path={path}: rel-L2 vs reference {rel_l2(out.cpu(), g['out']):.3e}  finite={bool(torch.isfinite(out).all())}", flush=True)
        del m
    print(f"   rows vs panel: rel-L2 {rel_l2(outs[2], outs[0]):.3e}", flush=True)

def timing(wl, S, path, streams=1, trace=False):
    B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
    cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
    w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
    w.model.set_option("streams", streams); w.model.set_option("mlp_path", path)
    batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
    zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
    a0, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=False)
    w.model.profile(True)
    a, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=False)
    rep = w.model.profile_report()
    w.model.profile(False)
    tot = sum(v["ms"] for v in rep.values())
    print(f"{wl} S={S} mlp_path={path} streams={streams}: total event ms {tot:.2f} -> per NFE {tot / S:.3f} ms; finite={bool(torch.isfinite(a).all())}")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        fl = bench.algorithmic_flops(k, B, T, L)
        us = v["ms"] / v["count"] * 1e3
        tf = f"{fl / (us * 1e-6) / 1e12:7.1f} TF" if fl else ""
        print(f"  {k:16s} n={v['count']:4d} avg {us:9.1f} us  {100 * v['ms'] / tot:5.1f}%  {tf}")
    if trace:
        nw = (B * T * L + 31) // 32
        buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
        w.model.phase_trace(buf)
        w.inference(batch, zs=zs, num_steps=1, use_graph=False)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(nw, 8).astype(np.int64)
        t = t[t[:, 0] != 0]
        names = ["(proj +) LN", "P0+P1", "22 iterations", "E0+E1", "epilogue"]
        t0 = t[:, 0].min()
        print(f"  k_mlp_rows stamps over {len(t)} waves; kernel span {t[:, 5].max() - t0} ticks (100 MHz s_memtime)")
        for i, nm in enumerate(names):
            d = t[:, i + 1] - t[:, i]
            print(f"    {nm:14s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} ticks")
        if t[:, 6].any():
            print(f"    fused out-projection: rows + 24 blocks {np.mean(t[:, 6] - t[:, 0]):8.0f}  residual epilogue (keep) {np.mean(t[:, 7] - t[:, 6]):8.0f}  LayerNorm from registers {np.mean(t[:, 1] - t[:, 7]):8.0f} ticks")
        start = t[:, 0] - t0
        print(f"    wave start: first round <= {np.percentile(start, 45):.0f}, later median {np.percentile(start, 80):.0f}; lifetime mean {(t[:, 5] - t[:, 0]).mean():.0f}")
        np.save(os.path.join(ROOT, "gpurun_out", "mlp_rows_trace.npy"), t)
    return a

a0 = timing("tetrapeptide_fwdsim_crop4_T1000_B16", 3, 0)
a1 = timing("tetrapeptide_fwdsim_crop4_T1000_B16", 3, 1, trace=True)
d = (a1 - a0).float()
print(f"cfg-2 S=3 atom14 rows vs panel: rms {d.pow(2).mean().sqrt():.4f} A  max {d.abs().max():.4f} A")
timing("atlas_crop256_T250_B1", 3, 0)
timing("atlas_crop256_T250_B1", 3, 1)
