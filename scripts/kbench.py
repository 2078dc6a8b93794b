"""GPU: per-kernel-class hipEvent timings of one Euler rollout + a quick parity check (dev loop helper).
    python scripts/kbench.py [workload] [S] [option=value ...]      (library options, e.g. flash_proj=0)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from conftest import load_golden, weights_for, rel_l2
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
from mdgen_amd.model import LatentMDGenModel
torch.set_grad_enabled(False)
dev = torch.device("cuda")
# parity quick check vs the reference golden (micro path + flash path)
for name in ("fwd_full_pep", "fwd_full_atlas"):
    g = load_golden(name)
    cfg, sd = weights_for(g)
    m = LatentMDGenModel(cfg); m.load_state_dict(sd)
    out = m.forward(x=g["x"].to(dev), t=g["t"].to(dev), mask=g["mask"].to(dev),
                    start_frames=(g["start_rot"].to(dev), g["start_trans"].to(dev)),
                    x_cond=g["x_cond"].to(dev), x_cond_mask=g["x_cond_mask"].to(dev), aatype=g["aatype"].to(dev))
    print(f"parity {name}: rel-L2 {rel_l2(out.cpu(), g['out']):.3e}", flush=True)
    del m
wl = sys.argv[1] if len(sys.argv) > 1 else "tetrapeptide_fwdsim_crop4_T1000_B16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5
OPTS = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)   # full-batch launches on one stream (what the profiling leg measures)
for k, v in OPTS.items():
    w.model.set_option(k, v)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=S, use_graph=False)
w.model.profile(True)
a, _ = w.inference(batch, zs=zs, num_steps=S, use_graph=False)
rep = w.model.profile_report()
w.model.profile(False)
assert torch.isfinite(a).all()
tot = sum(v["ms"] for v in rep.values())
print(f"{wl} S={S} {OPTS}: total event ms {tot:.2f} -> per NFE {tot / S:.3f} ms")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    fl = bench.algorithmic_flops(k, B, T, L)
    us = v["ms"] / v["count"] * 1e3
    tf = f"{fl / (us * 1e-6) / 1e12:7.1f} TF" if fl else ""
    print(f"  {k:16s} n={v['count']:4d} avg {us:9.1f} us  {100 * v['ms'] / tot:5.1f}%  {tf}")
