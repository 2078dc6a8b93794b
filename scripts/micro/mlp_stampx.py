"""Dev tool (GPU): where does an fc1 stage of k_mlp spend its time?  Needs the experiment build
    KFILE=k_gemm KPFX=MLP bash scripts/micro/flash_variants.sh STAMPX
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPX.so python scripts/micro/mlp_stampx.py
which adds three s_memtime stamps around / inside the fc1 stage of chunk 5: barrier exit, after k-step 11, stage end."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
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
w.model.set_option("streams", 1)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
nwg = (B * T * L + 63) // 64
buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device=dev)
w.model.phase_trace(buf)
w.inference(batch, zs=zs, num_steps=1, use_graph=False)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 4, 32).astype(np.int64)
# chunk 5: slot 10 = after Y(4) (2+2*4), 25 = X(5) start (after barrier), 30 = after k-step 11, 11 = after X(5)
bar = t[..., 25] - t[..., 10]; h1 = t[..., 30] - t[..., 25]; h2 = t[..., 11] - t[..., 30]
for name, d in (("barrier wait before X(5)", bar), ("X(5) k-steps 0-11", h1), ("X(5) k-steps 12-23", h2)):
    d = d.reshape(-1)
    print(f"{name:28s} mean {d.mean():7.0f} p10 {np.percentile(d,10):7.0f} p50 {np.percentile(d,50):7.0f} p90 {np.percentile(d,90):7.0f}")
