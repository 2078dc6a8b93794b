"""GPU: k_chain_l4 phase stamps + kernel time for the library named by MDGEN_AMD_LIB (product, or experiment builds of
`KFILE=k_chain KPFX=CHAIN bash scripts/micro/flash_variants.sh NOSTORE NOEPI` -- values WRONG in those: timing only)."""
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
B, T, L, abs_pos, n_pad = bench.WORKLOADS["tetrapeptide_fwdsim_crop4_T1000_B16"]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1); w.model.set_option("chain_path", 2)   # the row-owner kernel (1 = the panel form)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
w.model.profile(True)
w.inference(batch, zs=zs, num_steps=3, use_graph=False)
rep = w.model.profile_report(); w.model.profile(False)
nw = (B * T * L + 31) // 32
buf = torch.zeros(nw * 10, dtype=torch.int64, device=dev)
w.model.phase_trace(buf)
w.inference(batch, zs=zs, num_steps=1, use_graph=False)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nw, 10).astype(np.int64); t = t[t[:, 0] != 0]
d = np.diff(t[:, :9], axis=1).mean(0)
names = ["prologue+LN", "residue qkv+attn (12 stages)", "out-proj (4)", "LN 2", "q_T (4)", "k_T (4)", "v_T (4)", "tail"]
k = "chain_L_qkvT"
print(f"{os.environ.get('MDGEN_AMD_LIB', 'product')}: {k} {rep[k]['ms'] / rep[k]['count'] * 1e3:.1f} us; flash_T {rep['flash_T']['ms'] / rep['flash_T']['count'] * 1e3:.1f} us; "
      + "; ".join(f"{n} {x:.0f}" for n, x in zip(names, d)) + f"; lifetime {(t[:, 8] - t[:, 0]).mean():.0f} cycles")
