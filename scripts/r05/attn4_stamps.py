"""Dev tool (GPU): phase stamps of k_ln_qkv_attn4<true> (the L = 4 residue-axis sub-layer in one kernel) from the
`KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS` experiment build.

    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/r05/attn4_stamps.py [workload]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
from mdgen_amd._lib import lib

torch.set_grad_enabled(False)
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 else "tetrapeptide_fwdsim_crop4_T1000_B16"
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev)
w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
torch.cuda.synchronize()
host = np.zeros(8192 * 16, dtype=np.uint64)
fn = lib.mdgen_dev_attn4_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
s = host.reshape(-1, 16).astype(np.int64)
s = s[(s[:, 0] > 0) & (s[:, 10] > 0)]
names = ["LN prologue", "Q GEMM", "Q epilogue (bias, RoPE, stash)", "K GEMM", "K bias + RoPE", "scores + softmax", "V GEMM",
         "P V + panel stores + barrier", "out-projection GEMM", "barrier + residual epilogue"]
d = np.diff(s[:, :11], axis=1)
print(f"{wl}: {len(s)} waves stamped; wave lifetime {np.mean(s[:, 10] - s[:, 0]):.0f} cycles (the trunk's last launch of the kernel)")
for n, v in zip(names, d.mean(0)):
    print(f"  {n:34s} {v:8.0f}  ({100 * v / np.mean(s[:, 10] - s[:, 0]):4.1f} %)")
