"""Dev tool (GPU): phase stamps of k_mlp8 (the eight-wave panel MLP kernel of the small launches; split form at B = 1) from the
`KFILE=k_gemm KPFX=MLP8 bash scripts/micro/flash_variants.sh STAMPS` experiment build (stamps held in SGPRs, one store branch).
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/r06/mlp8_stamps.py [workload]"""
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
wl = sys.argv[1] if len(sys.argv) > 1 else "tetrapeptide_fwdsim_crop4_T1000_B1"
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
fn = lib.mdgen_dev_mlp8_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
s = host.reshape(-1, 16).astype(np.int64)
s = s[s[:, 0] > 0]
names = ["rows table + attention-output rows (PRE)", "out-projection GEMM (PRE)", "residual epilogue (PRE)", "LayerNorm prologue", "fc1(first chunk) + GELU",
         "chunk loop (fc1 || GELU, fc2)", "last fc2", "exchange in LDS", "partial stores + arrival (split)", "(last arriver) partial loads", "residual epilogue"]
last = s[s[:, 11] > 0]
print(f"{wl}: {len(s)} waves stamped, {len(last)} of them in a workgroup that ran the final epilogue")
d = np.diff(last[:, :12], axis=1)
print(f"  waves that finish a panel: lifetime {np.mean(last[:, 11] - last[:, 0]):.0f} cycles")
for n, v in zip(names, d.mean(0)):
    print(f"    {n:44s} {v:8.0f}  ({100 * v / np.mean(last[:, 11] - last[:, 0]):4.1f} %)")
other = s[(s[:, 11] == 0) & (s[:, 9] > 0)]
if len(other):
    print(f"  waves that leave after their partial: lifetime {np.mean(other[:, 9] - other[:, 0]):.0f} cycles")
