"""Dev tool (GPU): phase budget of k_ln_qkv<false> (LN prologue, the three GEMMs and their epilogues) from the
-DMDGEN_DEV_QKV_STAMPS experiment build:
    KFILE=k_gemm KPFX=QKV bash scripts/micro/flash_variants.sh STAMPS
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/micro/qkv_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.wrapper import NewMDGenWrapper
from mdgen_amd._lib import lib
torch.set_grad_enabled(False)
dev = torch.device("cuda")
wl = sys.argv[1] if len(sys.argv) > 1 else "tetrapeptide_fwdsim_crop4_T1000_B16"
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev); w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
torch.cuda.synchronize()
host = np.zeros(16384 * 8, dtype=np.uint64)
fn = lib.mdgen_dev_qkv_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
s = host.reshape(-1, 8).astype(np.int64)
s = s[s[:, 7] > 0]
names = ["LN prologue", "Q GEMM", "Q epilogue (bias, RoPE, fragment stores)", "K GEMM", "K epilogue", "V GEMM", "V epilogue"]
print(f"{wl}: {len(s)} waves; lifetime mean {np.mean(s[:, 7] - s[:, 0]):.0f} cycles")
for i, n in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print(f"  {n:44s} mean {d.mean():7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}")
