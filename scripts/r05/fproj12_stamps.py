"""Dev tool (GPU): per-wave cycle stamps + HW_ID placement of k_flash_proj12 from the -DMDGEN_DEV_FLASH_STAMPS experiment build.

    bash scripts/micro/flash_variants.sh STAMPS
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/r05/fproj12_stamps.py [workload]

Rows of twelve waves per workgroup: 0 start, 1..3 after pass 0..2, 5 after the barrier, 6 after the GEMM, 7 end (s_memtime); 8 / 9
s_memrealtime; 10 HW_ID (wave slot 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13).  Prints how the waves of a workgroup were placed on
the four SIMDs and how many workgroups shared a CU."""
import ctypes as C
import os
import sys
from collections import Counter
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
for k, v in (("streams", 1), ("flash_proj", 2), ("flash_proj_form", 12)):
    w.model.set_option(k, v)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
torch.cuda.synchronize()
host = np.zeros(8192 * 16, dtype=np.uint64)
fn = lib.mdgen_dev_fproj_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
a = host.reshape(-1, 16).astype(np.int64)
n_wg = int((a[:, 0] > 0).sum()) // 12
s = a[:n_wg * 12].reshape(n_wg, 12, 16)
life = s[:, :, 7] - s[:, :, 0]
mhz = life.sum() / (s[:, :, 9] - s[:, :, 8]).sum() * 100.0
t0 = s[:, :, 8].min()
print(f"{wl}: {n_wg} workgroups stamped, launch span {(s[:, :, 9].max() - t0) / 100.0:.1f} us, shader clock {mhz:.0f} MHz")
print(f"  wave lifetime {life.mean():.0f} cycles; passes (waves 0..7) {[float((s[:, :8, 1 + k] - s[:, :8, k]).mean().round()) for k in range(3)]}, "
      f"(waves 8..11) {[float((s[:, 8:, 1 + k] - s[:, 8:, k]).mean().round()) for k in range(3)]}")
print(f"  barrier wait {np.mean(s[:, :, 5] - s[:, :, 3]):.0f}; out-projection GEMM {np.mean(s[:, :, 6] - s[:, :, 5]):.0f}; "
      f"barrier + residual epilogue {np.mean(s[:, :, 7] - s[:, :, 6]):.0f}")
hw = s[:, :, 10]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
pat = Counter(tuple(np.bincount(simd[i], minlength=4).tolist()) for i in range(n_wg))
print(f"  waves of a workgroup per SIMD (0..3): {pat.most_common(6)}")
start = (s[:, 0, 8] - t0) / 100.0
end = (s[:, :, 9].max(1) - t0) / 100.0
print(f"  workgroup start (us): p10 {np.percentile(start, 10):.1f} p25 {np.percentile(start, 25):.1f} p50 {np.percentile(start, 50):.1f} "
      f"p75 {np.percentile(start, 75):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}; end: p50 {np.percentile(end, 50):.1f} max {end.max():.1f}")
# workgroups resident together on a CU: key = (XCD = block % 8, SE, SH, CU)
key = [(i % 8, int(se[i, 0]), int(sh[i, 0]), int(cu[i, 0])) for i in range(n_wg)]
cus = Counter(key)
print(f"  {len(cus)} distinct CUs; workgroups per CU over the launch: {Counter(cus.values()).most_common(6)}")
first = start < 1.0
per_cu_first = Counter(k for k, f in zip(key, first) if f)
print(f"  first round (started within 1 us): {int(first.sum())} workgroups on {len(per_cu_first)} CUs, per CU {Counter(per_cu_first.values()).most_common(4)}")
# SIMD load of the first round per CU
load = Counter()
for i in range(n_wg):
    if first[i]:
        for sd in simd[i]:
            load[(key[i], int(sd))] += 1
print(f"  first round: waves per (CU, SIMD): {Counter(load.values()).most_common(6)}")
