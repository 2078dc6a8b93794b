"""Dev tool (GPU): per-wave cycle stamps of the attention kernel from the -DMDGEN_DEV_FLASH_STAMPS experiment build.

    bash scripts/micro/flash_variants.sh STAMPS
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/micro/flash_stamps.py [workload]

Prints the shader clock actually sustained inside the kernel (s_memtime ticks per s_memrealtime tick, the latter a
constant 100 MHz), the cycles one wave spends per (32-key, 32-query) pair in the loop, and the prologue / epilogue."""
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
nseq, ln = B * L, T                       # the last k_flash launch of a step is the temporal attention of the last layer
nqc, nt = (ln + 63) // 64, ln // 32 + 1
nwaves = ((nseq * 4 + 7) // 8) * nqc * 8 * 4
n = min(nwaves, 32768)
host = np.zeros(32768 * 16, dtype=np.uint64)
fn = lib.mdgen_dev_flash_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
s = host.reshape(-1, 16)[:n].astype(np.int64)
s = s[s[:, 0] > 0]
cyc, real = s[:, 3] - s[:, 0], s[:, 5] - s[:, 4]
span_real = s[:, 5].max() - s[:, 4].min()
mhz = cyc.sum() / real.sum() * 100.0
pairs = 2 * nt
loop = s[:, 2] - s[:, 1]
fast = s[:, 6] == 0
print(f"{wl}: {len(s)} waves stamped; robust loop: {int((s[:, 6] & 1).sum())} from the start (first tile masked), "
      f"{int((s[:, 6] == 2).sum())} after an overflow of the fixed anchor; kernel span {span_real / 100.0:.1f} us")
if (s[:, 6] == 2).any():
    bad = np.nonzero(s[:, 6] == 2)[0]
    print("  overflowed waves (index = workgroup*4 + wave):", bad[:16], "... workgroups", np.unique(bad // 4)[:16])
    f32 = lambda x: np.array(x, dtype=np.uint32).view(np.float32)
    for i in bad[:6]:
        print(f"    wave {i}: fixed anchor M {f32(s[i, 8])[()]:.2f}  final robust shift {f32(s[i, 11])[()]:.2f}  bad-lane mask {int(s[i, 9]) & 0xffffffffffffffff:016x} "
              f"l(A) {f32(s[i, 10] & 0xffffffff)[()]:.3e} l(B) {f32((s[i, 10] >> 32) & 0xffffffff)[()]:.3e}")
cyc, real = s[:, 3] - s[:, 0], s[:, 5] - s[:, 4]
span_real = s[:, 5].max() - s[:, 4].min()
mhz = cyc.sum() / real.sum() * 100.0
pairs = 2 * nt
loop = s[:, 2] - s[:, 1]
fast = s[:, 6] == 0
print(f"{wl}: {len(s)} waves stamped; robust loop: {int((s[:, 6] & 1).sum())} from the start (first tile masked), "
      f"{int((s[:, 6] == 2).sum())} after an overflow of the fixed anchor; kernel span {span_real / 100.0:.1f} us")
if (s[:, 6] == 2).any():
    bad = np.nonzero(s[:, 6] == 2)[0]
    print("  overflowed waves (index = workgroup*4 + wave):", bad[:16], "... workgroups", np.unique(bad // 4)[:16])
    f32 = lambda x: np.array(x, dtype=np.uint32).view(np.float32)
    for i in bad[:6]:
        print(f"    wave {i}: fixed anchor M {f32(s[i, 8])[()]:.2f}  final robust shift {f32(s[i, 11])[()]:.2f}  bad-lane mask {int(s[i, 9]) & 0xffffffffffffffff:016x} "
              f"l(A) {f32(s[i, 10] & 0xffffffff)[()]:.3e} l(B) {f32((s[i, 10] >> 32) & 0xffffffff)[()]:.3e}  "
              f"s0[0] {f32(s[i, 12] & 0xffffffff)[()]:.3e} s0[5] {f32((s[i, 12] >> 32) & 0xffffffff)[()]:.3e} lane max {f32(s[i, 13] & 0xffffffff)[()]:.3e} "
              f"row max {f32(s[i, 14] & 0xffffffff)[()]:.3e} vm0 {(int(s[i, 14]) >> 32) & 0xffffffff:08x} len {(int(s[i, 13]) >> 32) & 0xffff} qc {(int(s[i, 13]) >> 48) & 0xffff} "
              f"q0[0] {int(s[i, 15]) & 0xffffffff:08x} q1[2] {(int(s[i, 15]) >> 32) & 0xffffffff:08x} k0[0] {int(s[i, 7]) & 0xffffffff:08x} k1[2] {(int(s[i, 7]) >> 32) & 0xffffffff:08x}")
print(f"  sustained shader clock inside the kernel: {mhz:.0f} MHz")
if not fast.any():
    fast[:] = True
print(f"  wave lifetime {cyc.mean():.0f} cycles; prologue {np.mean(s[:, 1] - s[:, 0]):.0f}; loop {loop[fast].mean():.0f} "
      f"= {loop[fast].mean() / pairs:.1f} cycles per pair per wave ({pairs} pairs); epilogue n/a")
order = np.argsort(s[:, 0])
for name, idx in (("first 3072 waves (full occupancy)", order[:3072]), ("last 1024 waves", order[-1024:])):
    f = idx[s[idx, 6] == 0]
    print(f"  {name}: loop {np.mean(loop[f]) / pairs:.1f} cycles per pair per wave -> x{3 if 'first' in name else 1} waves/SIMD")
