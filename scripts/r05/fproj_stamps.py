"""Dev tool (GPU): per-wave cycle stamps of k_flash_proj from the -DMDGEN_DEV_FLASH_STAMPS experiment build.

    bash scripts/micro/flash_variants.sh STAMPS
    MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so python scripts/r05/fproj_stamps.py [workload] [option=value ...]

Per wave: the four head-group jobs, the barrier wait, the out-projection GEMM, the residual epilogue; the shader clock
(s_memtime ticks per 100 MHz s_memrealtime tick) and how the workgroups' lifetimes tile the launch."""
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
wl = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "tetrapeptide_fwdsim_crop4_T1000_B16"
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[1:] if "=" in kv}
B, T, L, abs_pos, n_pad = bench.WORKLOADS[wl]
cfg = ModelConfig(crop=L, num_frames=T, abs_pos_emb=abs_pos, sim_condition=True)
w = NewMDGenWrapper(cfg, device=dev)
w.model.load_state_dict(synth_state_dict(cfg, 0))
w.model.set_option("streams", 1)
w.model.set_option("flash_proj", 2)
for k, v in opts.items():
    w.model.set_option(k, v)
batch = bench.synth_batch(B, T, L, n_pad, dev, seed=100)
zs = torch.randn(B, T, L, 21, generator=torch.Generator().manual_seed(137)).to(dev)
w.inference(batch, zs=zs, num_steps=2, use_graph=False)
torch.cuda.synchronize()
host = np.zeros(8192 * 16, dtype=np.uint64)
fn = lib.mdgen_dev_fproj_stamps
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(host.ctypes.data, host.nbytes) == 0
s = host.reshape(-1, 16).astype(np.int64)
s = s[s[:, 0] > 0]
life = s[:, 7] - s[:, 0]
real = s[:, 9] - s[:, 8]
mhz = life.sum() / real.sum() * 100.0
span = (s[:, 9].max() - s[:, 8].min()) / 100.0
hg = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3]], 1)
print(f"{wl} {opts}: {len(s)} waves stamped, launch span {span:.1f} us, shader clock {mhz:.0f} MHz")
print(f"  wave lifetime {life.mean():.0f} cycles (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f})")
print(f"  head-group jobs {hg.mean(0).round().tolist()} (mean {hg.mean():.0f}); barrier wait {np.mean(s[:, 5] - s[:, 4]):.0f}; "
      f"out-projection GEMM {np.mean(s[:, 6] - s[:, 5]):.0f}; barrier + residual epilogue {np.mean(s[:, 7] - s[:, 6]):.0f}")
# flash_job's own stamps, per head group (rows 4096 hg + workgroup * 4 + wave): 0 start, 1 anchor done, 2 loop done, 3 end
inner = np.zeros(32768 * 16, dtype=np.uint64)
fi = lib.mdgen_dev_flash_stamps
fi.argtypes = [C.c_void_p, C.c_size_t]
assert fi(inner.ctypes.data, inner.nbytes) == 0
inner = inner.reshape(-1, 16).astype(np.int64)
nw = len(host.reshape(-1, 16)[host.reshape(-1, 16)[:, 0] > 0])
first = (s[:, 8] - s[:, 8].min()) < 100       # waves of the launch's first round (started within 1 us)
for g in range(4):
    r = inner[4096 * g:4096 * g + nw]
    ok = r[:, 0] > 0
    pro, loop, tail = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2]
    f = ok & first[:len(r)] if len(first) >= len(r) else ok
    l = ok & ~first[:len(r)] if len(first) >= len(r) else ok
    print(f"  job {g}: anchor/prologue {pro[ok].mean():.0f}  loop {loop[ok].mean():.0f}  tail {tail[ok].mean():.0f}   first round: {pro[f].mean():.0f} / {loop[f].mean():.0f} / {tail[f].mean():.0f}"
          f"   later: {pro[l].mean() if l.any() else 0:.0f} / {loop[l].mean() if l.any() else 0:.0f} / {tail[l].mean() if l.any() else 0:.0f}")
if os.environ.get("FPROJ_DUMP"):
    np.save(os.environ["FPROJ_DUMP"], s)
# how many workgroups were resident per CU over time: start times (realtime) sorted -> rounds
st = np.sort(s[::4, 8] - s[:, 8].min()) / 100.0
print(f"  workgroup start times (us): p25 {np.percentile(st, 25):.1f} p50 {np.percentile(st, 50):.1f} p75 {np.percentile(st, 75):.1f} max {st.max():.1f}")
