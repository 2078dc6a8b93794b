"""GPU: phase stamps of k16_linear_wide (library built by KFILE=k_wide16 KPFX=WIDE bash scripts/micro/flash_variants.sh STAMPS;
run with MDGEN_AMD_LIB=gpurun_out/dev_libs/libmdgen_amd_STAMPS.so).  s_memtime ticks at 100 MHz."""
import ctypes, os, sys, runpy
import numpy as np
sys.argv = [sys.argv[0], "1", "250", "256", "1", "16"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "train_bench.py"), run_name="__main__")
from mdgen_amd import _lib
L = _lib.lib
n = 512 * 2 * 24
buf = (ctypes.c_ulonglong * n)()
L.mdgen_dev_wide_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", L.mdgen_dev_wide_stamps(buf, n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 2, 24).astype(np.int64)
ok = a[:, 0, 0] > 0
a = a[ok]
t0 = a[:, :, 0].min()
print("workgroups stamped", a.shape[0])
d = np.diff(a, axis=2)   # [wg][wave][23]
names = ["prologue"] + [f"k{i}:{p}" for i in range(3) for p in ("->ks0", "ks0", "ks1", "ks2", "ks3", "stage", "barrier")] + ["rest"]
names = names[:1] + names[2:]   # the interval before the first ks0 stamp is empty for k0
for w in (0, 1):
    print("wave", "0" if w == 0 else "5", " median ticks (10 ns):")
    print("  ".join(f"{n}={np.median(d[:, w, i]):.0f}" for i, n in enumerate(names)))
print("start spread (ticks):", np.percentile(a[:, 0, 0] - t0, [0, 25, 50, 75, 100]))
print("total per wg (ticks): median", np.median(a[:, 0, 23] - a[:, 0, 0]))
