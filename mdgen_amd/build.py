"""Build libmdgen_amd.so (gfx950) in-tree with hipcc.  `python -m mdgen_amd.build [-j N] [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with gpurun.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmdgen_amd.so")
SOURCES = ["api.hip", "k_gemm.hip", "k_rows.hip", "k_flash.hip", "k_small.hip", "k_se3.hip", "k_fp32.hip", "k_optim.hip", "k_fp32_bwd.hip", "k_attn16.hip", "k_wide16.hip"]
HEADERS = ["common.h", "dev.h", "panel.h", "rows.h", "linear.h", "kernels.h", "train.inc", os.path.join("..", "..", "include", "mdgen_amd.h")]
# -fno-slp-vectorize: keeps hipcc from fusing scalar fp32 math into v_pk_{mul,add,fma}_f32.  On MI355X those
# packed ops (a) are an anti-lever beside MFMAs (MI355X_MICROARCH "price of one filler") and (b) produced
# intermittently wrong results in lanes 48-63 when two waves shared a SIMD (DESIGN.md "packed-fp32 hazard").
# check_isa() fails the build if a v_pk_*_f32 or an MFMA with dst overlapping a source is emitted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function", "-Wno-pass-failed"]
# per-file extras.  k_flash: no NaN-aware code in it, and without this every fmaxf on an MFMA result gets a
# canonicalising v_max(x, x) in front (15 extra VALU per key tile in a VALU-bound loop).
EXTRA = {"k_flash.hip": ["-fno-honor-nans"]}


def flags_for(src: str):
    return FLAGS + EXTRA.get(os.path.basename(src), [])


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


_MFMA_RE = None


def check_isa(cc: str, src: str) -> None:
    """Fail the build if hipcc emitted an MFMA whose destination overlaps its A/B sources (see
    csrc/common.h `opaque_zero`): such code runs, but corrupts accumulators intermittently."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([cc] + [f for f in flags_for(src) if f not in ("-fPIC",)] + ["-S", "--cuda-device-only", src, "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("ISA check failed to compile " + src + "\n" + r.stderr)
        pat = re.compile(r"v_mfma_\w+ ([va])\[(\d+):(\d+)\], ([va])\[(\d+):(\d+)\], ([va])\[(\d+):(\d+)\], (.*)$")
        n = 0
        own_m0 = os.path.basename(src) in ("k_rows.hip", "k_wide16.hip")   # rows.h dma_frag owns M0 there (not saved / restored)
        in_asm = False
        for line in open(out):
            if own_m0:
                if "#ASMSTART" in line:
                    in_asm = True
                elif "#ASMEND" in line:
                    in_asm = False
                elif not in_asm and re.search(r"\bm0\b", line.split(";")[0]):
                    raise RuntimeError(f"{os.path.basename(src)}: compiler-generated M0 access: {line.strip()}")
            if re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
                raise RuntimeError(f"{os.path.basename(src)}: packed fp32 VALU op emitted: {line.strip()}")
            m = pat.search(line)
            if not m:
                continue
            n += 1
            g = m.groups()
            dc, ac, bc = g[0], g[3], g[6]   # register file of each operand: v = VGPR, a = accumulation register
            d0, d1, a0, a1, b0, b1 = int(g[1]), int(g[2]), int(g[4]), int(g[5]), int(g[7]), int(g[8])
            if (ac == dc and not (a1 < d0 or a0 > d1)) or (bc == dc and not (b1 < d0 or b0 > d1)):
                raise RuntimeError(f"{os.path.basename(src)}: MFMA destination overlaps a source operand: {line.strip()}")
    return n


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    todo = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            todo.append((s, o))

    def compile_one(so):
        s, o = so
        cmd = [cc] + flags_for(s) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if os.path.basename(s).startswith("k_"):
            try:
                check_isa(cc, s)
            except Exception:
                if os.path.exists(o):
                    os.remove(o)
                raise
        return s, r.stderr

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for s, err in ex.map(compile_one, todo):
                if verbose:
                    print("compiled", os.path.basename(s))
                    if err.strip():
                        print(err.strip())
    if force or todo or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=8)
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.j))
