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
SOURCES = ["api.hip", "k_gemm.hip", "k_flash.hip", "k_small.hip", "k_se3.hip"]
HEADERS = ["common.h", "panel.h", "kernels.h", os.path.join("..", "..", "include", "mdgen_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


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
        cmd = [cc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
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
