"""Build libmoonshine.so (HIP kernels for gfx950 + host C++) in-tree with hipcc.

    python -m moonshine_amd.build            # incremental
    python -m moonshine_amd.build --force

hipcc cross-compiles gfx950 without a GPU.  Objects go to moonshine_amd/_build/, the library to
moonshine_amd/lib/libmoonshine.so (git-ignored, but shipped to the GPU box with the tree).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmoonshine.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]


# extra flags per source file ("packed-FP32 glitch")
PER_FILE: dict[str, list[str]] = {f: ["-fno-slp-vectorize"] for f in ("k_gemm_dec.hip", "k_gemm.hip", "k_attn.hip", "k_misc.hip", "k_stream.hip")}


def hipcc() -> str:
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found: the MI355X engine needs the ROCm toolchain to build")


def sources() -> list[str]:
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".cpp"):
            out.append(os.path.join(CSRC, f))
    return out


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src: str, force: bool, hdr_t: float) -> tuple[str, str]:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
        return obj, ""
    cmd = [hipcc(), f"--offload-arch={ARCH}", *COMMON, *PER_FILE.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    hdr_t = _newest_header()
    objs = []
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        for obj, log in ex.map(lambda s: _compile(s, force, hdr_t), srcs):
            objs.append(obj)
            if verbose and log:
                print(log, file=sys.stderr)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
