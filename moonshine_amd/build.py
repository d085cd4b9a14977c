"""Build libmoonshine.so (HIP kernels for gfx950 + host C++) in-tree with hipcc.

    python -m moonshine_amd.build            # incremental
    python -m moonshine_amd.build --force

hipcc cross-compiles gfx950 without a GPU.  Objects go to moonshine_amd/_build/, the library to
moonshine_amd/lib/libmoonshine.so (git-ignored, but shipped to the GPU box with the tree); beside it
libmoonshine_dev.so = the same objects + csrc/dev_hooks.cpp (msh_test_*: kernel-alone tests and microbenchmarks),
which the product library does not export.
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
LIB_DEV = os.path.join(LIBDIR, "libmoonshine_dev.so")   # the same objects + the development hooks (include/moonshine_hip_dev.h)
DEV_ONLY = {"dev_hooks.cpp"}                            # sources that never go into the product library
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]


# Device code is built WITHOUT the packed-FP32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 /
# v_pk_mov_b32): see DESIGN.md section 5b, "packed-FP32 glitch".  `-fno-slp-vectorize` alone only stops the
# SLP-formed packs (the loop vectoriser and the DAG combiner still emitted 84 of them); switching the target
# feature off removes the instructions from the selector altogether.  The host pass of the same hipcc call does
# not know the feature and says so on stderr ("not a recognized feature"): filtered below.
NO_PACKED_FP32 = ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PER_FILE: dict[str, list[str]] = {}
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
PACKED_FP32_RE = r"\bv_pk_(fma|mul|add)_f32\b"


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
    cmd = [hipcc(), f"--offload-arch={ARCH}", *COMMON, *NO_PACKED_FP32, *PER_FILE.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    log = "\n".join(l for l in r.stderr.splitlines() if "not a recognized feature for this target" not in l)
    return obj, log


def device_disassembly(obj: str) -> str:
    """gfx950 disassembly of the code object embedded in a hipcc object file ('' if it has no device code)."""
    import tempfile

    objcopy, bundler, objdump = (os.path.join(LLVM_BIN, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        r = subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(d, "copy.o")], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""
        r = subprocess.run([bundler, "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}", f"--output={co}", "--unbundle"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"cannot unbundle the {ARCH} code object of {obj}: {r.stderr}")
        r = subprocess.run([objdump, "-d", co], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump failed on {obj}: {r.stderr}")
        return r.stdout


def check_no_packed_fp32(objs: list[str]) -> dict[str, int]:
    """Disassemble every gfx950 code object and fail on any packed-FP32 arithmetic instruction.  Returns
    {object: number of device instructions inspected} so that callers can see the check really ran."""
    import re

    pat = re.compile(PACKED_FP32_RE)
    seen: dict[str, int] = {}
    bad: list[str] = []
    for obj in objs:
        asm = device_disassembly(obj)
        if not asm:
            continue
        fn = "?"
        n = 0
        for line in asm.splitlines():
            if line.endswith(">:"):
                fn = line.split("<", 1)[-1][:-2]
            elif line.startswith("\t"):
                n += 1
                if pat.search(line):
                    bad.append(f"{os.path.basename(obj)}: {fn}: {line.strip().split('//')[0].strip()}")
        seen[os.path.basename(obj)] = n
    if bad:
        raise RuntimeError("packed-FP32 instructions in the gfx950 code (see DESIGN.md 5b):\n  " + "\n  ".join(bad[:40]) +
                           (f"\n  ... {len(bad)} in total" if len(bad) > 40 else ""))
    return seen


def build(force: bool = False, verbose: bool = False, check: bool = True) -> str:
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
    stamp = os.path.join(OBJ, "no_packed_fp32.ok")
    if check and (force or not os.path.exists(stamp) or any(os.path.getmtime(o) > os.path.getmtime(stamp) for o in objs)):
        seen = check_no_packed_fp32(objs)
        with open(stamp, "w") as f:
            f.write("\n".join(f"{k} {v}" for k, v in sorted(seen.items())) + "\n")
        if verbose:
            print("no packed-FP32 instructions in", ", ".join(f"{k} ({v} instr.)" for k, v in sorted(seen.items())), file=sys.stderr)
    product = [o for o in objs if os.path.basename(o)[:-2] not in DEV_ONLY]
    for lib, members in ((LIB, product), (LIB_DEV, objs)):
        if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in members):
            cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *members, "-lpthread"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    try:
        path = build(force="--force" in sys.argv, verbose=True)
    except Exception as e:  # the last line of the output says which it was (a `| tail -1` once hid a failed link for half an hour)
        print(e, file=sys.stderr)
        print("BUILD FAILED: the library was NOT relinked", file=sys.stderr)
        sys.exit(1)
    print(path)
