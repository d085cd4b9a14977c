"""Build oracle/_ref/libmoonshine_ref_host.so: the REFERENCE's own host sources for the byte / integer / sample rows
(resampler, tokenizer, word alignment, context biaser / extractor, WAV I/O) behind the C shim oracle/ref_shim.cpp.

TEST INFRASTRUCTURE ONLY: the library is loaded by tests/test_ref_host_diff.py to diff libmoonshine.so's msh_host_*
helpers against the reference byte for byte.  Sources are compiled where they lie under /root/reference (nothing is
copied), with plain g++ -- not the reference's cmake build.  The output directory oracle/_ref/ is git-ignored but
travels to the GPU box with the tree.  /root/reference does not exist there: build_ref() then returns None and the
tests use the prebuilt library (or skip when there is none).

    python -m oracle.build_ref
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CORE = "/root/reference/core"
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libmoonshine_ref_host.so")
REF_SOURCES = ["resampler.cpp", "word-alignment.cpp", "context-biaser.cpp", "context-extractor.cpp",
               "bin-tokenizer/bin-tokenizer.cpp", "moonshine-utils/debug-utils.cpp", "moonshine-utils/string-utils.cpp",
               "moonshine-utils/file-utils.cpp"]


def ref_library_path() -> str | None:
    return LIB if os.path.exists(LIB) else None


def build_ref(force: bool = False) -> str | None:
    if not os.path.isdir(REF_CORE):
        return None
    srcs = [os.path.join(REF_CORE, s) for s in REF_SOURCES] + [os.path.join(HERE, "ref_shim.cpp")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) > os.path.getmtime(s) for s in srcs):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O2", "-fPIC", "-shared", "-w", f"-I{REF_CORE}", f"-I{REF_CORE}/moonshine-utils",
           f"-I{REF_CORE}/bin-tokenizer", "-o", LIB, *srcs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building oracle/_ref failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build_ref(force=True))
