"""ctypes binding of the device-layer C ABI (include/moonshine_hip.h).

Used by the tests and bench.py.  There is no fallback: if ``libmoonshine.so`` is missing or no
MI355X is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmoonshine.so")


class MshError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[msh {code}] {msg}")
        self.code = code


class ModelInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden", "ffn", "enc_layers", "dec_layers", "heads", "head_dim", "vocab", "bos", "eos")] + [
        ("arch", C.c_char * 16)
    ]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_uint64), ("flops", C.c_double), ("bytes", C.c_double)]


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libmoonshine.so and declare every msh_* prototype."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: run `python -m moonshine_amd.build` (needs hipcc); there is no CPU fallback")
    lib = C.CDLL(p)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float
    P = C.POINTER
    protos = {
        "msh_device_count": (i32, []),
        "msh_version": (C.c_char_p, []),
        "msh_create": (i32, [i32, P(vp)]),
        "msh_destroy": (None, [vp]),
        "msh_last_error": (C.c_char_p, [vp]),
        "msh_load_weights_file": (i32, [vp, C.c_char_p, i32]),
        "msh_load_weights_memory": (i32, [vp, vp, u64, i32]),
        "msh_model_info_get": (i32, [vp, P(ModelInfo)]),
        "msh_encode": (i32, [vp, P(vp), P(u64), u32, i32, f32]),
        "msh_decode": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, i32]),
        "msh_transcribe_tokens": (i32, [vp, P(vp), P(u64), u32, i32, f32, i32, vp, vp, i32]),
        "msh_max_decode_steps": (i32, [vp]),
        "msh_clip_frames": (i32, [vp, u32]),
        "msh_set_keep_encoder_output": (i32, [vp, i32]),
        "msh_get_encoder_output": (i32, [vp, u32, vp]),
        "msh_profile_enable": (i32, [vp, i32]),
        "msh_profile_reset": (i32, [vp]),
        "msh_profile_count": (i32, [vp]),
        "msh_profile_get": (i32, [vp, i32, P(ProfileEntry)]),
        "msh_synchronize": (i32, [vp]),
        "msh_host_tokens_to_text": (C.c_int64, [vp, u64, vp, u64, vp, u64]),
        "msh_host_sanitize_utf8": (C.c_int64, [vp, u64, vp, u64]),
        "msh_host_resample": (C.c_int64, [vp, u64, f32, f32, vp, u64]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


DECLARED_SYMBOLS = [
    "msh_device_count", "msh_version", "msh_create", "msh_destroy", "msh_last_error", "msh_load_weights_file",
    "msh_load_weights_memory", "msh_model_info_get", "msh_encode", "msh_decode", "msh_transcribe_tokens",
    "msh_max_decode_steps", "msh_clip_frames", "msh_set_keep_encoder_output", "msh_get_encoder_output",
    "msh_profile_enable", "msh_profile_reset", "msh_profile_count", "msh_profile_get", "msh_synchronize",
    "msh_host_tokens_to_text", "msh_host_sanitize_utf8", "msh_host_resample",
]


class Engine:
    """One MI355X engine (one GPU, one stream).  Thin, typed wrapper; no arithmetic here."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.msh_create(device, C.byref(h))
        if rc != 0:
            raise MshError(rc, self.lib.msh_last_error(None).decode())
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.msh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc < 0:
            raise MshError(rc, self.lib.msh_last_error(self.h).decode())
        return rc

    def load_weights_file(self, path: str, arch: int = -1):
        self._check(self.lib.msh_load_weights_file(self.h, path.encode(), arch))

    def load_weights_memory(self, blob: bytes, arch: int = -1):
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.lib.msh_load_weights_memory(self.h, C.cast(buf, C.c_void_p), len(blob), arch))

    def info(self) -> ModelInfo:
        mi = ModelInfo()
        self._check(self.lib.msh_model_info_get(self.h, C.byref(mi)))
        return mi

    # -- batch calls -------------------------------------------------------------------------
    def _clip_args(self, clips, device_ptrs):
        n = len(clips)
        ptrs = (C.c_void_p * n)()
        lens = (C.c_uint64 * n)()
        keep = []
        if device_ptrs is not None:
            for i, (p, ln) in enumerate(device_ptrs):
                ptrs[i] = p
                lens[i] = ln
        else:
            for i, c in enumerate(clips):
                a = np.ascontiguousarray(c, dtype=np.float32)
                keep.append(a)
                ptrs[i] = a.ctypes.data
                lens[i] = a.shape[0]
        return ptrs, lens, keep

    def encode(self, clips=None, max_tokens_per_second: float = 6.5, device_ptrs=None):
        """clips: list of 1-D float32 arrays (host), or device_ptrs: list of (device address, n_samples)."""
        n = len(device_ptrs) if device_ptrs is not None else len(clips)
        ptrs, lens, keep = self._clip_args(clips if clips is not None else [None] * n, device_ptrs)
        self._check(self.lib.msh_encode(self.h, ptrs, lens, n, 1 if device_ptrs is not None else 0, max_tokens_per_second))
        self._n = n

    def set_keep_encoder_output(self, keep: bool = True):
        self._check(self.lib.msh_set_keep_encoder_output(self.h, 1 if keep else 0))

    def encoder_output(self, clip: int) -> np.ndarray:
        T = self._check(self.lib.msh_clip_frames(self.h, clip))
        out = np.empty((T, self.info().hidden), np.float32)
        self._check(self.lib.msh_get_encoder_output(self.h, clip, out.ctypes.data))
        return out

    def max_decode_steps(self) -> int:
        return self._check(self.lib.msh_max_decode_steps(self.h))

    def decode(self, forced_steps: int = -1, teacher: np.ndarray | None = None, want_logits: int = 0):
        """Returns (tokens list per clip, logits [steps, n, V] or None)."""
        n = self._n
        steps = forced_steps if forced_steps >= 0 else self.max_decode_steps()
        stride = steps + 1
        tokens = np.full((n, stride), -1, np.int32)
        counts = np.zeros(n, np.int32)
        t_ptr, t_stride = None, 0
        if teacher is not None:
            teacher = np.ascontiguousarray(teacher, dtype=np.int32)
            t_ptr, t_stride = teacher.ctypes.data, teacher.shape[1]
        logits = None
        l_ptr = None
        if want_logits > 0:
            logits = np.zeros((want_logits, n, self.info().vocab), np.float32)
            l_ptr = logits.ctypes.data
        self._check(self.lib.msh_decode(self.h, forced_steps, t_ptr, t_stride, l_ptr, want_logits, tokens.ctypes.data, counts.ctypes.data, stride))
        return [tokens[i, : counts[i]].tolist() for i in range(n)], logits

    def transcribe_tokens(self, clips=None, max_tokens_per_second: float = 6.5, forced_steps: int = -1, device_ptrs=None):
        self.encode(clips, max_tokens_per_second, device_ptrs)
        toks, _ = self.decode(forced_steps)
        return toks

    # -- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self.lib.msh_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.msh_profile_reset(self.h))

    def profile(self) -> list[dict]:
        n = self._check(self.lib.msh_profile_count(self.h))
        out = []
        for i in range(n):
            pe = ProfileEntry()
            self._check(self.lib.msh_profile_get(self.h, i, C.byref(pe)))
            out.append({"name": pe.name.decode(), "ms": pe.ms, "launches": int(pe.launches), "flops": pe.flops, "bytes": pe.bytes})
        return out

    def synchronize(self):
        self._check(self.lib.msh_synchronize(self.h))
