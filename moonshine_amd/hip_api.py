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


class StreamInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "encoder_dim", "decoder_dim", "depth", "nheads", "head_dim", "vocab_size", "bos_id", "eos_id", "frame_len",
        "total_lookahead", "max_seq_len", "enc_layers", "encoder_heads", "max_slots", "memory_capacity")]


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
        "msh_set_kv_dtype": (i32, [vp, i32]),
        "msh_set_cross_mode": (i32, [vp, i32]),
        "msh_set_uniform_kernels": (i32, [vp, i32]),
        "msh_uniform_kernels": (i32, [vp]),
        "msh_cross_absorbed": (i32, [vp]),
        "msh_cross_absorbed_supported": (i32, [vp]),
        "msh_profile_enable": (i32, [vp, i32]),
        "msh_profile_reset": (i32, [vp]),
        "msh_profile_count": (i32, [vp]),
        "msh_profile_get": (i32, [vp, i32, P(ProfileEntry)]),
        "msh_synchronize": (i32, [vp]),
        "msh_profile_event_overhead_ms": (C.c_double, [vp, i32]),
        "msh_profile_cross_attention_ms": (C.c_double, [vp, i32]),
        "msh_profile_decode_chain": (i32, [vp, i32]),
        "msh_set_capture_cross_attention": (i32, [vp, i32]),
        "msh_get_cross_attention": (C.c_int64, [vp, C.c_uint32, vp, C.c_uint64, vp]),
        "msh_set_batches_in_flight": (i32, [vp, i32]),
        "msh_submit_transcribe_tokens": (C.c_int64, [vp, vp, vp, C.c_uint32, i32, C.c_float, i32, vp, vp, i32]),
        "msh_wait": (i32, [vp, C.c_int64]),
        "msh_stream_create": (i32, [i32, C.c_char_p, C.c_char_p, i32, i32, P(vp)]),
        "msh_stream_create_from_memory": (i32, [i32, vp, u64, C.c_char_p, i32, i32, P(vp)]),
        "msh_stream_destroy": (None, [vp]),
        "msh_stream_last_error": (C.c_char_p, [vp]),
        "msh_stream_info_get": (i32, [vp, P(StreamInfo)]),
        "msh_stream_open": (i32, [vp]),
        "msh_stream_close": (i32, [vp, i32]),
        "msh_stream_reset": (i32, [vp, i32]),
        "msh_stream_process_audio": (i32, [vp, i32, vp, P(vp), P(u64), vp]),
        "msh_stream_encode": (i32, [vp, i32, vp, vp, vp]),
        "msh_stream_decoder_reset": (i32, [vp, i32, vp]),
        "msh_stream_decode_tokens": (i32, [vp, i32, vp, P(vp), vp, vp]),
        "msh_stream_cross_attention": (C.c_int64, [vp, i32, vp, i32, vp, u64, vp]),
        "msh_stream_decode_full": (i32, [vp, i32, vp, P(vp), vp, vp, vp, vp, i32, vp]),
        "msh_stream_set_bias": (i32, [vp, i32, vp, vp, vp, vp, vp, i32]),
        "msh_stream_query": (i32, [vp, i32, i32]),
        "msh_stream_profile_enable": (i32, [vp, i32]),
        "msh_stream_profile_reset": (i32, [vp]),
        "msh_stream_profile_count": (i32, [vp]),
        "msh_stream_profile_get": (i32, [vp, i32, P(ProfileEntry)]),
        "msh_stream_get_memory": (i32, [vp, i32, vp]),
        "msh_stream_get_features": (i32, [vp, i32, vp]),
        "msh_host_tokens_to_text": (C.c_int64, [vp, u64, vp, u64, vp, u64]),
        "msh_host_sanitize_utf8": (C.c_int64, [vp, u64, vp, u64]),
        "msh_host_resample": (C.c_int64, [vp, u64, f32, f32, vp, u64]),
        "msh_set_hw_queues": (i32, [i32]),
        "msh_host_load_wav": (C.c_int64, [C.c_char_p, vp, u64, vp]),
        "msh_host_save_wav": (i32, [C.c_char_p, vp, u64, i32]),
        "msh_host_silero_probabilities": (C.c_int64, [vp, u64, vp, u64, vp, u64, vp]),
        "msh_host_vad_segments": (C.c_int64, [vp, u64, f32, C.c_int32, C.c_int32, u64, u64, u64, vp, u64, C.c_int32, u64, vp, u64]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


_dev_lib = None


def load_dev_library() -> C.CDLL:
    """libmoonshine_dev.so: the product library's objects + the development hooks of include/moonshine_hip_dev.h (msh_test_*:
    kernel-alone test entry points, microbenchmarks).  A second, independent copy of the library in the process: the hooks
    allocate and free their own device buffers and share nothing with engines created through load_library()."""
    global _dev_lib
    if _dev_lib is not None:
        return _dev_lib
    p = os.path.join(os.path.dirname(LIB_PATH), "libmoonshine_dev.so")
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: run `python -m moonshine_amd.build`")
    lib = load_library(p)    # (every product prototype: the development library exports the whole product API too)
    vp, i32 = C.c_void_p, C.c_int32
    protos = {
        "msh_test_debug_read": (C.c_int64, [vp, C.c_char_p, vp, C.c_uint64]),
        "msh_test_device_alloc": (i32, []),
        "msh_test_gemm_microbench": (C.c_float, [i32, i32, i32, C.c_int64, i32, i32, i32]),
        "msh_test_mlp_microbench": (C.c_float, [i32, i32, i32, i32, i32]),
        "msh_test_mlp_run": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
        "msh_test_mlp_oproj_run": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "msh_test_qkv_panel": (C.c_float, [i32, i32, i32, vp, vp, vp, vp, vp]),
        "msh_test_mlp_oproj_y_run": (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "msh_test_cross_absorbed": (C.c_float, [vp, vp, C.c_int64, vp, vp, i32, i32, vp, i32]),
        "msh_test_crossq2": (C.c_float, [vp, vp, vp, i32, i32, vp, i32]),
        "msh_test_enc_attention": (C.c_float, [i32, i32, i32, i32, i32, i32, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _dev_lib = lib
    return lib


DEV_SYMBOLS = ["msh_test_debug_read", "msh_test_device_alloc", "msh_test_gemm_microbench", "msh_test_mlp_microbench", "msh_test_mlp_run",
               "msh_test_mlp_oproj_run", "msh_test_mlp_oproj_y_run", "msh_test_qkv_panel", "msh_test_cross_absorbed", "msh_test_crossq2", "msh_test_enc_attention"]

DECLARED_SYMBOLS = [
    "msh_device_count", "msh_version", "msh_create", "msh_destroy", "msh_last_error", "msh_load_weights_file",
    "msh_load_weights_memory", "msh_model_info_get", "msh_encode", "msh_decode", "msh_transcribe_tokens",
    "msh_max_decode_steps", "msh_clip_frames", "msh_set_keep_encoder_output", "msh_get_encoder_output", "msh_set_kv_dtype",
    "msh_profile_enable", "msh_profile_reset", "msh_profile_count", "msh_profile_get", "msh_synchronize", "msh_profile_event_overhead_ms", "msh_profile_cross_attention_ms", "msh_profile_decode_chain",
    "msh_set_batches_in_flight", "msh_submit_transcribe_tokens", "msh_wait", "msh_set_capture_cross_attention", "msh_get_cross_attention",
    "msh_host_tokens_to_text", "msh_host_sanitize_utf8", "msh_host_resample", "msh_host_text_to_tokens",
    "msh_host_biaser_bonuses", "msh_host_context_terms", "msh_host_dtw", "msh_host_median_filter", "msh_host_align_words", "msh_host_load_wav", "msh_host_save_wav", "msh_set_hw_queues", "msh_host_silero_probabilities", "msh_host_vad_segments", "msh_stream_create", "msh_stream_create_from_memory", "msh_stream_destroy",
    "msh_stream_last_error", "msh_stream_info_get", "msh_stream_open", "msh_stream_close", "msh_stream_reset",
    "msh_stream_process_audio", "msh_stream_encode", "msh_stream_decoder_reset", "msh_stream_decode_tokens", "msh_stream_cross_attention",
    "msh_stream_decode_full", "msh_stream_set_bias", "msh_stream_query", "msh_stream_get_memory",
    "msh_stream_profile_enable", "msh_stream_profile_reset", "msh_stream_profile_count", "msh_stream_profile_get",
    "msh_set_cross_mode", "msh_set_uniform_kernels", "msh_uniform_kernels", "msh_cross_absorbed", "msh_cross_absorbed_supported",
    "msh_stream_get_features",
]


class Engine:
    """One MI355X engine (one GPU, one stream).  Thin, typed wrapper; no arithmetic here."""

    def __init__(self, device: int = 0, dev: bool = False):
        """dev = True: the engine lives in libmoonshine_dev.so (same objects + the msh_test_* hooks), which is what debug_read /
        graph_captures need -- the product library exports no test hook."""
        self.lib = load_dev_library() if dev else load_library()
        self.dev = dev
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

    def set_kv_dtype(self, dtype: str = "bf16"):
        """Cross K / V storage: "bf16" (default) or "fp8" (e4m3, per-row scales fixed at load); before set_batches_in_flight."""
        self._check(self.lib.msh_set_kv_dtype(self.h, {"bf16": 0, "fp8": 1}[dtype]))

    def set_cross_mode(self, mode: str = "kv"):
        """Form of the decoder's cross-attention, ONE per engine whatever the batch size: "kv" (projected K^T / V^T stream,
        the reference's form and the default) or "absorbed" (one pass over the encoder output for all heads, k_xattn.hip:
        pays from ~192 clips per batch on); before set_batches_in_flight."""
        self._check(self.lib.msh_set_cross_mode(self.h, {"default": 0, "kv": 1, "absorbed": 2}[mode]))

    def set_uniform_kernels(self, on: bool = True):
        """One kernel set (the large-batch one) for every call: a clip's ids do not depend on what shares its call."""
        self._check(self.lib.msh_set_uniform_kernels(self.h, 1 if on else 0))

    def cross_absorbed_supported(self) -> bool:
        return bool(self.lib.msh_cross_absorbed_supported(self.h))

    def graph_captures(self) -> int:
        """Decode-step hipGraphs instantiated so far (captured steps are cached per batch shape)."""
        return int(self._debug_read_fn()(self.h, b"graph_captures", None, 0))

    def _debug_read_fn(self):
        if not self.dev:
            raise MshError(-1, "debug_read / graph_captures need Engine(dev=True): the product library exports no test hook")
        return self.lib.msh_test_debug_read

    def cross_absorbed(self) -> bool:
        return bool(self.lib.msh_cross_absorbed(self.h))

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

    def profile_event_overhead_ms(self, iters: int = 200) -> float:
        return float(self.lib.msh_profile_event_overhead_ms(self.h, iters))

    def profile_decode_chain(self, reps: int = 8):
        """Adds "chain_<kernel group>" entries to profile(): per-launch cost inside a replayed hipGraph chain."""
        self._check(self.lib.msh_profile_decode_chain(self.h, reps))

    def profile_cross_attention_ms(self, rounds: int = 20) -> float:
        v = float(self.lib.msh_profile_cross_attention_ms(self.h, rounds))
        if v < 0:
            raise MshError(-1, (self.lib.msh_last_error(self.h) or b"").decode())
        return v

    def synchronize(self):
        self._check(self.lib.msh_synchronize(self.h))

    # -- word timestamps -------------------------------------------------------------------------
    def set_capture_cross_attention(self, on: bool = True):
        self._check(self.lib.msh_set_capture_cross_attention(self.h, 1 if on else 0))

    def cross_attention(self, clip: int) -> np.ndarray:
        """[layers*heads, steps, frames] fp32 cross-attention probabilities of the last decode() for one clip."""
        dims = (C.c_int32 * 3)()
        n = int(self.lib.msh_get_cross_attention(self.h, clip, None, 0, dims))
        if n < 0:
            raise MshError(n, (self.lib.msh_last_error(self.h) or b"").decode())
        out = np.zeros((dims[0], dims[1], dims[2]), np.float32)
        if n > 0:
            n = int(self.lib.msh_get_cross_attention(self.h, clip, out.ctypes.data, out.size, dims))
            if n < 0:
                raise MshError(n, (self.lib.msh_last_error(self.h) or b"").decode())
        return out

    # -- batches in flight ------------------------------------------------------------------------
    def set_batches_in_flight(self, n: int):
        self._check(self.lib.msh_set_batches_in_flight(self.h, n))

    def submit_transcribe_tokens(self, clips=None, max_tokens_per_second: float = 6.5, forced_steps: int = -1,
                                 device_ptrs=None, max_steps: int | None = None):
        """Queue one batch on the lanes; returns a ticket for wait_tokens().  max_steps bounds the token rows when the
        reference budget applies (forced_steps < 0): ceil(longest clip in s * max_tokens_per_second)."""
        n = len(device_ptrs) if device_ptrs is not None else len(clips)
        ptrs, lens, keep = self._clip_args(clips if clips is not None else [None] * n, device_ptrs)
        if forced_steps >= 0:
            steps = forced_steps
        elif max_steps is not None:
            steps = max_steps
        else:
            longest = max(int(l) for l in lens)
            steps = max(1, int(np.ceil(longest / 16000.0 * max_tokens_per_second))) + 1  # >= the engine's own budget
        stride = steps + 1
        tokens = np.full((n, stride), -1, np.int32)
        counts = np.zeros(n, np.int32)
        t = self.lib.msh_submit_transcribe_tokens(self.h, ptrs, lens, n, 1 if device_ptrs is not None else 0,
                                                  max_tokens_per_second, forced_steps, tokens.ctypes.data, counts.ctypes.data, stride)
        if t < 0:
            raise MshError(int(t), (self.lib.msh_last_error(self.h) or b"").decode())
        return (int(t), tokens, counts, keep, ptrs, lens)

    def wait_tokens(self, ticket) -> list[list[int]]:
        t, tokens, counts = ticket[0], ticket[1], ticket[2]
        self._check(self.lib.msh_wait(self.h, t))
        return [tokens[i, : counts[i]].tolist() for i in range(tokens.shape[0])]

    def debug_read(self, name: str) -> np.ndarray:
        """Raw bytes of a decode buffer of the last decode() ("cache_k", "cache_v", "resid") -- test hook."""
        fn = self._debug_read_fn()
        size = int(fn(self.h, name.encode(), None, 0))
        if size < 0:
            raise MshError(-1, (self.lib.msh_last_error(self.h) or b"").decode())
        out = np.empty(size, np.uint8)
        fn(self.h, name.encode(), out.ctypes.data, size)
        return out


class StreamEngine:
    """msh_stream_engine: the batched streaming model (reference core/moonshine-streaming-model.h:73-201)."""

    def __init__(self, safetensors_path: str, config_json: str, device: int = 0, max_slots: int = 64,
                 max_memory_frames: int = 2048):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.msh_stream_create(device, safetensors_path.encode(), config_json.encode(), max_slots,
                                        max_memory_frames, C.byref(h))
        if rc != 0:
            raise MshError(rc, (self.lib.msh_stream_last_error(None) or b"").decode())
        self.h = h
        self._info = StreamInfo()
        self._check(self.lib.msh_stream_info_get(self.h, C.byref(self._info)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.msh_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc < 0:
            raise MshError(rc, (self.lib.msh_stream_last_error(self.h) or b"").decode())
        return rc

    @property
    def info(self) -> StreamInfo:
        return self._info

    def open(self) -> int:
        return self._check(self.lib.msh_stream_open(self.h))

    def close_stream(self, slot: int):
        self._check(self.lib.msh_stream_close(self.h, slot))

    def reset(self, slot: int):
        self._check(self.lib.msh_stream_reset(self.h, slot))

    @staticmethod
    def _slots(slots):
        return np.ascontiguousarray(slots, dtype=np.int32)

    def process_audio(self, slots, chunks) -> np.ndarray:
        s = self._slots(slots)
        keep = [np.ascontiguousarray(c, dtype=np.float32) for c in chunks]
        ptrs = (C.c_void_p * len(keep))(*[c.ctypes.data for c in keep])
        lens = (C.c_uint64 * len(keep))(*[c.shape[0] for c in keep])
        out = np.zeros(len(keep), np.int32)
        self._check(self.lib.msh_stream_process_audio(self.h, len(keep), s.ctypes.data, ptrs, lens, out.ctypes.data))
        return out

    def encode(self, slots, is_final) -> np.ndarray:
        s = self._slots(slots)
        fin = np.ascontiguousarray(np.broadcast_to(np.asarray(is_final, dtype=np.uint8), s.shape))
        out = np.zeros(s.shape[0], np.int32)
        self._check(self.lib.msh_stream_encode(self.h, s.shape[0], s.ctypes.data, fin.ctypes.data, out.ctypes.data))
        return out

    def decoder_reset(self, slots):
        s = self._slots(slots)
        self._check(self.lib.msh_stream_decoder_reset(self.h, s.shape[0], s.ctypes.data))

    def decode_tokens(self, slots, tokens, want_logits: bool = True):
        """tokens: one int sequence per stream.  Returns a list of [n_i, V] logits arrays."""
        s = self._slots(slots)
        keep = [np.ascontiguousarray(t, dtype=np.int32) for t in tokens]
        ptrs = (C.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
        lens = np.asarray([t.shape[0] for t in keep], np.int32)
        total = int(lens.sum())
        logits = np.empty((total, self._info.vocab_size), np.float32) if want_logits else None
        self._check(self.lib.msh_stream_decode_tokens(self.h, s.shape[0], s.ctypes.data, ptrs, lens.ctypes.data,
                                                      logits.ctypes.data if want_logits else None))
        if not want_logits:
            return None
        out, o = [], 0
        for n in lens:
            out.append(logits[o:o + n])
            o += n
        return out

    def cross_attention(self, slot: int, tokens) -> np.ndarray:
        """[depth*heads, len(tokens), memory_len] fp32 for `tokens` fed from an empty self cache (word timestamps)."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        dims = (C.c_int32 * 3)()
        n = int(self.lib.msh_stream_cross_attention(self.h, slot, t.ctypes.data, len(t), None, 0, dims))
        if n < 0:
            self._check(n)
        out = np.zeros((dims[0], dims[1], dims[2]), np.float32)
        self._check(int(min(0, self.lib.msh_stream_cross_attention(self.h, slot, t.ctypes.data, len(t), out.ctypes.data, out.size, dims))))
        return out

    def decode_full(self, slots, drafts=None, max_tokens=None):
        """Returns (list of token lists, accepted counts)."""
        s = self._slots(slots)
        n = s.shape[0]
        stride = self._info.max_seq_len + 8
        toks = np.zeros((n, stride), np.int32)
        counts = np.zeros(n, np.int32)
        acc = np.zeros(n, np.int32)
        if drafts is not None:
            keep = [np.ascontiguousarray(d if d is not None else [], dtype=np.int32) for d in drafts]
            ptrs = (C.c_void_p * n)(*[(d.ctypes.data if d.shape[0] else None) for d in keep])
            dl = np.asarray([d.shape[0] for d in keep], np.int32)
            pa, la = ptrs, dl.ctypes.data
        else:
            pa, la = None, None
        mt = np.ascontiguousarray(max_tokens, dtype=np.int32) if max_tokens is not None else None
        self._check(self.lib.msh_stream_decode_full(self.h, n, s.ctypes.data, pa, la,
                                                    mt.ctypes.data if mt is not None else None, toks.ctypes.data,
                                                    counts.ctypes.data, stride, acc.ctypes.data))
        return [toks[i, :counts[i]].tolist() for i in range(n)], acc

    def set_bias(self, children: list[dict] | None, depth: list[int] | None = None, depth_bonus=None):
        """Install (or, with None / an empty trie, remove) the contextual-biasing trie.  children[n] = {token: child
        node} of node n (node 0 = root), depth[n] its depth, depth_bonus[d] the bonus of a depth-d token."""
        if not children or len(children) <= 1:
            self._check(self.lib.msh_stream_set_bias(self.h, 0, None, None, None, None, None, 0))
            return
        off, tok, node = [0], [], []
        for ch in children:
            for t in sorted(ch):
                tok.append(t)
                node.append(ch[t])
            off.append(len(tok))
        a = lambda x, dt: np.ascontiguousarray(x, dtype=dt)
        off, tok, node, dep, bon = a(off, np.int32), a(tok, np.int32), a(node, np.int32), a(depth, np.int32), a(depth_bonus, np.float32)
        self._check(self.lib.msh_stream_set_bias(self.h, len(children), off.ctypes.data, tok.ctypes.data, node.ctypes.data,
                                                 dep.ctypes.data, bon.ctypes.data, bon.shape[0]))

    def profile_enable(self, on: bool = True):
        self._check(self.lib.msh_stream_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.msh_stream_profile_reset(self.h))

    def profile(self) -> list[dict]:
        n = self._check(self.lib.msh_stream_profile_count(self.h))
        out = []
        for i in range(n):
            pe = ProfileEntry()
            self._check(self.lib.msh_stream_profile_get(self.h, i, C.byref(pe)))
            out.append({"name": pe.name.decode(), "ms": pe.ms, "launches": int(pe.launches), "flops": pe.flops, "bytes": pe.bytes})
        return out

    def query(self, slot: int, what: int) -> int:
        return self._check(self.lib.msh_stream_query(self.h, slot, what))

    def memory_len(self, slot: int) -> int:
        return self.query(slot, 0)

    def feature_count(self, slot: int) -> int:
        return self.query(slot, 1)

    def cache_len(self, slot: int) -> int:
        return self.query(slot, 2)

    def max_tokens_for(self, slot: int) -> int:
        return self.query(slot, 4)

    def memory(self, slot: int) -> np.ndarray:
        out = np.empty((self.memory_len(slot), self._info.decoder_dim), np.float32)
        if out.size:
            self._check(self.lib.msh_stream_get_memory(self.h, slot, out.ctypes.data))
        return out

    def features(self, slot: int) -> np.ndarray:
        out = np.empty((self.feature_count(slot), self._info.encoder_dim), np.float32)
        if out.size:
            self._check(self.lib.msh_stream_get_features(self.h, slot, out.ctypes.data))
        return out
