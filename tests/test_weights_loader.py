"""Checkpoint loading without a GPU (msh_host_check_weights, include/moonshine_hip.h: the validation half of
msh_load_weights_*): the variants a real `UsefulSensors/moonshine-*` safetensors file can come in -- fp32 / fp16 / bf16
tensors, with or without the tied-head alias `proj_out.weight` -- and the files that must be refused with a reason: an
untied head, a tensor the architecture does not have, a wrong shape, a missing tensor, an integer dtype, the wrong
architecture id.  (Reference: the ORT loader fails on any graph / initializer mismatch, core/moonshine-model.cpp:145-186;
SURVEY.md 8f.1 makes HF-named safetensors this engine's native format.)"""
import ctypes as C
import json
import struct

import numpy as np
import pytest

from moonshine_amd.hip_api import ModelInfo, load_library
from moonshine_amd.synth import ARCHS, make_weights


def _bf16_bits(a: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)   # round to nearest even


def _blob(tensors: dict, dtype: str = "F32", overrides: dict | None = None) -> bytes:
    """safetensors bytes with every float tensor stored as `dtype` (overrides: name -> (dtype, raw ndarray))."""
    header, chunks, off = {}, [], 0
    for name, arr in tensors.items():
        dt, raw = dtype, None
        if overrides and name in overrides:
            dt, raw = overrides[name]
        if raw is None:
            raw = {"F32": lambda x: x.astype("<f4"), "F16": lambda x: x.astype("<f2"), "BF16": _bf16_bits}[dt](arr)
        b = np.ascontiguousarray(raw).tobytes()
        header[name] = {"dtype": dt, "shape": list(arr.shape), "data_offsets": [off, off + len(b)]}
        chunks.append(b)
        off += len(b)
    header["__metadata__"] = {"arch": "micro", "heads": "4"}
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    return struct.pack("<Q", len(hj)) + hj + b"".join(chunks)


def _check(blob: bytes, arch: int = -1):
    lib = load_library()
    lib.msh_host_check_weights.restype = C.c_int32
    lib.msh_host_check_weights.argtypes = [C.c_char_p, C.c_uint64, C.c_int32, C.POINTER(ModelInfo), C.c_char_p, C.c_uint64]
    mi, err = ModelInfo(), C.create_string_buffer(1024)
    rc = lib.msh_host_check_weights(blob, len(blob), arch, C.byref(mi), err, len(err))
    return rc, mi, err.value.decode(errors="replace")


@pytest.fixture(scope="module")
def w():
    return make_weights(ARCHS["micro"], 0)


@pytest.mark.parametrize("dtype", ["F32", "F16", "BF16"])
def test_float_dtypes_load(w, dtype):
    rc, mi, err = _check(_blob(w, dtype))
    assert rc == 0, err
    assert (mi.hidden, mi.ffn, mi.enc_layers, mi.dec_layers, mi.heads, mi.vocab) == (64, 256, 2, 2, 4, 512)


def test_mixed_dtypes_and_tied_head_alias(w):
    mixed = {"model.decoder.embed_tokens.weight": ("BF16", None), "model.encoder.conv1.weight": ("F16", None)}
    assert _check(_blob(w, "F32", mixed))[0] == 0
    tied = dict(w)
    tied["proj_out.weight"] = w["model.decoder.embed_tokens.weight"].copy()      # the alias HF writes when it does not dedupe
    rc, _, err = _check(_blob(tied))
    assert rc == 0, err
    tied["model.decoder.layers.0.self_attn.rotary_emb.inv_freq"] = np.ones(7, np.float32)   # a persisted position buffer
    assert _check(_blob(tied))[0] == 0


def test_untied_head_is_refused(w):
    untied = dict(w)
    h = w["model.decoder.embed_tokens.weight"].copy()
    h[3, 5] += 1.0
    untied["proj_out.weight"] = h
    rc, _, err = _check(_blob(untied))
    assert rc != 0 and "proj_out.weight" in err and "tied" in err


def test_unexpected_tensor_is_named(w):
    extra = dict(w)
    extra["model.encoder.layers.0.self_attn.q_proj.bias"] = np.zeros(64, np.float32)   # Moonshine attention has no bias
    rc, _, err = _check(_blob(extra))
    assert rc != 0 and "q_proj.bias" in err


def test_missing_tensor_wrong_shape_integer_dtype_wrong_arch(w):
    missing = {k: v for k, v in w.items() if k != "model.decoder.layers.1.mlp.fc2.bias"}
    rc, _, err = _check(_blob(missing))
    assert rc != 0 and "fc2.bias" in err
    bad = dict(w)
    bad["model.encoder.layers.0.mlp.fc1.bias"] = np.zeros(255, np.float32)
    rc, _, err = _check(_blob(bad))
    assert rc != 0 and "fc1.bias" in err
    ints = {"model.decoder.norm.weight": ("I64", np.ones(64, np.int64))}
    rc, _, err = _check(_blob(w, "F32", ints))
    assert rc != 0 and "I64" in err and "norm.weight" in err
    rc, _, err = _check(_blob(w), arch=1)       # MOONSHINE_MODEL_ARCH_BASE wants hidden 416
    assert rc != 0 and "BASE" in err and "64" in err
    assert _check(b"\x00" * 16)[0] != 0 and _check(b"")[0] != 0
