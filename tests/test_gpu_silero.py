"""Silero VAD on the device (msh_silero_*, k_silero.hip) against the ORACLE (oracle/silero_ref.py::SileroRef, the numpy
restatement of the network behind reference core/silero-vad.cpp:78-173, itself pinned on an independent torch build in
tests/test_silero_vad.py) and, as a second assert, against the library's host implementation
(msh_host_silero_probabilities): the probability of every hop of every clip, each clip from a fresh state, ragged
lengths incl. clips shorter than one hop."""
import ctypes as C
import os

import numpy as np
import pytest

import margins
from moonshine_amd.hip_api import load_library
from moonshine_amd.synth import make_audio, make_silero_weights, save_safetensors
from oracle.silero_ref import SileroRef

pytestmark = pytest.mark.gpu
TOL = 1e-4   # the bound the host code is held to against the oracle


@pytest.fixture(scope="module")
def blob(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("silero") / "silero_vad.safetensors")
    save_safetensors(p, make_silero_weights(2))
    return open(p, "rb").read()


def _lib():
    lib = load_library()
    lib.msh_silero_create.restype = C.c_int32
    lib.msh_silero_create.argtypes = [C.c_int32, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.msh_silero_destroy.argtypes = [C.c_void_p]
    lib.msh_silero_probabilities.restype = C.c_int64
    lib.msh_silero_probabilities.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64), C.c_uint64,
                                             C.POINTER(C.c_float), C.c_uint64]
    lib.msh_silero_last_error.restype = C.c_char_p
    lib.msh_silero_last_error.argtypes = [C.c_void_p]
    lib.msh_host_silero_probabilities.restype = C.c_int64
    lib.msh_host_silero_probabilities.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float),
                                                  C.c_uint64, C.POINTER(C.c_float)]
    return lib


def _device_probs(lib, h, clips):
    n = len(clips)
    ptrs = (C.POINTER(C.c_float) * n)(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in clips])
    lens = (C.c_uint64 * n)(*[c.shape[0] for c in clips])
    total = sum(c.shape[0] // 512 for c in clips)
    out = np.zeros(max(total, 1), np.float32)
    got = lib.msh_silero_probabilities(h, ptrs, lens, n, out.ctypes.data_as(C.POINTER(C.c_float)), total)
    assert got == total, lib.msh_silero_last_error(h)
    res, off = [], 0
    for c in clips:
        k = c.shape[0] // 512
        res.append(out[off:off + k].copy())
        off += k
    return res


def _host_probs(lib, blob, clip):
    k = clip.shape[0] // 512
    out = np.zeros(max(k, 1), np.float32)
    got = lib.msh_host_silero_probabilities(blob, len(blob), clip.ctypes.data_as(C.POINTER(C.c_float)), clip.shape[0],
                                            out.ctypes.data_as(C.POINTER(C.c_float)), k, None)
    assert got == k
    return out[:k]


def _oracle_probs(clip):
    """SileroRef hop by hop from a fresh state (what SileroVad::predict returns, reference core/silero-vad.cpp:78-173)."""
    net = SileroRef(make_silero_weights(2))
    return np.asarray([net.predict(clip[i * 512:(i + 1) * 512]) for i in range(clip.shape[0] // 512)], np.float32)


def test_device_probabilities_match_the_oracle_and_the_host_network(blob):
    lib = _lib()
    h = C.c_void_p()
    assert lib.msh_silero_create(0, blob, len(blob), C.byref(h)) == 0
    lens = [160000, 512, 300, 0, 1023, 16000 * 3 + 77, 52000, 160000, 8192]
    clips = [np.ascontiguousarray(make_audio(400 + i, max(n, 1))[:n], dtype=np.float32) for i, n in enumerate(lens)]
    dev = _device_probs(lib, h, clips)
    worst = worst_host = 0.0
    for c, d in zip(clips, dev):
        want = _oracle_probs(c)                      # HIP path vs the oracle: the parity claim
        assert d.shape == want.shape
        host = _host_probs(lib, blob, c)             # and vs the product's host code (streams use that one)
        assert d.shape == host.shape
        if len(want):
            assert np.isfinite(d).all()
            worst = max(worst, float(np.abs(d - want).max()))
            worst_host = max(worst_host, float(np.abs(d - host).max()))
    margins.record(device_vs_oracle_max_abs=worst, device_vs_host_max_abs=worst_host, tolerance=TOL)
    assert worst < TOL, worst
    assert worst_host < TOL, worst_host
    # the probabilities move (a constant output would pass a lazy tolerance on a saturated network)
    assert float(np.std(dev[0])) > 1e-3
    # a second call reuses the workspace: same result
    again = _device_probs(lib, h, clips)
    for a, b in zip(dev, again):
        assert np.array_equal(a, b)
    lib.msh_silero_destroy(h)


def test_device_silero_throughput_report(blob, capsys):
    import time

    lib = _lib()
    h = C.c_void_p()
    assert lib.msh_silero_create(0, blob, len(blob), C.byref(h)) == 0
    clips = [np.ascontiguousarray(make_audio(500 + (i % 16), 160000), dtype=np.float32) for i in range(512)]
    _device_probs(lib, h, clips)
    t = time.perf_counter()
    _device_probs(lib, h, clips)
    dt = time.perf_counter() - t
    with capsys.disabled():
        print(f"\n[device Silero] 512 clips x 10 s: {dt * 1e3:.1f} ms = {512 * 10 / dt:.0f} audio-s/s")
    lib.msh_silero_destroy(h)
