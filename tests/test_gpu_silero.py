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


def test_keep_audio_leaves_the_clips_whole_hops_on_the_device(blob):
    """msh_silero_probabilities_keep_audio: the same probabilities, and device pointers to every clip's whole hops -- the
    caller's samples verbatim (read back with hipMemcpy) --, over more than one chunk of the network (64 Ki hops), NULL for a clip
    without a whole hop; after msh_silero_release_audio the next call may reuse the buffers and still returns its own audio."""
    lib = _lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.msh_silero_probabilities_keep_audio.restype = C.c_int64
    lib.msh_silero_probabilities_keep_audio.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64), C.c_uint64,
                                                        C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_void_p)]
    lib.msh_silero_release_audio.restype = C.c_int32
    lib.msh_silero_release_audio.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert lib.msh_silero_create(0, blob, len(blob), C.byref(h)) == 0

    def run(clips):
        n = len(clips)
        ptrs = (C.POINTER(C.c_float) * n)(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in clips])
        lens = (C.c_uint64 * n)(*[c.shape[0] for c in clips])
        total = sum(c.shape[0] // 512 for c in clips)
        out = np.zeros(max(total, 1), np.float32)
        dev = (C.c_void_p * n)()
        got = lib.msh_silero_probabilities_keep_audio(h, ptrs, lens, n, out.ctypes.data_as(C.POINTER(C.c_float)), total, dev)
        assert got == total, lib.msh_silero_last_error(h)
        for c, d in zip(clips, dev):
            whole = c.shape[0] // 512 * 512
            if whole == 0:
                assert not d
                continue
            assert d and d % 4 == 0
            back = np.empty(whole, np.float32)
            assert hip.hipMemcpy(back.ctypes.data, d, whole * 4, 2) == 0
            assert np.array_equal(back, c[:whole])
        return out[:total], [int(d or 0) for d in dev]

    # 230 clips of 10 s = 71.9 Ki hops: two chunks; ragged ones in front
    lens = [160000, 512, 300, 0, 1023, 16000 * 3 + 77] + [160000] * 224
    clips = [np.ascontiguousarray(make_audio(430 + (i % 24), max(n, 1))[:n], dtype=np.float32) for i, n in enumerate(lens)]
    probs, where = run(clips)
    plain = np.concatenate(_device_probs(lib, h, clips))
    assert np.array_equal(probs, plain)
    assert lib.msh_silero_release_audio(h) == 0
    probs2, where2 = run(clips[:40][::-1])
    assert where2[-1] != 0 and set(where2) & set(where)            # buffers are reused after the release
    assert np.array_equal(probs2, np.concatenate(_device_probs(lib, h, clips[:40][::-1])))
    # the two halves (msh_silero_submit / _collect), two submissions outstanding: the same probabilities and the same audio
    lib.msh_silero_submit.restype = C.c_int64
    lib.msh_silero_submit.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64), C.c_uint64, C.c_int32]
    lib.msh_silero_collect.restype = C.c_int64
    lib.msh_silero_collect.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_void_p), C.c_uint64]
    assert lib.msh_silero_release_audio(h) == 0
    groups = [clips[:6], clips[6:100], clips[100:206]]
    keep = []

    def submit(g):
        ptrs = (C.POINTER(C.c_float) * len(g))(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in g])
        lens = (C.c_uint64 * len(g))(*[c.shape[0] for c in g])
        keep.append((ptrs, lens))
        t = lib.msh_silero_submit(h, ptrs, lens, len(g), 1)
        assert t >= 0, lib.msh_silero_last_error(h)
        return t

    def collect(t, g):
        total = sum(c.shape[0] // 512 for c in g)
        out = np.zeros(max(total, 1), np.float32)
        dev = (C.c_void_p * len(g))()
        assert lib.msh_silero_collect(h, t, out.ctypes.data_as(C.POINTER(C.c_float)), total, dev, len(g)) == total, lib.msh_silero_last_error(h)
        for c, d in zip(g, dev):
            whole = c.shape[0] // 512 * 512
            if whole:
                back = np.empty(whole, np.float32)
                assert hip.hipMemcpy(back.ctypes.data, d, whole * 4, 2) == 0
                assert np.array_equal(back, c[:whole])
        return out[:total]

    # 16-bit PCM (msh_silero_submit_pcm16): two bytes per sample over PCIe, x / 32768 on the device -- the probabilities and the
    # kept fp32 audio of the float submission of those values, bit for bit
    lib.msh_silero_submit_pcm16.restype = C.c_int64
    lib.msh_silero_submit_pcm16.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.c_uint64), C.c_uint64, C.c_int32]
    g16 = [np.clip(np.round(c * 6000.0), -32768, 32767).astype(np.int16) for c in clips[:40]]
    gf = [(c.astype(np.float32) / np.float32(32768.0)).astype(np.float32) for c in g16]
    p16 = (C.POINTER(C.c_int16) * len(g16))(*[c.ctypes.data_as(C.POINTER(C.c_int16)) for c in g16])
    l16 = (C.c_uint64 * len(g16))(*[c.shape[0] for c in g16])
    t16 = lib.msh_silero_submit_pcm16(h, p16, l16, len(g16), 1)
    assert t16 >= 0, lib.msh_silero_last_error(h)
    probs16 = collect(t16, gf)                  # (checks the kept audio against the float values)
    probs_f = collect(submit(gf), gf)
    assert np.array_equal(probs16, probs_f) and float(np.std(probs16)) > 1e-4

    t0, t1 = submit(groups[0]), submit(groups[1])
    assert lib.msh_silero_submit(h, keep[0][0], keep[0][1], 6, 1) < 0          # a third one is refused
    assert lib.msh_silero_collect(h, t1, None, 0, None, 0) < 0                  # out of order
    got = [collect(t0, groups[0])]
    t2 = submit(groups[2])
    got += [collect(t1, groups[1]), collect(t2, groups[2])]
    assert np.array_equal(np.concatenate(got), probs[:sum(len(g) for g in got)])
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
