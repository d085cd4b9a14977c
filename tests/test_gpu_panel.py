"""The encoder's A-stationary panel GEMM (k_panel.hip) through its C-ABI test hook, against a plain numpy evaluation of the
same op: LayerNorm (eps 1e-5, scale folded into the weight) -> x W^T -> interleaved-pair RoPE on the q and k sections
(modeling_moonshine.py:132-154, 260-276: rotate pairs (2j, 2j+1) of every head for j < rot_pairs) -> bf16, q | k row-major
[R][2D] and V transposed [D][R].  Operands are rounded to bf16 as the kernel rounds them; tolerance = bf16 output rounding
(2^-8 relative) plus accumulation-order noise."""
import ctypes as C

import numpy as np
import pytest

from moonshine_amd.hip_api import load_dev_library as load_library   # msh_test_*: the development library

pytestmark = pytest.mark.gpu


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def run(R, D, iters=1):
    lib = load_library()
    lib.msh_test_qkv_panel.restype = C.c_float
    lib.msh_test_qkv_panel.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    qk = np.zeros((R, 2 * D), np.uint16)
    vt = np.zeros((D, R), np.uint16)
    h = np.zeros((R, D), np.float32)
    w = np.zeros((3 * D, D), np.float32)
    pos = np.zeros(R, np.int32)
    ms = lib.msh_test_qkv_panel(R, D, iters, qk.ctypes.data, vt.ctypes.data, h.ctypes.data, w.ctypes.data, pos.ctypes.data)
    assert ms >= 0
    return ms, bf16_bits_to_f32(qk), bf16_bits_to_f32(vt), h, w, pos


def reference(h, w, pos, D):
    DH, RP = (52, 23) if D == 416 else (36, 16)
    x = h.astype(np.float64)
    y = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    y = bf16_round(y.astype(np.float32)).astype(np.float64)
    out = y @ bf16_round(w).astype(np.float64).T                       # [R][3D]
    p = np.maximum(pos, 0).astype(np.float64)
    inv = 1.0 / np.power(10000.0, (2.0 * np.arange(RP)) / (2.0 * RP))
    ang = (p[:, None] * inv[None, :]).astype(np.float32)                # the table is built in fp32
    cos, sin = np.cos(ang).astype(np.float64), np.sin(ang).astype(np.float64)
    qk = out[:, :2 * D].copy().reshape(-1, 2 * D // DH, DH)
    for j in range(RP):
        a, b = qk[:, :, 2 * j].copy(), qk[:, :, 2 * j + 1].copy()
        qk[:, :, 2 * j] = a * cos[:, None, j] - b * sin[:, None, j]
        qk[:, :, 2 * j + 1] = b * cos[:, None, j] + a * sin[:, None, j]
    return qk.reshape(-1, 2 * D), out[:, 2 * D:].T


@pytest.mark.parametrize("R,D", [(128, 416), (8, 416), (1000, 416), (3336, 288), (264, 288)])
def test_qkv_panel_matches_numpy(R, D):
    _, qk, vt, h, w, pos = run(R, D)
    want_qk, want_vt = reference(h, w, pos, D)
    for got, want, name in ((qk, want_qk, "q|k"), (vt, want_vt, "v^T")):
        assert np.isfinite(got).all(), name
        err = np.abs(got - want)
        tol = 2.0 ** -7 * np.abs(want) + 2e-3 * np.abs(want).max()
        bad = err > tol
        assert not bad.any(), (name, int(bad.sum()), float(err.max()), np.argwhere(bad)[:5].tolist())
    assert float(np.abs(want_qk).max()) > 0.5      # the comparison is not about zeros


def test_qkv_panel_speed_report(capsys):
    R, D = 256 * 416, 416
    ms, *_ = run(R, D, iters=20)
    flops = 2.0 * R * D * 3 * D
    with capsys.disabled():
        print(f"\n[qkv panel] R = {R}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s = {flops / ms / 1e9 / 2500:.3f} of the MFMA peak "
              f"(tiled: 0.33 ms + 0.048 ms LayerNorm)")


@pytest.mark.parametrize("arch", ["base", "tiny"])
def test_encoder_on_the_qkv_panel_kernel_vs_oracle(tmp_path_factory, monkeypatch, arch):
    """The whole encoder with the panel kernels (QKV panel, fused o-proj + MLP) forced on at a small ragged batch (they are the default from 16 k / 32 k rows on,
    where tests/test_gpu_parity.py::test_base_batch256_benchmark_path_vs_oracle runs it): last_hidden_state against the
    ORACLE at the stated encoder tolerance, and against the tiled-GEMM path of the same engine."""
    from oracle import moonshine_ref as ref
    from oracle.weights import make_audio
    from test_gpu_parity import _enc_check, _engine

    e, w, cfg = _engine(tmp_path_factory, arch, 3)
    lens = [160000, 48000, 159744, 100000, 7000, 31999]
    clips = [make_audio(70 + i, n) for i, n in enumerate(lens)]
    e.set_keep_encoder_output(True)
    monkeypatch.setenv("MSH_ENC_QKV_PANEL", "2")
    monkeypatch.setenv("MSH_ENC_MLP", "3")          # and the fused o-proj + MLP kernel at any size
    e.encode(clips)
    panel = [e.encoder_output(i).copy() for i in range(len(clips))]
    monkeypatch.setenv("MSH_ENC_QKV_PANEL", "0")
    monkeypatch.setenv("MSH_ENC_MLP", "0")
    e.encode(clips)
    tiled = [e.encoder_output(i).copy() for i in range(len(clips))]
    if arch == "base":
        # MSH_ENC_MLP=2: the MLP block alone in the fused kernel (its stages follow the o-proj stages in the packed weights);
        # it only engages from 32 k rows on, so this batch is 80 x 10 s
        big = [make_audio(200 + i, 160000) for i in range(80)]
        e.set_keep_encoder_output(True)
        monkeypatch.setenv("MSH_ENC_MLP", "2")
        e.encode(big)
        a = [e.encoder_output(i).copy() for i in (0, 41, 79)]
        monkeypatch.setenv("MSH_ENC_MLP", "1")
        e.encode(big)
        b = [e.encoder_output(i).copy() for i in (0, 41, 79)]
        for x, y in zip(a, b):
            _enc_check(x, y)
        _enc_check(a[1], ref.encoder_forward(w, cfg, big[41]))
    worst = 0.0
    for i in (0, 1, 4, 5):
        want = ref.encoder_forward(w, cfg, clips[i])
        r, _ = _enc_check(panel[i], want)
        worst = max(worst, r)
    for a, b in zip(panel, tiled):
        assert a.shape == b.shape
        _enc_check(a, b)
    print(f"[{arch}] encoder with the QKV panel kernel: rel-RMS {worst:.2e} against the oracle")


@pytest.mark.parametrize("arch,kv", [("base", "bf16"), ("base", "fp8"), ("tiny", "bf16")])
def test_cross_kv_on_the_panel_kernel_vs_oracle(tmp_path_factory, monkeypatch, arch, kv):
    """Cross-attention K^T / V^T of all decoder layers written by the panel kernel (forced on at a small ragged batch; the
    default from 16 k rows on, where the batch-256 oracle tests run it): teacher-forced decoder logits against the ORACLE at
    the stated tolerance, and the K^T / V^T buffers themselves against the same engine's tiled cross-KV GEMM (the same
    products summed in a different order: equal up to a bf16 / e4m3 step on a small fraction of the elements, zero padding
    keys at the same places)."""
    from oracle import moonshine_ref as ref
    from oracle.weights import make_audio
    from test_gpu_parity import LOGIT_MAXABS, _engine, _teacher_logit_check

    monkeypatch.setenv("MSH_ENC_CROSS_KV_PANEL", "2")     # (read at load as well: the packed weight is only uploaded on request)
    e, w, cfg = _engine(tmp_path_factory, arch, 5, dev=True)   # (debug_read: the development library's hook)
    if kv == "fp8":
        e.set_kv_dtype("fp8")
    clips = [make_audio(90 + i, n) for i, n in enumerate([160000, 52000, 159744, 3000, 100000])]
    monkeypatch.setenv("MSH_ENC_CROSS_KV_PANEL", "2")
    _teacher_logit_check(e, w, cfg, [clips[1], clips[4]], 6)
    e.encode(clips)
    kp, vp = e.debug_read("cross_k").copy(), e.debug_read("cross_v").copy()
    monkeypatch.setenv("MSH_ENC_CROSS_KV_PANEL", "0")
    e.encode(clips)
    kt, vt = e.debug_read("cross_k").copy(), e.debug_read("cross_v").copy()
    assert kp.shape == kt.shape and kp.size > 0
    if kv == "bf16":
        f = lambda b: (b.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
        for a, b in ((f(kp), f(kt)), (f(vp), f(vt))):
            assert np.isfinite(a).all()
            # the same products summed in a different order, rounded to bf16: equal up to one bf16 step on a small fraction
            assert float(np.abs(a - b).max()) <= 2.0 ** -6 * max(1.0, float(np.abs(b).max())), float(np.abs(a - b).max())
            assert float((a != b).mean()) < 0.05
            assert ((a == 0) == (b == 0)).all()          # the zero padding keys sit at the same places
    else:
        assert float((kp != kt).mean()) < 0.05 and float((vp != vt).mean()) < 0.05      # e4m3 bytes
