"""Encoder self-attention kernels alone (k_attn.hip, through the development library's msh_test_enc_attention): the
LDS-resident kernel of round 6 (one 8-wave workgroup per (clip, head), keys and values staged once) against the block-streaming
kernel it replaces for clips of 257..448 frames, and both against a float64 softmax(q k^T / sqrt(dh)) v of the same bf16 inputs
(modeling_moonshine.py:171-193 eager_attention_forward, non-causal, the encoder graph the reference runs at
core/moonshine-model.cpp:270-274).  The queries reach the kernels PRE-SCALED by rsqrt(dh) * log2(e) (folded into the q
projection at load), so the reference is softmax_2(q' k^T) = 2^(s - max) / sum."""
import numpy as np
import pytest

from moonshine_amd.hip_api import load_dev_library

pytestmark = pytest.mark.gpu

D, H, DH = 416, 8, 52


def _bf16(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _inputs(n_clips, T):
    """The hook's own generator (xorshift32 -> uniform [-1, 1), q | k scaled by 2), restated: q | k [R][2D], V^T [D][ld]."""
    rows = (T + 7) // 8 * 8
    R = rows * n_clips
    ld = (R + 127) // 128 * 128
    n = R * 2 * D + D * ld
    x = np.uint32(2463534242)
    vals = np.empty(n, np.float32)
    with np.errstate(over="ignore"):
        for i in range(n):
            x ^= np.uint32(x << np.uint32(13))
            x ^= np.uint32(x >> np.uint32(17))
            x ^= np.uint32(x << np.uint32(5))
            vals[i] = np.float32(int(x) >> 8) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    qk = vals[: R * 2 * D].reshape(R, 2 * D) * np.float32(2.0)
    qk[:, :D] *= np.float32(np.float32(1.0) / np.sqrt(np.float32(DH)) * np.float32(1.4426950408889634))   # the hook's `qscale`
    vt = vals[R * 2 * D:].reshape(D, ld)
    rnd = lambda a: _bf16(((a.view(np.uint32) + 0x7FFF + ((a.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16))  # noqa: E731
    return rnd(np.ascontiguousarray(qk)), rnd(np.ascontiguousarray(vt)), rows


@pytest.mark.parametrize("n_clips,T", [(2, 415), (1, 257), (3, 448), (2, 383)])
def test_resident_kernel_equals_streaming_kernel_and_float64(n_clips, T):
    lib = load_dev_library()
    rows = (T + 7) // 8 * 8
    outs = {}
    for v in (0, 1, 100):
        o = np.zeros((n_clips * rows, D), np.uint16)
        assert lib.msh_test_enc_attention(v, n_clips, T, D, H, 0, o.ctypes.data) >= 0
        outs[v] = o
    assert (outs[1] == outs[0]).all()                      # the streaming kernel's two shapes: identical bits
    d = np.abs(_bf16(outs[100]).astype(np.float64) - _bf16(outs[0]).astype(np.float64))
    assert d.max() < 8e-3, d.max()                         # resident vs streaming: other rounding points, a bf16 ulp of O(1) outputs
    qk, vt, rows = _inputs(n_clips, T)
    got = _bf16(outs[100]).astype(np.float64)
    worst = 0.0
    for b in range(n_clips):
        r0 = b * rows
        for h in range(H):
            q = qk[r0:r0 + T, h * DH:(h + 1) * DH].astype(np.float64)
            k = qk[r0:r0 + T, D + h * DH:D + (h + 1) * DH].astype(np.float64)
            v = vt[h * DH:(h + 1) * DH, r0:r0 + T].astype(np.float64).T
            s = q @ k.T
            p = np.exp2(s - s.max(-1, keepdims=True))
            want = (p / p.sum(-1, keepdims=True)) @ v
            worst = max(worst, float(np.abs(got[r0:r0 + T, h * DH:(h + 1) * DH] - want).max()))
        assert (outs[100][r0 + T:r0 + rows] == 0).all()      # the clip's padding rows are written as zeros
    assert worst < 2e-2, worst      # bf16 probabilities and outputs of O(1) values
