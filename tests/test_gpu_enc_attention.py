"""Encoder self-attention kernel alone (k_attn.hip enc_attention_res_kernel, through the development library's
msh_test_enc_attention): keys and values of a (clip, head) resident in LDS, one chunk of 448 keys at a time.  Checked against a
float64 softmax(q k^T / sqrt(dh)) v of the same bf16 inputs (modeling_moonshine.py:171-193 eager_attention_forward, non-causal;
the encoder graph the reference runs at core/moonshine-model.cpp:270-274) on one-chunk clips, clips of several chunks and
several workgroups per (clip, head), and: the instantiation with the chunk loop (batches holding a clip of more than 448
frames) gives a short clip the SAME BITS as the one without -- a clip's encoder output never depends on its batch's longest
clip.  The queries reach the kernel PRE-SCALED by rsqrt(dh) * log2(e) (folded into the q projection at load), so the reference
is softmax_2(q' k^T) = 2^(s - max) / sum."""
import numpy as np
import pytest

from moonshine_amd.hip_api import load_dev_library

pytestmark = pytest.mark.gpu

D, H, DH = 416, 8, 52


def _bf16(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _inputs(n_clips, T):
    """The hook's own generator (xorshift32 -> uniform [-1, 1), q | k scaled by 2, q by the attention scale), restated:
    q | k [R][2D], V^T [D][ld]."""
    rows = (T + 7) // 8 * 8
    R = rows * n_clips
    ld = (R + 127) // 128 * 128
    n = R * 2 * D + D * ld
    x = 2463534242
    vals = np.empty(n, np.float32)
    for i in range(n):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        vals[i] = np.float32(x >> 8) * np.float32(1.0 / 8388608.0) - np.float32(1.0)
    qk = vals[: R * 2 * D].reshape(R, 2 * D) * np.float32(2.0)
    qk[:, :D] *= np.float32(np.float32(1.0) / np.sqrt(np.float32(DH)) * np.float32(1.4426950408889634))   # the hook's `qscale`
    vt = vals[R * 2 * D:].reshape(D, ld)
    rnd = lambda a: _bf16(((a.view(np.uint32) + 0x7FFF + ((a.view(np.uint32) >> 16) & 1)) >> 16).astype(np.uint16))  # noqa: E731
    return rnd(np.ascontiguousarray(qk)), rnd(np.ascontiguousarray(vt)), rows


def _run(variant, n_clips, T):
    lib = load_dev_library()
    rows = (T + 7) // 8 * 8
    o = np.zeros((n_clips * rows, D), np.uint16)
    assert lib.msh_test_enc_attention(variant, n_clips, T, D, H, 0, o.ctypes.data) >= 0
    return o


@pytest.mark.parametrize("n_clips,T", [(2, 415), (1, 257), (3, 448), (2, 383), (3, 41), (1, 64), (1, 900), (1, 1353)])
def test_encoder_attention_vs_float64(n_clips, T):
    out = _run(1, n_clips, T)                       # the instantiation that takes any length
    if T <= 448:
        assert (_run(0, n_clips, T) == out).all()   # ... and the one batches of short clips run: identical bits
    qk, vt, rows = _inputs(n_clips, T)
    got = _bf16(out).astype(np.float64)
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
        assert (out[r0 + T:r0 + rows] == 0).all()      # the clip's padding rows are written as zeros
    assert worst < 2e-2, worst      # bf16 probabilities and outputs of O(1) values
