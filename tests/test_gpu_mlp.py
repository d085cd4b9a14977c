"""The fused encoder MLP kernel (moonshine_amd/csrc/k_mlp.hip: LayerNorm + fc1 + GELU + fc2 + residual in one launch) alone,
against the oracle's restatement of the block (oracle/moonshine_ref.py::encoder_forward, hf modeling_moonshine.py:69-85 and
:382-411) on the same random inputs: every supported hidden size, row counts that are not multiples of the 128-row panel
or of a wave's 32 rows, rows with a large mean (the single-pass moments must not cancel), one and many panels.
Tolerance: bf16 operands, fp32 accumulate -- rel-RMS <= 6e-3 and max-abs <= 6e-2 on the block's output."""
import ctypes as C

import numpy as np
import pytest

from moonshine_amd.hip_api import load_dev_library as load_library   # msh_test_*: the development library
from oracle import moonshine_ref as ref

pytestmark = pytest.mark.gpu


def _run(h, w1, g, b1, w2, b2):
    lib = load_library()
    fp = C.POINTER(C.c_float)
    lib.msh_test_mlp_run.restype = C.c_int32
    lib.msh_test_mlp_run.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, fp, fp]
    out = np.ascontiguousarray(h, np.float32).copy()
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w1, g, b1, w2, b2)]
    R, D = out.shape
    rc = lib.msh_test_mlp_run(out.ctypes.data_as(fp), R, D, w1.shape[0], *[a.ctypes.data_as(fp) for a in arrs])
    assert rc == 0
    return out


@pytest.mark.parametrize("D,F,R", [(64, 256, 40), (64, 32, 1), (64, 256, 129), (288, 1152, 300), (416, 1664, 424), (416, 1664, 1000), (416, 64, 33)])
def test_fused_mlp_block_vs_oracle(D, F, R):
    rng = np.random.default_rng(D + F + R)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    b1 = (rng.standard_normal(F) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    h = rng.standard_normal((R, D)).astype(np.float32) * 2.0
    h[:: 7] += 300.0          # rows whose mean dwarfs their spread
    h[3 % R] *= 1e-3          # and a tiny one
    got = _run(h, w1, g, b1, w2, b2)
    y = ref.layer_norm_nobias(h, g)
    want = (h + ref.gelu(y @ w1.T + b1) @ w2.T + b2).astype(np.float32)
    err = got - want
    blk = want - h             # compare on the block's own output (the residual passes through exactly)
    relrms = float(np.sqrt((err ** 2).mean()) / np.sqrt((blk ** 2).mean()))
    assert relrms <= 6e-3, relrms
    assert float(np.abs(err).max()) <= 6e-2
    assert np.isfinite(got).all()


def _run_o(h, w1, g, b1, w2, b2, ao, wo):
    lib = load_library()
    fp = C.POINTER(C.c_float)
    lib.msh_test_mlp_oproj_run.restype = C.c_int32
    lib.msh_test_mlp_oproj_run.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32] + [fp] * 7
    out = np.ascontiguousarray(h, np.float32).copy()
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w1, g, b1, w2, b2, ao, wo)]
    R, D = out.shape
    rc = lib.msh_test_mlp_oproj_run(out.ctypes.data_as(fp), R, D, w1.shape[0], *[a.ctypes.data_as(fp) for a in arrs])
    assert rc == 0
    return out


@pytest.mark.parametrize("D,F,R", [(64, 256, 40), (64, 256, 129), (288, 1152, 300), (416, 1664, 424), (416, 1664, 1000), (416, 64, 33)])
def test_fused_oproj_mlp_block_vs_oracle(D, F, R):
    """The same kernel with the attention output projection in front (hf modeling_moonshine.py:382-411: hidden = residual +
    o_proj(attn); hidden = hidden + mlp(LayerNorm(hidden))): H' = H + AO Wo^T formed in the accumulators, LayerNorm taken from
    those registers.  The attention output is compared as the kernel sees it (bf16)."""
    rng = np.random.default_rng(7 * D + F + R)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (rng.standard_normal(F) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    h = rng.standard_normal((R, D)).astype(np.float32) * 2.0
    h[:: 7] += 300.0
    ao = rng.standard_normal((R, D)).astype(np.float32)
    got = _run_o(h, w1, g, b1, w2, b2, ao, wo)
    u = np.ascontiguousarray(ao, np.float32).view(np.uint32).astype(np.uint64)
    ao16 = (((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(np.float32)
    h1 = h + ao16 @ wo.T
    y = ref.layer_norm_nobias(h1, g)
    want = (h1 + ref.gelu(y @ w1.T + b1) @ w2.T + b2).astype(np.float32)
    err = got - want
    blk = want - h
    relrms = float(np.sqrt((err ** 2).mean()) / np.sqrt((blk ** 2).mean()))
    assert relrms <= 6e-3, relrms
    assert float(np.abs(err).max()) <= 8e-2     # two bf16 GEMM chains in a row
    assert np.isfinite(got).all()


@pytest.mark.parametrize("D,F,R", [(64, 256, 129), (288, 1152, 300), (416, 1664, 424), (416, 64, 33)])
def test_fused_block_hands_the_next_layers_layernorm_over(D, F, R):
    """Round 6: the kernel's second output -- LayerNorm (no scale, eps 1e-5) of the rows it just produced, bf16, in the
    fragment-major order the next layer's QKV panel kernel loads as its MFMA operand (k_mlp.hip YOUT, k_panel.hip AM = 2;
    hf modeling_moonshine.py:382-411: the next layer's input_layernorm on this layer's output).  Checked against LayerNorm of
    the kernel's OWN fp32 output (same values the panel kernel used to re-read), de-interleaved by the documented layout."""
    rng = np.random.default_rng(11 * D + F + R)
    w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
    w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
    wo = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    b1 = (rng.standard_normal(F) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    h = rng.standard_normal((R, D)).astype(np.float32) * 2.0
    h[:: 7] += 300.0
    ao = rng.standard_normal((R, D)).astype(np.float32)
    lib = load_library()
    fp = C.POINTER(C.c_float)
    out = h.copy()
    Rp = (R + 127) // 128 * 128
    y = np.zeros(Rp * D, np.uint16)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w1, g, b1, w2, b2, ao, wo)]
    rc = lib.msh_test_mlp_oproj_y_run(out.ctypes.data, R, D, F, *[a.ctypes.data for a in arrs], y.ctypes.data)
    assert rc == 0
    np.testing.assert_array_equal(out, _run_o(h, w1, g, b1, w2, b2, ao, wo))     # the residual stream itself: unchanged bits
    KS = D // 16
    yf = (y.astype(np.uint32) << 16).view(np.float32).reshape(Rp // 32, KS, 64, 8)
    rows = np.empty((Rp, D), np.float32)
    for lane in range(64):
        for s in range(KS):
            rows[np.arange(Rp // 32) * 32 + (lane & 31), 16 * s + 8 * (lane >> 5):16 * s + 8 * (lane >> 5) + 8] = yf[:, s, lane, :]
    want = ref.layer_norm_nobias(out.astype(np.float64), np.ones(D)).astype(np.float32)
    err = np.abs(rows[:R] - want)
    assert float(err.max()) <= 2e-2, float(err.max())          # bf16 of O(1) normalised values (<= 4 sigma: 2^-6 steps)
    assert float(np.sqrt((err ** 2).mean())) <= 3e-3
