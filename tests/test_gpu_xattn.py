"""Absorbed cross-attention (moonshine_amd/csrc/k_xattn.hip, msh_set_cross_mode): the decoder attends over the ENCODER
OUTPUT itself -- the key projection moved onto the query, the value projection onto the output projection -- so a decode
step reads T x D bf16 per clip and layer once for all heads instead of K^T and V^T (half the bytes of the kernel that bounds
batched decode), and the encoder projects no cross K/V at all.  It is the same function as the reference's
graph (transformers modeling_moonshine.py:265-330, reference core/moonshine-model.cpp:380-517 for the loop) in exact
arithmetic, rounds at different points in bf16, and is held here to the SAME gates as the classic path: the kernel alone
against numpy, then the parity bodies of tests/test_gpu_parity.py / test_gpu_long_parity.py re-run with the form forced on
(logits max-abs <= 5e-2, ids where the oracle's margin > 0.1, HF goldens, the benchmarked 256 x 10 s configuration)."""
import ctypes as C
import os

import numpy as np
import pytest

import test_gpu_parity as tp
from oracle import moonshine_ref as ref
from oracle.weights import ARCHS, make_audio, make_weights

pytestmark = pytest.mark.gpu


def _bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32).reshape(x.shape)


def _run_kernel(qt, enc, Ts, starts, D, iters=0):
    from moonshine_amd.hip_api import load_dev_library as load_library   # msh_test_*: the development library

    lib = load_library()
    M = len(Ts)
    qt = np.ascontiguousarray(qt, np.float32)
    enc = np.ascontiguousarray(enc, np.float32)
    Ts = np.asarray(Ts, np.int32)
    starts = np.asarray(starts, np.int32)
    out = np.zeros((M, 8 * D), np.float32)
    ms = lib.msh_test_cross_absorbed(qt.ctypes.data, enc.ctypes.data, enc.shape[0], Ts.ctypes.data, starts.ctypes.data, M, D,
                                     out.ctypes.data, iters)
    assert ms >= 0, "kernel launch failed"
    return out.reshape(M, 8, D), ms


def _numpy_ctx(qt, enc16, Ts, starts, D):
    """softmax over the clip's frames of qt_h . enc[t] (base-2 exponent, the kernel's domain), weighted sum of the rows."""
    M = len(Ts)
    out = np.zeros((M, 8, D), np.float64)
    for b in range(M):
        E = enc16[starts[b]:starts[b] + Ts[b]].astype(np.float64)
        s = qt[b].reshape(8, D).astype(np.float64) @ E.T
        p = np.exp2(s - s.max(axis=1, keepdims=True))
        out[b] = (p / p.sum(axis=1, keepdims=True)) @ E
    return out


@pytest.mark.parametrize("D", [416, 288])
def test_kernel_vs_numpy_ragged(D):
    """Frame counts around every boundary of the kernel: one frame, one short of / exactly / one past a 16-key tile, fewer
    tiles than waves, the 10 s clip (415), a long clip (many rounds of the LDS ring), rows starting at odd offsets.  The
    random rows make any transposition or mis-assigned lane visible."""
    rng = np.random.default_rng(5)
    Ts = [1, 15, 16, 17, 33, 63, 64, 65, 415, 129, 1000, 7, 48, 415, 96, 2]
    starts, row = [], 3
    for t in Ts:
        starts.append(row)
        row += t + int(rng.integers(0, 9))
    R = row + 5
    enc = rng.standard_normal((R, D)).astype(np.float32)
    enc[:, ::7] *= 3.0
    enc16 = _bf16_round(enc)
    qt = (rng.standard_normal((len(Ts), 8 * D)) * (2.5 / np.sqrt(D))).astype(np.float32)
    got, _ = _run_kernel(qt, enc, Ts, starts, D)
    want = _numpy_ctx(qt, enc16, Ts, starts, D)
    err = np.abs(got - want)
    tol = 2.0 ** -8 * np.abs(want) + 2e-3    # the output is bf16: half an ulp relative, plus the products' own error
    worst = float((err - tol).max())
    assert worst <= 0, (worst, float(err.max()), np.unravel_index(np.argmax(err - tol), err.shape))


def test_kernel_sharp_and_flat_scores():
    """Softmax extremes: one dominating key late in the clip (the running reference must move after many tiles were
    accumulated), scores growing steadily along the clip (the reference moves again and again), and all-equal scores."""
    D = 416
    rng = np.random.default_rng(9)
    T = 415
    enc = rng.standard_normal((T, D)).astype(np.float32)
    enc16 = _bf16_round(enc)
    qt = np.zeros((3, 8 * D), np.float32)
    # clip 0: head h looks for row 400 - 10 h: qt = c * that row -> score ~ c |row|^2 >> others
    for h in range(8):
        qt[0, h * D:(h + 1) * D] = enc16[400 - 10 * h] * (40.0 / D)
    # clip 1: scores rise along the clip: qt . enc[t] = t / 4 through a planted coordinate
    enc2 = enc.copy()
    enc2[:, 0] = np.arange(T) / 64.0
    # clip 2: zeros -> uniform average
    encs = np.concatenate([enc, enc2, enc], axis=0)
    qt[1, 0::D] = 16.0
    got, _ = _run_kernel(qt, encs, [T, T, T], [0, T, 2 * T], D)
    want = _numpy_ctx(qt, _bf16_round(encs), [T, T, T], [0, T, 2 * T], D)
    err = np.abs(got - want)
    tol = 2.0 ** -8 * np.abs(want) + 2e-3
    assert float((err - tol).max()) <= 0, float(err.max())


def test_kernel_rate_at_the_benchmark_shape():
    """256 clips x 415 frames at D = 416 (BASELINE config 3): prints the kernel's rate; the bound is only a smoke alarm."""
    D, T, M = 416, 415, 256
    rng = np.random.default_rng(1)
    rows = 424
    enc = rng.standard_normal((M * rows, D)).astype(np.float32)
    qt = (rng.standard_normal((M, 8 * D)) * (2.0 / np.sqrt(D))).astype(np.float32)
    Ts, starts = [T] * M, [b * rows for b in range(M)]
    got, ms = _run_kernel(qt, enc, Ts, starts, D, iters=50)
    want = _numpy_ctx(qt[:2], _bf16_round(enc), Ts[:2], starts[:2], D)
    assert float(np.abs(got[:2] - want).max()) <= 2e-2
    mb = M * (T * D * 2 + 8 * D * 6) / 1e6
    print(f"\n[absorbed cross-attention] {M} x {T} frames: {ms * 1e3:.1f} us per launch, {mb:.1f} MB = {mb / ms / 1e3:.2f} TB/s")
    assert ms < 0.08


# ---------------------------------------------------------------- the engine with the form forced on
@pytest.mark.parametrize("D,M", [(416, 256), (416, 37), (288, 64), (416, 1)])
def test_two_stage_query_kernel_vs_numpy(D, M):
    """k_crossq.hip alone: qt_h = Wk_h^T (Wq_h' LN(x)) from the two factors, against float64 numpy on the bf16-rounded weights.
    Rows with a large common offset and an outlier feature exercise the LayerNorm; M = 37 / 1 leave a ragged last row tile.
    The kernel rounds LN(x) to bf16 (like every decode GEMM) and keeps q and its output as two bf16 halves: the bound is
    that of one bf16 operand rounding over K = D."""
    from moonshine_amd.hip_api import load_dev_library as load_library   # msh_test_*: the development library

    lib = load_library()
    rng = np.random.default_rng(D + M)
    dh = D // 8
    x = rng.standard_normal((M, D)).astype(np.float32)
    x += rng.standard_normal((M, 1)).astype(np.float32) * 3.0   # per-row offset
    x[:, 5] += 20.0                                              # an outlier feature
    wq = (rng.standard_normal((D, D)) / np.sqrt(D) * (1.4427 / np.sqrt(dh))).astype(np.float32)
    wk = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    out = np.zeros((M, 8 * D), np.float32)
    ms = lib.msh_test_crossq2(x.ctypes.data, wq.ctypes.data, wk.ctypes.data, M, D, out.ctypes.data, 0)
    assert ms >= 0, "kernel launch failed"
    xd = x.astype(np.float64)
    ln = (xd - xd.mean(1, keepdims=True)) / np.sqrt(xd.var(1, keepdims=True) + 1e-5)
    q = ln @ _bf16_round(wq).astype(np.float64).T                                  # [M][D] = (h, j)
    wk16 = _bf16_round(wk).astype(np.float64)
    want = np.einsum("mhj,hjd->mhd", q.reshape(M, 8, dh), wk16.reshape(8, dh, D))   # [M][8][D]
    got = out.reshape(M, 8, D).astype(np.float64)
    err = np.abs(got - want).max()
    scale = np.abs(want).max()
    assert err <= 6e-3 * scale, f"max abs err {err:.3e} against a range of {scale:.3e}"
    # and as tight as the merged-weight form the kernel replaces: relative RMS error
    rel = np.sqrt(((got - want) ** 2).mean() / (want ** 2).mean())
    assert rel <= 3e-3, f"relative RMS error {rel:.3e}"


def test_two_stage_query_kernel_rate():
    """Informational: per-launch time of the query kernel at the benchmark shape (256 rows), back-to-back launches."""
    from moonshine_amd.hip_api import load_dev_library as load_library   # msh_test_*: the development library

    lib = load_library()
    D, M = 416, 256
    rng = np.random.default_rng(3)
    x = rng.standard_normal((M, D)).astype(np.float32)
    wq = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    wk = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
    out = np.zeros((M, 8 * D), np.float32)
    ms = lib.msh_test_crossq2(x.ctypes.data, wq.ctypes.data, wk.ctypes.data, M, D, out.ctypes.data, 300)
    assert ms > 0
    print(f"\n[two-stage crossq] M = {M}: {ms * 1e3:.2f} us per launch back to back")


def _absorbed_engine(tmp_path_factory, arch, seed=0, weights=None):
    e, w, cfg = tp._engine(tmp_path_factory, arch, seed, weights)
    e.set_cross_mode("absorbed")
    return e, w, cfg


@pytest.fixture(scope="module")
def base_x(tmp_path_factory):
    return _absorbed_engine(tmp_path_factory, "base")


@pytest.fixture(scope="module")
def tiny_x(tmp_path_factory):
    return _absorbed_engine(tmp_path_factory, "tiny")


@pytest.mark.parametrize("case", ["base_10s", "base_vadtrunc"])
def test_base_against_hf_golden_absorbed(base_x, case, golden_dir):
    tp.test_base_against_hf_golden(base_x, case, golden_dir)
    assert base_x[0].cross_absorbed()


def test_tiny_against_hf_golden_absorbed(tiny_x, golden_dir):
    e, w, cfg = tiny_x
    g = np.load(os.path.join(golden_dir, "golden_tiny_2s.npz"))
    audio = make_audio(int(g["clip"]), int(g["n_samples"]))
    e.encode([audio])
    assert e.cross_absorbed()
    gold = g["tokens"].astype(np.int32)
    steps = len(gold) - 1
    toks, logits = e.decode(forced_steps=steps, teacher=gold[None, :], want_logits=steps)
    for i in range(steps):
        assert float(np.abs(logits[i, 0][g["logit_idx"][i]] - g["logit_val"][i]).max()) <= tp.LOGIT_MAXABS


def test_tiny_ragged_teacher_forced_vs_oracle_absorbed(tiny_x):
    """Ragged tiny batch (frame counts off every tile boundary, one very short clip), 12 teacher-forced steps against the
    oracle at the default tolerances."""
    e, w, cfg = tiny_x
    clips = [make_audio(20 + i, n) for i, n in enumerate([16000, 30000, 12345, 9000, 40007, 1300])]
    worst, flips = tp._teacher_logit_check(e, w, cfg, clips, 12)
    assert e.cross_absorbed()
    print(f"\ntiny ragged, absorbed: logits max-abs {worst:.3e}, {flips} near-tie flips")


def test_base_ragged_batch_vs_oracle_absorbed(base_x):
    e, w, cfg = base_x
    lens = [160000, 48000, 159744, 100000]
    clips = [make_audio(50 + i, n) for i, n in enumerate(lens)]
    worst, flips = tp._teacher_logit_check(e, w, cfg, [clips[1], clips[3]], 8)
    assert e.cross_absorbed()
    print(f"\nbase ragged, absorbed: logits max-abs {worst:.3e}, {flips} near-tie flips")


def test_long_clip_absorbed(tiny_x):
    """30 s and 44 s clips (T = 1249 / 1850: dozens of rounds of the LDS ring per wave) with 100 forced steps."""
    e, w, cfg = tiny_x
    clips = [make_audio(70, 480000), make_audio(71, 710400), make_audio(72, 20000)]
    worst, flips = tp._teacher_logit_check(e, w, cfg, clips, 100)
    assert e.cross_absorbed()
    print(f"\ntiny long clips, absorbed: logits max-abs {worst:.3e}, {flips} near-tie flips")


def test_base_batch256_benchmark_path_vs_oracle_absorbed(base_x):
    """The benchmarked configuration (base, 256 x 10 s, 65 forced steps) on the absorbed path, against the oracle, with the
    tolerances of the classic path's own test."""
    tp.test_base_batch256_benchmark_path_vs_oracle(base_x)
    assert base_x[0].cross_absorbed()


def test_one_form_per_engine_whatever_the_batch_size(tmp_path_factory, monkeypatch):
    """An engine holds ONE cross-attention form (msh_set_cross_mode), never one per batch: the default is the reference's
    projected form at any batch size (until round 4 mode 0 switched to the absorbed form at 192 clips, so a clip's ids
    depended on how many neighbours it had); within a form a batch's ids do not depend on which clips share it; the
    absorbed form refuses the word-timestamp capture and fp8 keys instead of silently changing form."""
    from moonshine_amd.hip_api import MshError

    # (8 clips of 1 s are 448 rows: alone they would take the split-K encoder GEMMs, among 200 the tiled ones -- another
    #  summation order; this test is about the DECODER's form, so the encoder is pinned to one set)
    monkeypatch.setenv("MSH_ENC_SMALL_ROWS", "0")
    e, w, cfg = tp._engine(tmp_path_factory, "tiny", 3)
    assert e.cross_absorbed_supported()
    clips = [make_audio(300 + i, 16000 + 37 * i) for i in range(200)]
    # The "small" batch has 136 clips: the decode kernels change their summation order at 64 clips and the LM head at 128
    # (DESIGN.md section 4), so byte-equal ids are promised between batches on the same side of those lines -- 136 and 200
    # clips run the same kernels.  (Until round 6 this compared 8 clips with 200 and held by the luck of six steps' margins.)
    n_small = 136
    e.encode(clips[:n_small])
    assert not e.cross_absorbed()
    small = e.decode(forced_steps=6)[0]
    e.encode(clips)
    assert not e.cross_absorbed()            # 200 clips: still the projected form
    big = e.decode(forced_steps=6)[0]
    assert big[:n_small] == small            # a clip among 136 == the same clip among 200
    e.set_cross_mode("absorbed")
    big_x = e.transcribe_tokens(clips, forced_steps=6)
    assert e.cross_absorbed()
    assert e.transcribe_tokens(clips[:n_small], forced_steps=6) == big_x[:n_small]
    assert e.cross_absorbed()                # 136 clips: still the absorbed form
    agree = sum(a == b for a, b in zip(small, big_x[:n_small]))
    assert agree >= n_small * 3 // 4, (agree, n_small)    # two roundings of the same function: near-ties may flip, nothing else
    tp.margins.record(clips_with_equal_ids_across_forms=agree, clips=n_small, steps=6)
    # the word-timestamp capture reads K^T, fp8 keys are projected keys: neither is available in the absorbed form
    e.lib.msh_set_capture_cross_attention(e.h, 1)
    with pytest.raises(MshError):
        e.encode(clips[:4])
    e.lib.msh_set_capture_cross_attention(e.h, 0)
    e.set_kv_dtype("fp8")
    with pytest.raises(MshError):
        e.encode(clips[:4])
    e.set_kv_dtype("bf16")
    assert e.transcribe_tokens(clips[:n_small], forced_steps=6) == big_x[:n_small]
    e.set_cross_mode("kv")
    assert e.transcribe_tokens(clips[:n_small], forced_steps=6) == small
    # an architecture without the absorbed operands says so
    m, _, _ = tp._engine(tmp_path_factory, "micro", 0)
    if not m.cross_absorbed_supported():
        with pytest.raises(MshError):
            m.set_cross_mode("absorbed")


def test_decode_graphs_are_cached_per_shape(tmp_path_factory):
    """Captured decode steps are kept per batch shape in a small LRU (engine.cpp DecodeGroup::graphs): alternating between
    shapes the engine has seen costs no further capture, a changing step budget inside one capacity bucket is the SAME
    shape, and the multi-step graph of a new shape is only built when the shape comes back."""
    e, w, cfg = tp._engine(tmp_path_factory, "tiny", 4, dev=True)   # (graph_captures: the development library's hook)
    a = [make_audio(500 + i, 32000) for i in range(6)]
    b = [make_audio(520 + i, 36000 + 640 * i) for i in range(9)]
    tb = e.transcribe_tokens(b, forced_steps=20)      # the larger shape first: the workspaces settle (a grown workspace
    c0 = e.graph_captures()                           # invalidates every captured pointer, and with it the cache)
    assert c0 in (1, 2)                               # first shape: one-step graph (+ the 8-step one unless MSH_DEC_GRAPH_STEPS=1)
    multi = c0 - 1
    ta = e.transcribe_tokens(a, forced_steps=20)
    assert e.graph_captures() == c0 + 1               # new shape: its one-step graph only
    assert e.transcribe_tokens(b, forced_steps=20) == tb
    assert e.graph_captures() == c0 + 1               # back to a cached shape: nothing captured
    ta33 = e.transcribe_tokens(a, forced_steps=33)    # another budget, same capacity bucket: the SAME shape, which has now
    assert [t[:21] for t in ta33] == ta               # come back -> it gets its 8-step graph, once
    c1 = e.graph_captures()
    assert c1 == c0 + 1 + multi
    for _ in range(3):
        assert e.transcribe_tokens(a, forced_steps=20) == ta
        assert e.transcribe_tokens(b, forced_steps=20) == tb
    assert e.graph_captures() == c1


def test_decode_gemm_tile_order_and_graph_blocking_do_not_change_ids(tmp_path_factory):
    """Two launch-side choices of this round must be invisible in the results: the XCD-aware block -> tile order of the decode
    GEMMs (MSH_DEC_XCD) and the eight-steps-per-replay decode graph (MSH_DEC_GRAPH_STEPS).  Both are read once per process, so
    the comparison runs in child processes: same weights, same clips, 19 free-running steps (two blocks of eight plus three
    single steps), all four combinations must give identical ids."""
    import json
    import subprocess
    import sys

    code = (
        "import json, sys, os\n"
        "sys.path.insert(0, %r)\n"
        "import tempfile\n"
        "from moonshine_amd.hip_api import Engine\n"
        "from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors\n"
        "cfg = ARCHS['tiny']; w = make_weights(cfg, 5)\n"
        "d = tempfile.mkdtemp(); p = os.path.join(d, 'model.safetensors')\n"
        "save_safetensors(p, w, {'arch': cfg.name, 'heads': str(cfg.heads)})\n"
        "e = Engine(0); e.load_weights_file(p)\n"
        "clips = [make_audio(900 + i, 20000 + 811 * i) for i in range(40)]\n"
        "print(json.dumps(e.transcribe_tokens(clips, forced_steps=19)))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for xcd in ("1", "0"):
        for gs in ("8", "1"):
            env = dict(os.environ, MSH_DEC_XCD=xcd, MSH_DEC_GRAPH_STEPS=gs)
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert all(o == outs[0] for o in outs[1:])
