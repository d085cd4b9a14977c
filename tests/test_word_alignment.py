"""Word alignment (reference core/word-alignment.cpp): the oracle's restatement is pinned on two
independent checks -- scipy's mirror-mode median filter and a brute-force search over every monotone path -- and the C++
host code (libmoonshine.so, msh_host_dtw / msh_host_median_filter / msh_host_align_words) must agree with it exactly.
No GPU is needed: the attention matrices here are synthetic."""
import ctypes as C
import itertools

import numpy as np
import pytest

from moonshine_amd.hip_api import load_library
from moonshine_amd.synth import encode_tokenizer_bin, synthetic_vocab
from oracle import host_ref
from oracle import word_align_ref as wa


@pytest.fixture(scope="module")
def lib():
    l = load_library()
    l.msh_host_dtw.restype = C.c_int64
    l.msh_host_dtw.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64]
    l.msh_host_median_filter.restype = C.c_int32
    l.msh_host_median_filter.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32]
    l.msh_host_align_words.restype = C.c_int64
    l.msh_host_align_words.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64,
                                       C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    return l


def _cxx_dtw(lib, cost):
    cost = np.ascontiguousarray(cost, np.float32)
    n, m = cost.shape
    a = np.zeros(n + m, np.int32)
    b = np.zeros(n + m, np.int32)
    k = lib.msh_host_dtw(cost.ctypes.data, n, m, a.ctypes.data, b.ctypes.data, n + m)
    assert 0 < k <= n + m
    return a[:k].tolist(), b[:k].tolist()


def _all_paths(n, m):
    """Every monotone path (0,0) -> (n-1,m-1) with steps (1,1), (1,0), (0,1)."""
    def rec(i, j):
        if (i, j) == (n - 1, m - 1):
            yield [(i, j)]
            return
        for di, dj in ((1, 1), (1, 0), (0, 1)):
            if i + di < n and j + dj < m:
                for rest in rec(i + di, j + dj):
                    yield [(i, j)] + rest
    return rec(0, 0)


def test_dtw_finds_the_cheapest_monotone_path():
    rng = np.random.default_rng(1)
    for n, m in [(1, 1), (1, 5), (4, 1), (3, 4), (4, 6), (5, 5)]:
        cost = rng.standard_normal((n, m)).astype(np.float32)
        ti, tj = wa.dtw(cost)
        assert (ti[0], tj[0]) == (0, 0) and (ti[-1], tj[-1]) == (n - 1, m - 1)
        steps = {(ti[k + 1] - ti[k], tj[k + 1] - tj[k]) for k in range(len(ti) - 1)}
        assert steps <= {(1, 1), (1, 0), (0, 1)}
        got = sum(float(cost[i, j]) for i, j in zip(ti, tj))
        best = min(sum(float(cost[i, j]) for i, j in p) for p in _all_paths(n, m))
        assert abs(got - best) < 1e-4


def test_dtw_tie_break_prefers_the_diagonal():
    # all-zero costs: every path costs 0; the rule "diagonal, then text-only, then time-only" gives this one
    assert wa.dtw(np.zeros((3, 5), np.float32)) == ([0, 0, 0, 1, 2], [0, 1, 2, 3, 4])
    assert wa.dtw(np.zeros((4, 2), np.float32)) == ([0, 1, 2, 3], [0, 0, 0, 1])


def test_median_filter_is_scipys_mirror_mode():
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(2)
    for shape, width in [((2, 3, 40), 7), ((1, 4, 9), 5), ((3, 2, 7), 7), ((2, 2, 33), 6)]:
        x = rng.standard_normal(shape).astype(np.float32)
        w = width | 1
        np.testing.assert_array_equal(wa.median_filter(x, width), ndi.median_filter(x, size=(1, 1, w), mode="mirror"))
    # rows shorter than the padding: reflections are clamped to the row (the reference's guard), nothing reads outside
    x = rng.standard_normal((1, 2, 3)).astype(np.float32)
    y = wa.median_filter(x, 7)
    assert y.shape == x.shape and np.isfinite(y).all()
    np.testing.assert_array_equal(wa.median_filter(x, 1), x)


def test_cxx_dtw_and_median_match_the_oracle(lib):
    rng = np.random.default_rng(3)
    for n, m in [(1, 1), (1, 9), (7, 1), (6, 11), (23, 57), (66, 415)]:
        cost = rng.standard_normal((n, m)).astype(np.float32)
        if n == 6:
            cost = np.round(cost)  # plenty of ties
        assert _cxx_dtw(lib, cost) == wa.dtw(cost)
    for shape, width in [((3, 5, 40), 7), ((2, 2, 3), 7), ((1, 1, 1), 7), ((4, 3, 415), 7), ((2, 2, 10), 4), ((2, 2, 10), 1)]:
        x = rng.standard_normal(shape).astype(np.float32)
        y = x.copy()
        assert lib.msh_host_median_filter(y.ctypes.data, shape[0] * shape[1], shape[2], width) == 0
        np.testing.assert_array_equal(y, wa.median_filter(x, width))


def _synthetic_attention(rng, heads, steps, frames, noise):
    """Attention that walks the frames monotonically (what a trained model produces) plus noise."""
    centers = np.sort(rng.uniform(0, frames - 1, steps))
    f = np.arange(frames)[None, None, :]
    att = np.exp(-0.5 * ((f - centers[None, :, None]) / 3.0) ** 2) + noise * rng.random((heads, steps, frames))
    return (att / att.sum(-1, keepdims=True)).astype(np.float32), centers


def _cxx_align(lib, blob, att, tokens, tpf):
    att = np.ascontiguousarray(att, np.float32)
    toks = np.asarray(tokens, np.int32)
    text = C.create_string_buffer(1 << 16)
    times = np.zeros((len(tokens) + 1, 3), np.float32)
    n = lib.msh_host_align_words(blob, len(blob), att.ctypes.data, att.shape[0], att.shape[1], att.shape[2], toks.ctypes.data,
                                 len(tokens), tpf, text, 1 << 16, times.ctypes.data, len(tokens) + 1)
    assert n >= 0
    texts = text.value.split(b"\n") if n else []
    return [{"text": texts[i], "start": times[i, 0], "end": times[i, 1], "confidence": times[i, 2]} for i in range(n)]


def test_cxx_align_words_matches_the_oracle_and_has_the_reference_properties(lib):
    vocab = synthetic_vocab(2048)
    blob = encode_tokenizer_bin(vocab)
    rng = np.random.default_rng(4)
    word_starts = [i for i in range(259, 2048) if vocab[i][:3] == wa.WORD_MARK and len(vocab[i]) > 3]
    inner = [i for i in range(260, 2048) if vocab[i][:3] != wa.WORD_MARK and not vocab[i].startswith(b"<")]
    for case in range(12):
        steps = int(rng.integers(2, 40))
        frames = int(rng.integers(8, 200))
        body = []
        while len(body) < steps - 1:  # steps rows = tokens after BOS (the last one is dropped as "EOS")
            body.append(int(rng.choice(word_starts)))
            for _ in range(int(rng.integers(0, 3))):
                body.append(int(rng.choice(inner)))
        body = body[: steps - 1]
        if case == 3:
            body[0] = 5            # a byte-fallback special at the front: skipped by the detokeniser
        if case == 4 and len(body) > 2:
            body[1] = 259          # the bare marker: an empty word, dropped
        tokens = [1] + body + [2]
        att, _ = _synthetic_attention(rng, 6, steps, frames, noise=0.3 if case % 2 else 0.0)
        tpf = float(np.float32(10.0 / frames))
        want = wa.align_words(att, tokens, tpf, vocab, host_ref.tokens_to_text)
        got = _cxx_align(lib, blob, att, tokens, tpf)
        assert [w["text"] for w in got] == [w["text"] for w in want]
        np.testing.assert_array_equal([w["start"] for w in got], [w["start"] for w in want])
        np.testing.assert_array_equal([w["end"] for w in got], [w["end"] for w in want])
        # the properties the reference's own test states (core/word-alignment-test.cpp:52-66); "end > start" is only
        # ">=" here: two synthetic tokens may attend to the same frame, and the midpoint snapping then leaves a
        # zero-width word (the reference would produce the same)
        prev = -1.0
        for w in got:
            assert w["end"] >= w["start"] >= prev and 0.0 <= w["confidence"] <= 1.0
            prev = w["start"]
        assert got and got[-1]["end"] <= 10.0 + 1e-3


def test_align_words_recovers_a_known_alignment():
    """Clean diagonal attention: every word must land on the frames its tokens attended to (within the filter width)."""
    vocab = synthetic_vocab(1024)
    rng = np.random.default_rng(5)
    starts = [i for i in range(260, 1024) if vocab[i][:3] == wa.WORD_MARK and len(vocab[i]) > 3]
    body = [int(rng.choice(starts)) for _ in range(9)]  # nine one-token words
    tokens = [1] + body + [2]
    steps, frames = len(tokens) - 1, 120
    att, centers = _synthetic_attention(rng, 8, steps, frames, noise=0.0)
    words = wa.align_words(att, tokens, 0.1, vocab, host_ref.tokens_to_text)
    assert len(words) == 9
    for w, c in zip(words, centers[:9]):
        assert w["start"] - 0.8 <= c * 0.1 <= w["end"] + 0.8


def test_degenerate_inputs(lib):
    vocab = synthetic_vocab(512)
    blob = encode_tokenizer_bin(vocab)
    att = np.full((2, 1, 5), 0.2, np.float32)  # constant rows: zero variance -> the 1e-10 guard
    assert wa.align_words(att, [1, 2], 0.1, vocab, host_ref.tokens_to_text) == []
    assert _cxx_align(lib, blob, att, [1, 2], 0.1) == []
    att3 = np.full((2, 2, 5), 0.2, np.float32)
    tok = [1, 300, 2]
    assert [w["text"] for w in _cxx_align(lib, blob, att3, tok, 0.1)] == [w["text"] for w in wa.align_words(att3, tok, 0.1, vocab, host_ref.tokens_to_text)]
