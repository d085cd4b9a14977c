"""Pins oracle/streaming_ref.py against the golden vectors produced by the reference's own graph
wrapper modules (tests/golden/make_golden_streaming.py), and checks the driver logic it restates
(window schedule, speculative verify) for self-consistency.  CPU only."""
import os

import numpy as np
import pytest

from oracle import streaming_ref as sr
from oracle.weights import STREAMING_ARCHS, make_audio, make_streaming_weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, f"golden_stream_{name}.npz"))


def drive(g, cfg, w):
    audio = make_audio(int(g["audio_index"]), int(g["n_samples"]))
    st = sr.StreamState(cfg)
    n_chunks = audio.shape[0] // 1280
    uc = int(g["update_chunks"])
    c = update = 0
    mem_lens, windows = [], {}
    while c < n_chunks:
        for _ in range(min(uc, n_chunks - c)):
            sr.process_audio_chunk(w, cfg, st, audio[c * 1280:(c + 1) * 1280])
            c += 1
        emitted_before = st.encoder_frames_emitted
        new = sr.encode(w, cfg, st, c >= n_chunks)
        mem_lens.append(st.memory_len)
        if update in (1, 3) and new > 0:
            start = max(0, emitted_before - 16 * cfg.depth)
            windows[update] = (start, sr.encoder(w, cfg, st.features[start:]))
        update += 1
    return st, mem_lens, windows


@pytest.mark.parametrize("name", ["micro_2s", "tiny_3s", "medium_2s"])
def test_oracle_matches_reference_graphs(name):
    g = load(name)
    cfg = STREAMING_ARCHS[str(g["arch"])]
    w = make_streaming_weights(cfg, int(g["seed"]))
    st, mem_lens, windows = drive(g, cfg, w)
    assert mem_lens == g["mem_lens"].tolist()
    np.testing.assert_allclose(st.features, g["features"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(st.memory, g["memory"], atol=5e-4, rtol=0)
    for u, (start, enc) in windows.items():
        assert start == int(g[f"window{u}_start"])
        np.testing.assert_allclose(enc, g[f"window{u}_encoded"], atol=5e-4, rtol=0)
    # wide (teacher-forced) call over the golden greedy tokens
    toks = g["greedy_tokens"].tolist()
    st.decoder_reset()
    logits = sr.decode_tokens(w, cfg, st, toks[:-1])
    np.testing.assert_allclose(logits[:, :64], g["wide_logits_sel"], atol=2e-3, rtol=0)
    ti = g["wide_top8_idx"]
    np.testing.assert_allclose(np.take_along_axis(logits, ti.astype(np.int64), axis=-1), g["wide_top8_val"],
                               atol=2e-3, rtol=0)
    assert [int(np.argmax(r)) for r in logits] == toks[1:]
    if "wide_logits" in g:
        np.testing.assert_allclose(logits, g["wide_logits"], atol=2e-3, rtol=0)
    np.testing.assert_allclose(st.k_cross[0, 0], g["k_cross_l0_h0"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(st.v_cross[-1, 1], g["v_cross_last_h1"], atol=5e-4, rtol=0)
    # token-by-token equals the wide call (cache growth path)
    st.decoder_reset()
    step = np.stack([sr.decode_tokens(w, cfg, st, [t])[0] for t in toks[:-1]])
    np.testing.assert_allclose(step, logits, atol=1e-4, rtol=0)


def test_streaming_equals_one_shot_encoder():
    """The window schedule re-encodes 16*depth frames of left context: the memory it accumulates equals
    encoding the whole feature sequence at once (export.py:56-58 makes the same claim for the frontend)."""
    cfg = STREAMING_ARCHS["micro_streaming"]
    w = make_streaming_weights(cfg, 3)
    audio = make_audio(9, 1280 * 40)
    st = sr.StreamState(cfg)
    for c in range(40):
        sr.process_audio_chunk(w, cfg, st, audio[c * 1280:(c + 1) * 1280])
        if c % 3 == 2:
            sr.encode(w, cfg, st, False)
    sr.encode(w, cfg, st, True)
    one = sr.FrontendState(cfg)
    feats = sr.frontend(w, cfg, one, audio)
    np.testing.assert_allclose(st.features, feats, atol=1e-5, rtol=0)
    mem = sr.adapter(w, cfg, sr.encoder(w, cfg, feats), 0)
    np.testing.assert_allclose(st.memory, mem, atol=2e-4, rtol=0)


def test_frontend_remainder_and_state():
    cfg = STREAMING_ARCHS["micro_streaming"]
    w = make_streaming_weights(cfg, 3)
    st = sr.FrontendState(cfg)
    f = sr.frontend(w, cfg, st, make_audio(1, 1280 + 37))
    assert f.shape == (4, cfg.enc_dim) and st.sample_len == 37 and st.frame_count == 16
    f = sr.frontend(w, cfg, st, make_audio(2, 1280 - 37))        # remainder completes a frame
    assert f.shape == (4, cfg.enc_dim) and st.sample_len == 0 and st.frame_count == 32


def test_decode_full_speculative_semantics():
    cfg = STREAMING_ARCHS["micro_streaming"]
    w = make_streaming_weights(cfg, 21)
    audio = make_audio(5, 1280 * 30)
    st = sr.StreamState(cfg)
    for c in range(30):
        sr.process_audio_chunk(w, cfg, st, audio[c * 1280:(c + 1) * 1280])
    sr.encode(w, cfg, st, True)
    assert st.memory_len == 120
    cap = sr.max_tokens_for_memory(cfg, st.memory_len)
    assert cap == 16                                              # ceil(2.4 * 6.5) = 15.6 -> 16
    st.decoder_reset()
    plain = sr.decode_full(w, cfg, st)
    assert 0 < len(plain) <= cap and cfg.eos not in plain
    # a correct draft is accepted whole and yields the same tokens
    stats = {}
    st.decoder_reset()
    assert sr.decode_full(w, cfg, st, plain[:7], stats) == plain and stats["accepted"] == 7
    # a draft corrupted at position 4 is cut there and the result is unchanged
    bad = list(plain[:9])
    bad[4] = (bad[4] + 1) % cfg.vocab
    if bad[4] in (cfg.bos, cfg.eos):
        bad[4] = 5
    st.decoder_reset()
    assert sr.decode_full(w, cfg, st, bad, stats) == plain and stats["accepted"] == 4
    # a draft longer than the cap is still accepted as far as it agrees (ref:1328-1330 pushes unconditionally)
    st.decoder_reset()
    out = sr.decode_full(w, cfg, st, plain, stats)
    assert out == plain and stats["accepted"] == len(plain)


def test_segment_streamer_flow():
    cfg = STREAMING_ARCHS["micro_streaming"]
    w = make_streaming_weights(cfg, 21)
    audio = make_audio(6, 16000 * 2 + 500)
    a = sr.SegmentStreamer(w, cfg, use_speculative_decoding=True)
    b = sr.SegmentStreamer(w, cfg, use_speculative_decoding=False)
    last_a = last_b = None
    for upto, final in ((8000, False), (16000, False), (24000, False), (audio.shape[0], True)):
        last_a = a.update(audio[:upto], final)
        last_b = b.update(audio[:upto], final)
    # 32500 samples -> 25 whole chunks -> 100 features, all emitted on the final update
    assert a.st.memory_len == 100 and a.samples_processed == 25 * 1280
    assert last_a[0] == cfg.bos and last_b[0] == cfg.bos
    np.testing.assert_allclose(a.st.memory, b.st.memory, atol=0, rtol=0)
    # both are greedy from BOS over the same memory: they agree up to the shorter budget
    ca = [t for t in last_a[1:] if t != cfg.eos]
    cb = [t for t in last_b[1:] if t != cfg.eos]
    n = min(len(ca), len(cb))
    assert n > 0 and ca[:n] == cb[:n]
