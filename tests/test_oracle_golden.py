"""Pin the numpy oracle to the HuggingFace float implementation (the float oracle
the reference names, docs/models/accuracy.md:14-19) via the committed vectors
that tests/golden/make_golden.py produced from it.  CPU only."""
import os

import numpy as np
import pytest

from oracle import moonshine_ref as ref
from oracle.weights import ARCHS, make_audio, make_weights

CASES = ["micro_1s", "micro_ragged", "tiny_2s", "base_10s", "base_vadtrunc"]
_wcache = {}


def _weights(arch, seed):
    key = (arch, seed)
    if key not in _wcache:
        _wcache.clear()
        _wcache[key] = make_weights(ARCHS[arch], seed)
    return _wcache[key]


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_hf_golden(case, golden_dir):
    g = np.load(os.path.join(golden_dir, f"golden_{case}.npz"))
    arch, seed, clip, n = str(g["arch"]), int(g["seed"]), int(g["clip"]), int(g["n_samples"])
    cfg = ARCHS[arch]
    w = _weights(arch, seed)
    audio = make_audio(clip, n)
    taps = {}
    enc = ref.encoder_forward(w, cfg, audio, taps)
    rows = g["enc_rows"]
    assert enc.shape[0] == ref.conv_out_lengths(n)[2]
    # fp32 vs fp32, different summation orders: 2e-4 abs on O(1) values
    np.testing.assert_allclose(taps["conv3_gelu"][rows], g["conv3"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(enc[rows], g["enc"], atol=5e-4, rtol=1e-4)
    assert abs(float(np.abs(enc).mean()) - float(g["enc_absmean"])) < 1e-4
    gold_tokens = g["tokens"].tolist()
    steps = len(gold_tokens) - 1
    # teacher-forced on the golden ids so one near-tie cannot cascade
    toks, logits = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, return_logits=True, teacher=gold_tokens)
    for i in range(steps):
        idx, val = g["logit_idx"][i], g["logit_val"][i]
        np.testing.assert_allclose(logits[i][idx], val, atol=2e-3, rtol=1e-4)
        margin = val[0] - val[1]
        if margin > 5e-3:
            assert toks[i + 1] == gold_tokens[i + 1], (i, margin)


def test_greedy_free_running_matches_golden_micro(golden_dir):
    g = np.load(os.path.join(golden_dir, "golden_micro_1s.npz"))
    cfg = ARCHS["micro"]
    w = _weights("micro", int(g["seed"]))
    audio = make_audio(int(g["clip"]), int(g["n_samples"]))
    enc = ref.encoder_forward(w, cfg, audio)
    toks = ref.greedy_decode(w, cfg, enc, len(g["tokens"]) - 1, ignore_eos=True)
    assert toks == g["tokens"].tolist()


def test_max_len_and_lengths():
    from oracle.host_ref import max_decode_len

    assert ref.conv_out_lengths(160000) == (2499, 831, 415)
    assert ref.conv_out_lengths(159744)[2] == 414
    assert max_decode_len(160000) == 65                        # ceil(10 * 6.5)
    assert max_decode_len(159744) == 65
    assert max_decode_len(16000) == 7


def test_hf_teacher_forced_helper_matches_the_oracle_and_hf_step_by_step():
    """`oracle/hf_baseline.teacher_forced_logits` (one causal decoder call for all positions: what the full-batch GPU parity
    test compares 256 x 65 x 32768 logits with) against (a) HF's own step-by-step KV-cache loop on the same ids and (b) the
    numpy oracle, teacher-forced: same argmax everywhere, logits within fp32 noise."""
    pytest.importorskip("transformers")
    from oracle import hf_baseline as hb

    cfg = ARCHS["micro"]
    w = make_weights(cfg, 0)
    model = hb.build_hf(cfg, w)
    clips = np.stack([make_audio(40 + i, 16000) for i in range(3)])
    steps = 9
    toks = np.asarray(hb.greedy(model, cfg, clips, steps), np.int32)
    lg = hb.teacher_forced_logits(model, clips, toks, sub_batch=2)
    assert lg.shape == (3, steps, cfg.vocab)
    assert (lg.argmax(-1) == toks[:, 1:]).all()
    for b in range(3):
        enc = ref.encoder_forward(w, cfg, clips[b])
        o_toks, o_lg = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, return_logits=True, teacher=toks[b].tolist())
        assert float(np.abs(np.stack(o_lg) - lg[b]).max()) < 2e-4
