"""Engine option kv_dtype = fp8 (msh_set_kv_dtype, include/moonshine_hip.h): the decoder's cross-attention K / V stored as
e4m3 bytes with per-row scales fixed at load -- half the bytes of the HBM-bound kernel that dominates a decode step.  It is a
deviation from the bf16 storage the parity tolerances were stated for, so it ships OFF by default and is held to exactly
the same gates as the default path, against the same oracle and golden vectors: logits max-abs <= 5e-2,
greedy ids identical wherever the oracle's margin exceeds 0.1 (teacher-forced), on the HF goldens, on a ragged batch and
on the benchmarked configuration (base, 256 x 10 s, 65 steps).  Switching back restores the bf16 ids bit for bit."""
import os

import numpy as np
import pytest

import margins
import test_gpu_parity as tp
from oracle import moonshine_ref as ref
from oracle.weights import ARCHS, make_audio, make_weights

pytestmark = pytest.mark.gpu


def _fp8_engine(tmp_path_factory, arch, seed=0):
    e, w, cfg = tp._engine(tmp_path_factory, arch, seed)
    e.set_kv_dtype("fp8")
    return e, w, cfg


@pytest.fixture(scope="module")
def micro8(tmp_path_factory):
    return _fp8_engine(tmp_path_factory, "micro")


@pytest.fixture(scope="module")
def base8(tmp_path_factory):
    return _fp8_engine(tmp_path_factory, "base")


def test_micro_ragged_teacher_forced_logits_fp8(micro8):
    """Ragged micro batch with frame counts that are not multiples of 8 / 16 (row padding of the byte layout), 12
    teacher-forced steps: logits and ids against the oracle at the default path's tolerances."""
    e, w, cfg = micro8
    clips = [make_audio(20 + i, n) for i, n in enumerate([16000, 30000, 12345, 9000, 40007])]
    tp._teacher_logit_check(e, w, cfg, clips, 12)


def test_very_short_clips_fp8_error_is_bounded_and_reported(micro8):
    """Where fp8 K / V is weakest: a clip of ONE encoder frame (895 samples) or a handful -- the softmax is over so few keys
    that the 3-bit mantissa of the V rows reaches the residual stream un-averaged (the bf16 path shows the same effect at
    its own scale: 0.0503 on a one-frame clip, tests/test_gpu_long_parity.py).  ids are held to the usual rule; the logits
    to 0.15, and the measured value is printed -- anything that needs the stated 5e-2 on sub-0.2-second clips keeps bf16."""
    e, w, cfg = micro8
    clips = [make_audio(60 + i, n) for i, n in enumerate([895, 1300, 2500])]
    encs = [ref.encoder_forward(w, cfg, c) for c in clips]
    gold = [ref.greedy_decode(w, cfg, enc, 8, ignore_eos=True, return_logits=True) for enc in encs]
    e.encode(clips)
    toks, logits = e.decode(forced_steps=8, teacher=np.asarray([g[0] for g in gold], np.int32), want_logits=8)
    worst = 0.0
    for b in range(len(clips)):
        for i in range(8):
            g = gold[b][1][i]
            worst = max(worst, float(np.abs(logits[i, b] - g).max()))
            top2 = np.partition(g, -2)[-2:]
            if float(top2[1] - top2[0]) > 0.2:
                assert toks[b][i + 1] == gold[b][0][i + 1], (b, i)
    margins.record(logits_max_abs=worst, tolerance=0.15)
    print(f"fp8 K/V on clips of {[x.shape[0] for x in encs]} frames: logits max-abs {worst:.4f}")
    assert worst <= 0.15, worst


@pytest.mark.parametrize("case", ["base_10s", "base_vadtrunc"])
def test_base_against_hf_golden_fp8(base8, case, golden_dir):
    tp.test_base_against_hf_golden(base8, case, golden_dir)


def test_tiny_against_hf_golden_fp8(tmp_path_factory, golden_dir):
    e, w, cfg = _fp8_engine(tmp_path_factory, "tiny")
    g = np.load(os.path.join(golden_dir, "golden_tiny_2s.npz"))
    audio = make_audio(int(g["clip"]), int(g["n_samples"]))
    e.encode([audio])
    gold = g["tokens"].astype(np.int32)
    steps = len(gold) - 1
    toks, logits = e.decode(forced_steps=steps, teacher=gold[None, :], want_logits=steps)
    for i in range(steps):
        assert float(np.abs(logits[i, 0][g["logit_idx"][i]] - g["logit_val"][i]).max()) <= tp.LOGIT_MAXABS


def test_base_batch256_benchmark_path_vs_oracle_fp8(base8):
    """The gate bench.py's `fp8_kv` figure stands on: the benchmarked configuration, unchanged test body."""
    tp.test_base_batch256_benchmark_path_vs_oracle(base8)


def test_base_long_clip_fp8(base8):
    """30 s clip (3 chunks of 512 keys) and a ragged neighbour, 80 steps, teacher-forced against the oracle."""
    e, w, cfg = base8
    clips = [make_audio(810, 480_000), make_audio(811, 52_000)]
    tp._teacher_logit_check(e, w, cfg, clips, 16)


def test_switching_kv_dtype_back_restores_bf16_ids_and_fp8_rejects_capture(tmp_path_factory):
    from moonshine_amd.hip_api import MshError

    e, w, cfg = tp._engine(tmp_path_factory, "micro", 0)
    clips = [make_audio(30 + i, 16000 + 1000 * i) for i in range(6)]
    want = e.transcribe_tokens(clips, forced_steps=9)
    e.set_kv_dtype("fp8")
    got8 = e.transcribe_tokens(clips, forced_steps=9)
    assert all(len(t) == 10 for t in got8)
    with pytest.raises(MshError):
        e.set_capture_cross_attention(True)          # the capture kernel reads bf16 keys
    e.set_kv_dtype("bf16")
    assert e.transcribe_tokens(clips, forced_steps=9) == want
    e.set_capture_cross_attention(True)
    with pytest.raises(MshError):
        e.set_kv_dtype("fp8")
    e.set_capture_cross_attention(False)
    same = sum(a == b for a, b in zip(want, got8))
    print(f"fp8 K/V: {same} of {len(clips)} micro clips with ids equal to the bf16 run over 9 free-running steps")
