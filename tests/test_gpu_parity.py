"""GPU parity: the HIP path (through the C ABI of include/moonshine_hip.h) against the CPU oracle
and the committed HF golden vectors.  Run on the MI355X box with `pytest -m gpu`.

Stated tolerances (bf16 GEMM operands, fp32 accumulate / softmax / norms):
  encoder last_hidden_state : rel-RMS <= 1e-2, max-abs <= 6e-2   (values are O(1) after the final LN)
  logits                    : max-abs <= 5e-2 over the compared entries
  greedy ids                : identical wherever the oracle's top-1 margin exceeds 0.1
Integer / control behaviour (EOS stop, step budget, token lists, batch independence) is bit-exact.
"""
import os

import numpy as np
import pytest

import margins
from oracle import moonshine_ref as ref
from oracle.host_ref import max_decode_len
from oracle.weights import ARCHS, make_audio, make_weights, save_safetensors

pytestmark = pytest.mark.gpu

ENC_RELRMS = 1e-2
ENC_MAXABS = 6e-2
LOGIT_MAXABS = 5e-2
MARGIN = 0.1


def _engine(tmp_path_factory, arch, seed=0, weights=None, dev=False):
    """dev = True: an engine of the development library (debug_read / graph_captures are hooks of that library only)."""
    from moonshine_amd.hip_api import Engine

    cfg = ARCHS[arch]
    w = weights if weights is not None else make_weights(cfg, seed)
    d = tmp_path_factory.mktemp(f"w_{arch}_{seed}")
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    e = Engine(0, dev=dev)
    e.load_weights_file(path)
    os.remove(path)
    return e, w, cfg


@pytest.fixture(scope="module")
def micro(tmp_path_factory):
    return _engine(tmp_path_factory, "micro", 0)


@pytest.fixture(scope="module")
def base(tmp_path_factory):
    return _engine(tmp_path_factory, "base", 0)


def _enc_check(got, want):
    err = got - want
    relrms = float(np.sqrt((err**2).mean()) / np.sqrt((want**2).mean()))
    maxabs = float(np.abs(err).max())
    prev = margins.RECORDS.get(os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" (")[0], {})
    margins.record(encoder_rel_rms=max(relrms, prev.get("encoder_rel_rms", 0.0)), encoder_max_abs=max(maxabs, prev.get("encoder_max_abs", 0.0)),
                   encoder_checks=prev.get("encoder_checks", 0) + 1)
    assert relrms <= ENC_RELRMS, (relrms, maxabs)
    assert maxabs <= ENC_MAXABS, (relrms, maxabs)
    return relrms, maxabs


def test_model_info(micro, base):
    e, _, cfg = base
    mi = e.info()
    assert (mi.hidden, mi.ffn, mi.enc_layers, mi.dec_layers, mi.heads, mi.head_dim, mi.vocab) == (416, 1664, 8, 8, 8, 52, 32768)
    assert (mi.bos, mi.eos) == (1, 2)
    assert micro[0].info().head_dim == 16


def test_micro_encoder_ragged_batch(micro):
    e, w, cfg = micro
    lens = [16000, 23789, 9000, 895, 40000]
    clips = [make_audio(10 + i, n) for i, n in enumerate(lens)]
    e.set_keep_encoder_output(True)
    e.encode(clips)
    outs = []
    for i, c in enumerate(clips):
        got = e.encoder_output(i)
        want = ref.encoder_forward(w, cfg, c)
        assert got.shape == want.shape == (ref.conv_out_lengths(lens[i])[2], cfg.hidden)
        _enc_check(got, want)
        outs.append(got)
    # batch independence: a clip encoded alone gives bit-identical frames
    for i in (1, 3):
        e.encode([clips[i]])
        np.testing.assert_array_equal(e.encoder_output(0), outs[i])


@pytest.mark.parametrize("arch", ["tiny", "base"])
def test_conv_k_order_tap_inner_vs_tap_major(tmp_path_factory, monkeypatch, arch):
    """The conv GEMMs walk K channel-block-major (gemm_common.h conv_k_offset; weights re-ordered at load).  The order the
    window lies in memory (MSH_CONV_KORDER=0, rounds 1-5) is the same sum in another order: both against the oracle, and
    against each other inside the tolerance.  tiny takes the register-staged tiled kernel, base the LDS-DMA one."""
    monkeypatch.setenv("MSH_DEV_KNOBS", "1")
    clips = [make_audio(70 + i, n) for i, n in enumerate([16000, 31000, 9000])]
    outs = {}
    for order in ("0", "1"):
        monkeypatch.setenv("MSH_CONV_KORDER", order)
        e, w, cfg = _engine(tmp_path_factory, arch, 3)
        e.set_keep_encoder_output(True)
        e.encode(clips)
        outs[order] = [e.encoder_output(i) for i in range(len(clips))]
        for i, c in enumerate(clips):
            _enc_check(outs[order][i], ref.encoder_forward(w, cfg, c))
        e.close()
    for a, b in zip(outs["0"], outs["1"]):
        d = a - b
        assert float(np.sqrt((d**2).mean()) / np.sqrt((b**2).mean())) <= 8e-3   # (bf16 roundings of x2 that fall the other way, through every layer: inside the 1e-2 each order is held to against the oracle)
        assert not np.array_equal(a, b) or arch == "micro"   # (a different order of the fp32 sum: equal bits would mean the switch did nothing)


@pytest.mark.parametrize("arch", ["micro", "tiny", "base"])
def test_groupnorm_statistics_from_conv1_row_sums(tmp_path_factory, monkeypatch, arch):
    """conv1's epilogue leaves the row sums of what it stores and the GroupNorm statistics are summed from those (engine.cpp,
    EpiTanhBf16::rowsum); MSH_GN_ROWSUMS=0 is the pass over the conv1 output they replace.  The same values summed in another
    order: every clip's {mean, rstd} agrees to fp32 rounding.  Ragged clips, three tile shapes (micro / tiny run the
    register-staged kernel on 64- / 144-column tiles, base the LDS-DMA kernel on 208-column tiles with staged stores)."""
    monkeypatch.setenv("MSH_DEV_KNOBS", "1")
    e, w, cfg = _engine(tmp_path_factory, arch, 5, dev=True)
    e.set_keep_encoder_output(True)
    clips = [make_audio(90 + i, n) for i, n in enumerate([16000, 40123, 895, 23789])]
    stats, outs = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MSH_GN_ROWSUMS", mode)
        e.encode(clips)
        stats[mode] = e.debug_read("gn_stats").view(np.float32).reshape(-1, 2).copy()
        outs[mode] = [e.encoder_output(i) for i in range(len(clips))]
    assert stats["0"].shape == (len(clips), 2) and np.all(stats["0"][:, 1] > 0)
    np.testing.assert_allclose(stats["1"], stats["0"], rtol=2e-6, atol=2e-7)
    for i, c in enumerate(clips):
        _enc_check(outs["1"][i], ref.encoder_forward(w, cfg, c))
    e.close()


def test_micro_too_short_clip_is_an_error(micro):
    from moonshine_amd.hip_api import MshError

    e, _, _ = micro
    with pytest.raises(MshError) as ei:
        e.encode([np.zeros(600, np.float32)])
    assert ei.value.code == -3


def _teacher_logit_check(e, w, cfg, clips, steps):
    """Teacher-force the oracle's own greedy ids through the GPU decoder; compare logits + argmax."""
    encs = [ref.encoder_forward(w, cfg, c) for c in clips]
    gold = []
    gold_logits = []
    for enc in encs:
        toks, lg = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, return_logits=True)
        gold.append(toks)
        gold_logits.append(lg)
    teacher = np.asarray(gold, np.int32)
    e.encode(clips)
    toks, logits = e.decode(forced_steps=steps, teacher=teacher, want_logits=steps)
    flips = 0
    worst = 0.0
    for b in range(len(clips)):
        for i in range(steps):
            g = gold_logits[b][i]
            d = float(np.abs(logits[i, b] - g).max())
            worst = max(worst, d)
            top2 = np.partition(g, -2)[-2:]
            margin = float(top2[1] - top2[0])
            if margin > MARGIN:
                assert toks[b][i + 1] == gold[b][i + 1], (b, i, margin)
            elif toks[b][i + 1] != gold[b][i + 1]:
                flips += 1
    margins.record(logits_max_abs=worst, near_tie_flips=flips, clips=len(clips), steps=steps)
    assert worst <= LOGIT_MAXABS, worst
    return worst, flips


def test_micro_decode_logits_teacher_forced(micro):
    e, w, cfg = micro
    clips = [make_audio(20 + i, n) for i, n in enumerate([16000, 30000, 12345])]
    _teacher_logit_check(e, w, cfg, clips, 12)


def test_micro_greedy_free_running(micro):
    e, w, cfg = micro
    clips = [make_audio(30 + i, 16000 + 1000 * i) for i in range(6)]
    got = e.transcribe_tokens(clips, forced_steps=7)
    for c, g in zip(clips, got):
        enc = ref.encoder_forward(w, cfg, c)
        toks, lg = ref.greedy_decode(w, cfg, enc, 7, ignore_eos=True, return_logits=True)
        # compare up to the first step whose oracle margin is within the stated tolerance
        for i in range(7):
            top2 = np.partition(lg[i], -2)[-2:]
            if top2[1] - top2[0] <= MARGIN:
                break
            assert g[i + 1] == toks[i + 1]
        assert g[0] == cfg.bos and len(g) == 8


def eos_test_weights():
    """micro weights whose greedy sequences hit EOS mid-sequence for some clips and run to the step
    budget for others, with clear (> 0.15) oracle margins on 5 of the 8 clips (found by a CPU search
    with the oracle): seed 39, tied embedding x2, vocabulary rows 412 <-> EOS swapped."""
    cfg = ARCHS["micro"]
    w = dict(make_weights(cfg, 39))
    E = w["model.decoder.embed_tokens.weight"] * np.float32(2.0)
    E[[412, cfg.eos]] = E[[cfg.eos, 412]]
    w["model.decoder.embed_tokens.weight"] = np.ascontiguousarray(E)
    lens = [16000, 48000, 20000, 32000, 9000, 64000, 24000, 40000]
    clips = [make_audio(40 + i, n) for i, n in enumerate(lens)]
    return cfg, w, lens, clips


def test_micro_eos_and_budget_semantics(tmp_path_factory):
    """EOS stop / per-clip step budget (reference core/moonshine-model.cpp:347-349, 380-517): token lists
    must equal the oracle loop exactly, including where each clip stops."""
    cfg, w, lens, clips = eos_test_weights()
    e, w, cfg = _engine(tmp_path_factory, "micro", 39, w)
    got = e.transcribe_tokens(clips)  # reference semantics: EOS + budget
    n_mid_eos = n_budget = compared = 0
    for c, g, n in zip(clips, got, lens):
        enc = ref.encoder_forward(w, cfg, c)
        budget = max_decode_len(n)
        toks, lg = ref.greedy_decode(w, cfg, enc, budget, return_logits=True)
        margins = [float(np.diff(np.partition(l, -2)[-2:])[0]) for l in lg]
        assert len(g) <= budget + 1
        if min(margins) > MARGIN:  # unambiguous clip: must match exactly, including where it stops
            assert g == toks, (n, g, toks)
            compared += 1
            n_mid_eos += int(toks[-1] == cfg.eos and len(toks) > 2)
            n_budget += int(toks[-1] != cfg.eos and len(toks) == budget + 1)
        else:
            k = next(i for i, m in enumerate(margins) if m <= MARGIN)
            assert g[: k + 1] == toks[: k + 1]
    assert compared >= 4 and n_mid_eos >= 1 and n_budget >= 1, (compared, n_mid_eos, n_budget)


@pytest.mark.parametrize("case", ["base_10s", "base_vadtrunc"])
def test_base_against_hf_golden(base, case, golden_dir):
    e, w, cfg = base
    g = np.load(os.path.join(golden_dir, f"golden_{case}.npz"))
    audio = make_audio(int(g["clip"]), int(g["n_samples"]))
    e.set_keep_encoder_output(True)
    e.encode([audio])
    enc = e.encoder_output(0)
    rows = g["enc_rows"]
    _enc_check(enc[rows], g["enc"])
    gold = g["tokens"].astype(np.int32)
    steps = len(gold) - 1
    toks, logits = e.decode(forced_steps=steps, teacher=gold[None, :], want_logits=steps)
    for i in range(steps):
        idx, val = g["logit_idx"][i], g["logit_val"][i]
        assert float(np.abs(logits[i, 0][idx] - val).max()) <= LOGIT_MAXABS
        if val[0] - val[1] > MARGIN:
            assert toks[0][i + 1] == gold[i + 1]


def test_tiny_against_hf_golden(tmp_path_factory, golden_dir):
    e, w, cfg = _engine(tmp_path_factory, "tiny", 0)
    g = np.load(os.path.join(golden_dir, "golden_tiny_2s.npz"))
    audio = make_audio(int(g["clip"]), int(g["n_samples"]))
    e.set_keep_encoder_output(True)
    e.encode([audio])
    _enc_check(e.encoder_output(0)[g["enc_rows"]], g["enc"])
    gold = g["tokens"].astype(np.int32)
    steps = len(gold) - 1
    toks, logits = e.decode(forced_steps=steps, teacher=gold[None, :], want_logits=steps)
    for i in range(steps):
        assert float(np.abs(logits[i, 0][g["logit_idx"][i]] - g["logit_val"][i]).max()) <= LOGIT_MAXABS


def test_base_ragged_batch_vs_oracle(base):
    e, w, cfg = base
    lens = [160000, 48000, 159744, 100000]
    clips = [make_audio(50 + i, n) for i, n in enumerate(lens)]
    e.set_keep_encoder_output(True)
    e.encode(clips)
    for i in (1, 3):
        _enc_check(e.encoder_output(i), ref.encoder_forward(w, cfg, clips[i]))
    _teacher_logit_check(e, w, cfg, [clips[1]], 8)


def test_base_full_size_batch_properties(base):
    """Size-independent properties at a benchmark-sized batch (32 x 10 s, 65 forced steps):
    run-to-run determinism, permutation invariance, and batch == single-clip ids."""
    e, w, cfg = base
    n = 32
    clips = [make_audio(100 + i, 160000) for i in range(n)]
    a = e.transcribe_tokens(clips, forced_steps=65)
    b = e.transcribe_tokens(clips, forced_steps=65)
    assert a == b
    assert all(len(t) == 66 and t[0] == cfg.bos for t in a)
    perm = np.random.default_rng(0).permutation(n)
    c = e.transcribe_tokens([clips[j] for j in perm], forced_steps=65)
    for k, j in enumerate(perm):
        assert c[k] == a[j]
    single = e.transcribe_tokens([clips[5]], forced_steps=65)
    assert single[0] == a[5]
    # reference semantics on the same batch: every list is a prefix-compatible, EOS-terminated or
    # budget-limited version of the forced run
    d = e.transcribe_tokens(clips)
    for t_forced, t_ref in zip(a, d):
        assert t_ref == t_forced[: len(t_ref)]
        assert len(t_ref) == 66 or t_ref[-1] == cfg.eos


@pytest.mark.parametrize("shape", ["uniform", "ragged"])
def test_encoder_layer_loop_in_two_halves_is_bit_identical(base, monkeypatch, shape):
    """A large batch's encoder layers run as two halves (cut at a clip boundary on a 128-row panel boundary) on two HIP
    streams when the engine has the GPU to itself (engine.cpp run_encoder): every kernel of the loop works row by row or
    clip by clip, so encoder outputs and ids must equal those of the single chain (MSH_ENC_SPLIT=0), for a uniform batch
    (256 x 10 s, the cut in the middle) and a ragged one (the cut wherever a clip starts on a panel boundary)."""
    e, _, cfg = base
    clips = ([make_audio(100 + i, 160000) for i in range(256)] if shape == "uniform"
             else [make_audio(900 + i, 160000 - 1280 * (i % 40)) for i in range(256)])
    e.set_cross_mode("absorbed")
    e.set_keep_encoder_output(True)
    try:
        outs = []
        for sp in ("0", "1"):
            monkeypatch.setenv("MSH_ENC_SPLIT", sp)
            ids = e.transcribe_tokens(clips, forced_steps=8)
            outs.append((ids, [e.encoder_output(c) for c in (0, 100, 127, 128, 129, 200, 255)]))
        assert outs[0][0] == outs[1][0]
        for a_, b_ in zip(outs[0][1], outs[1][1]):
            assert np.isfinite(a_).all() and np.array_equal(a_, b_)
    finally:
        e.set_keep_encoder_output(False)
        e.set_cross_mode("kv")


def test_decode_groups_match_single_stream(tmp_path_factory, monkeypatch, micro):
    """The multi-stream decode split (MSH_DEC_GROUPS) must not change a single token: ragged micro batch,
    reference EOS semantics, 3 uneven groups vs the default single group."""
    cfg, w, lens, clips = eos_test_weights()
    e1, _, _ = _engine(tmp_path_factory, "micro", 39, w)
    want = e1.transcribe_tokens(clips)
    monkeypatch.setenv("MSH_DEC_GROUPS", "3")
    e3, _, _ = _engine(tmp_path_factory, "micro", 39, w)
    got = e3.transcribe_tokens(clips)
    assert got == want
    forced = e3.transcribe_tokens(clips, forced_steps=9)
    monkeypatch.delenv("MSH_DEC_GROUPS")
    assert forced == e1.transcribe_tokens(clips, forced_steps=9)


def test_fused_argmax_lm_head_matches_logits_path(tmp_path_factory):
    """At batch >= 128 the LM head reduces each 128 x 208 tile to (max, first index) in its epilogue and the
    logits are never written.  Tokens must equal the path that materialises the logits (requested here through
    want_logits) exactly -- reference EOS / budget semantics and forced steps, ragged batch of 150 clips."""
    cfg, w, lens, clips = eos_test_weights()
    e, w, cfg = _engine(tmp_path_factory, "micro", 39, w)
    many = [clips[i % len(clips)][: len(clips[i % len(clips)]) - 64 * (i // len(clips))] for i in range(150)]
    e.encode(many)
    fused, _ = e.decode()
    e.encode(many)
    plain, lg = e.decode(want_logits=3)
    assert fused == plain
    assert lg is not None and len({len(t) for t in fused}) > 2
    # the tokens are the first-max argmax of those logits
    for b in range(0, 150, 7):
        for step in range(min(3, len(plain[b]) - 1)):
            assert plain[b][step + 1] == int(np.argmax(lg[step, b]))
    e.encode(many)
    f9, _ = e.decode(forced_steps=9)
    e.encode(many)
    p9, _ = e.decode(forced_steps=9, want_logits=1)
    assert f9 == p9


def test_batches_in_flight_match_synchronous_calls(base):
    """include/moonshine_hip.h msh_submit_transcribe_tokens / msh_wait: six ragged batches on three lanes (encoders and
    decode loops of different batches overlap on the GPU) must return exactly the ids of the synchronous call, in both
    the reference stopping rule (EOS / per-clip budget) and the forced-steps mode; bad tickets are errors."""
    from moonshine_amd.hip_api import MshError

    e, _, _ = base
    batches = []
    for b in range(6):
        n = 3 + 5 * b
        batches.append([make_audio(100 * b + i, 16000 + 2377 * ((7 * i + b) % 23)) for i in range(n)])
    want_free = [e.transcribe_tokens(c) for c in batches]
    want_forced = [e.transcribe_tokens(c, forced_steps=9) for c in batches]
    e.set_batches_in_flight(3)
    try:
        for _ in range(2):  # second round reuses warmed lanes (captured graphs, grown workspaces)
            tickets = [e.submit_transcribe_tokens(c) for c in batches] + [e.submit_transcribe_tokens(c, forced_steps=9) for c in batches]
            got = [e.wait_tokens(t) for t in tickets]
            assert got[:6] == want_free
            assert got[6:] == want_forced
        with pytest.raises(MshError):
            e.wait_tokens((12345, np.zeros((1, 1), np.int32), np.zeros(1, np.int32)))
        # the synchronous path keeps working next to the lanes
        t = e.submit_transcribe_tokens(batches[5])
        assert e.transcribe_tokens(batches[0]) == want_free[0]
        assert e.wait_tokens(t) == want_free[5]
    finally:
        e.set_batches_in_flight(0)
    with pytest.raises(MshError):
        e.submit_transcribe_tokens(batches[0])


@pytest.mark.parametrize("arch", ["tiny", "base"])
def test_decode_tile_shapes_of_the_two_regimes_give_the_same_ids(tmp_path_factory, arch):
    """A decode step enqueued beside other lanes takes throughput tile shapes (32-row workgroups in the QKV / o-proj / fc2 /
    context GEMMs, k_gemm_dec.hip dec_gemm_prefer_throughput), a lone engine the latency shapes.  The summation order per output
    element does not depend on the shape: 224 ragged clips (>= 192: the throughput shapes apply) on two lanes must give exactly
    the ids of the synchronous call, in both cross-attention forms, at both widths (K = 288 and K = 416 instantiations)."""
    e, _, cfg = _engine(tmp_path_factory, arch, 3)
    clips = [make_audio(4000 + i, 16000 + 1531 * ((5 * i) % 29)) for i in range(224)]
    for form in ("absorbed", "kv"):
        e.set_cross_mode(form)
        want = e.transcribe_tokens(clips, forced_steps=12)
        e.set_batches_in_flight(2)
        try:
            t1 = e.submit_transcribe_tokens(clips, forced_steps=12)
            t2 = e.submit_transcribe_tokens(clips[::-1], forced_steps=12)
            assert e.wait_tokens(t1) == want
            assert e.wait_tokens(t2) == want[::-1]
        finally:
            e.set_batches_in_flight(0)
    e.set_cross_mode("kv")
    assert len({tuple(t) for t in want}) > 100   # the clips do decode to different things


# stated tolerance for cross-attention probabilities (softmax outputs in [0, 1]; bf16 GEMM operands upstream, bf16 K,
# fp32 scores and softmax): max-abs 5e-3 (observed 2.0e-3 on 22-frame clips, where single probabilities reach 0.13)
ATT_MAXABS = 5e-3


@pytest.mark.parametrize("arch", ["micro", "base"])
def test_cross_attention_capture_vs_oracle(arch, micro, base):
    """msh_set_capture_cross_attention / msh_get_cross_attention: the probabilities of every layer, head and step,
    in the [layers*heads][steps][frames] layout align_words takes (reference core/moonshine-model.cpp:616-640),
    against the oracle run on the SAME token sequence; ragged batch, EOS ignored."""
    e, w, cfg = micro if arch == "micro" else base
    lens = [16000, 30011, 9000] if arch == "micro" else [40000, 23456]
    steps = 9 if arch == "micro" else 6
    clips = [make_audio(300 + i, n) for i, n in enumerate(lens)]
    e.set_capture_cross_attention(True)
    try:
        toks = e.transcribe_tokens(clips, forced_steps=steps)
        for i, c in enumerate(clips):
            att = e.cross_attention(i)
            enc = ref.encoder_forward(w, cfg, c)
            _, want = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, teacher=toks[i], return_cross_attention=True)
            assert att.shape == want.shape == (cfg.dec_layers * cfg.heads, steps, enc.shape[0])
            np.testing.assert_allclose(att.sum(-1), 1.0, atol=1e-4)
            assert float(np.abs(att - want).max()) <= ATT_MAXABS
        # the capture does not change the ids, and switching it off restores the graph-replayed path
        assert e.transcribe_tokens(clips, forced_steps=steps) == toks
    finally:
        e.set_capture_cross_attention(False)
    assert e.transcribe_tokens(clips, forced_steps=steps) == toks
    from moonshine_amd.hip_api import MshError

    e2_toks = e.transcribe_tokens(clips[:1], forced_steps=2)
    assert len(e2_toks[0]) == 3
    e.set_capture_cross_attention(True)
    e.encode(clips[:1])
    with pytest.raises(MshError):  # nothing decoded since the capture was switched on
        e.cross_attention(0)
    e.set_capture_cross_attention(False)


def test_base_batch256_benchmark_path_vs_oracle(base):
    """The configuration bench.py quotes (BASELINE config 3: base, 256 x 10 s, 65 forced steps) on the path it
    runs -- hipGraph replay, 64-column / 32-row decode GEMM tiles (M >= 96 / 128), tiled LM head with the argmax
    fused into its epilogue -- against the ORACLE, not against the engine itself:
      * ids: for clips spread over the batch (first / last rows, both sides of the 16-row MFMA tile and 128-row LM-head
        tile boundaries) the oracle is teacher-forced with the GPU's own ids (no cascade); wherever its top-1 margin
        exceeds 0.1 its first-max argmax must be the GPU's next id (reference loop core/moonshine-model.cpp:380-517);
      * logits: a second pass at the SAME batch with the logits materialised (eager, same GEMM instantiations, LM head
        without the argmax epilogue) must give max-abs <= 5e-2 on those clips and EXACTLY the ids of the fused path
        for all 256 clips."""
    e, w, cfg = base
    n, steps, logit_steps = 256, 65, 12
    clips = [make_audio(1234 + i, 160000) for i in range(n)]
    toks = e.transcribe_tokens(clips, forced_steps=steps)          # graph + fused argmax
    assert all(len(t) == steps + 1 and t[0] == cfg.bos for t in toks)
    teacher = np.asarray(toks, np.int32)
    e.encode(clips)
    toks_eager, logits = e.decode(forced_steps=steps, teacher=teacher, want_logits=logit_steps)
    assert toks_eager == toks                                      # fused (max, first index) epilogue == argmax of logits
    picked = [0, 15, 16, 127, 128, 255]
    checked = flips = 0
    worst = 0.0
    for b in picked:
        enc = ref.encoder_forward(w, cfg, clips[b])
        o_toks, o_logits = ref.greedy_decode(w, cfg, enc, steps, ignore_eos=True, return_logits=True, teacher=toks[b])
        for i in range(steps):
            top2 = np.partition(o_logits[i], -2)[-2:]
            if float(top2[1] - top2[0]) > MARGIN:
                assert toks[b][i + 1] == o_toks[i + 1], (b, i, float(top2[1] - top2[0]))
                checked += 1
            elif toks[b][i + 1] != o_toks[i + 1]:
                flips += 1
            if i < logit_steps:
                worst = max(worst, float(np.abs(logits[i, b] - o_logits[i]).max()))
    margins.record(logits_max_abs=worst, ids_checked=checked, near_tie_flips=flips, clips_checked=len(picked), cross_absorbed=bool(e.cross_absorbed()))
    assert worst <= LOGIT_MAXABS, worst
    assert checked >= len(picked) * steps // 2, (checked, flips)   # the check must not be vacuous
    print(f"batch-256 parity: {checked} ids checked against the oracle, {flips} near-tie flips, logits max-abs {worst:.3e}")


@pytest.mark.parametrize("form", ["absorbed", "kv"])
def test_batches_in_flight_soak_base_256(base, form):
    """Soak test of the overlapped mode bench.py quotes (its cross-attention form, absorbed, and the projected one): 56 base
    batches of 256 x 10 s on 4 lanes (encoder GEMMs of one batch next to the decode kernels of three others), every one
    bit-equal to the ids of the serial pass.  The device code is built without packed-FP32 instructions (build.py checks
    the disassembly): with them a few clips per batch used to differ."""
    e, _, cfg = base
    steps = 65
    e.set_cross_mode(form)
    batches = [[make_audio(5000 + 300 * b + i, 160000) for i in range(256)] for b in range(4)]
    want = [e.transcribe_tokens(c, forced_steps=steps) for c in batches]
    assert want[0] != want[1]
    e.set_batches_in_flight(4)
    try:
        bad = []
        for rnd in range(2):
            tickets = [(k % 4, e.submit_transcribe_tokens(batches[k % 4], forced_steps=steps)) for k in range(28)]
            for k, (b, t) in enumerate(tickets):
                got = e.wait_tokens(t)
                if got != want[b]:
                    bad.append((rnd, k, sum(g != w_ for g, w_ in zip(got, want[b]))))
        assert not bad, f"(round, submission, differing clips): {bad}"
    finally:
        e.set_batches_in_flight(0)
        e.set_cross_mode("kv")
