"""GPU tests of the streaming architectures through the public C API (include/moonshine-c-api.h):
moonshine_load_transcriber_from_files(arch = *_STREAMING) -> streams / one-shot calls ->
Transcriber::transcribe_segment_with_streaming_model's flow (reference core/transcriber.cpp:1311-1487)
-> transcript_t.  The C++ host glue is checked against a Python restatement of that flow driving the same
device engine through msh_stream_* (the numerics of the engine are covered by test_gpu_streaming.py)."""
import math
import os

import numpy as np
import pytest

from moonshine_amd import api
from moonshine_amd.synth import STREAMING_ARCHS, make_audio, synthetic_vocab, write_streaming_model_dir
from oracle import host_ref

pytestmark = pytest.mark.gpu

CFG = STREAMING_ARCHS["micro_streaming"]


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("micro_streaming_model"))
    write_streaming_model_dir(d, CFG, seed=21)
    return d


@pytest.fixture(scope="module")
def engine(model_dir):
    from moonshine_amd.hip_api import StreamEngine

    e = StreamEngine(os.path.join(model_dir, "model.safetensors"), CFG.streaming_config_json(), max_slots=8,
                     max_memory_frames=1024)
    yield e
    e.close()


class GlueMirror:
    """transcriber.cpp:1311-1487 for one line, over the device engine."""

    def __init__(self, eng, speculative=True, mtps=6.5, decode_incomplete=True):
        self.eng, self.spec, self.mtps, self.inc = eng, speculative, mtps, decode_incomplete
        self.slot = eng.open()
        self.processed = 0
        self.last = []
        self.first = True
        self.vocab = synthetic_vocab(CFG.vocab)
        self.accepted = []

    def update(self, audio, is_final) -> bytes:
        eng, s = self.eng, self.slot
        is_new, self.first = self.first, False
        if len(audio) == 0:
            return b""
        if self.processed < len(audio):
            cc = (len(audio) - self.processed) // 1280
            if cc:
                eng.process_audio([s], [audio[self.processed:self.processed + cc * 1280]])
            eng.encode([s], [is_final])
            self.processed += cc * 1280
        if eng.memory_len(s) == 0:
            return b""
        if not is_final and not self.inc:
            return b""
        eng.decoder_reset([s])
        if self.spec and not is_new and self.last:
            draft = [t for t in self.last if t not in (CFG.bos, CFG.eos)]
            (out,), acc = eng.decode_full([s], drafts=[draft])
            self.accepted.append((int(acc[0]), len(draft)))
            tokens = [CFG.bos] + out
        else:
            budget = min(int(math.ceil(float(np.float32(np.float32(len(audio)) / np.float32(16000.0)) * np.float32(self.mtps)))), 256)
            (out,), _ = eng.decode_full([s], max_tokens=[budget])
            tokens = [CFG.bos] + out + ([CFG.eos] if len(out) < budget else [])
        self.last = tokens
        return host_ref.sanitize_text(host_ref.tokens_to_text(self.vocab, tokens))

    def close(self):
        self.eng.close_stream(self.slot)


def test_streaming_arch_stream_flow(model_dir, engine):
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "transcription_interval": "0.3"})
    audio = make_audio(70, 16000 * 4 + 700)
    s = t.create_stream()
    t.start_stream(s)
    mirror = GlueMirror(engine)
    texts = []
    step = 9000
    for i in range(0, len(audio), step):
        t.add_audio(s, audio[i:i + step])
        lines = t.transcribe_stream(s)
        assert len(lines) == 1 and not lines[0].is_complete
        want = mirror.update(lines[0].audio_data, False)
        assert lines[0].text_bytes == want
        texts.append(want)
    t.stop_stream(s)
    final = t.transcribe_stream(s, flags=0)
    assert final[0].is_complete
    want = mirror.update(final[0].audio_data, True)
    assert final[0].text_bytes == want and len(want) > 0
    assert len(mirror.accepted) >= 5            # every pass after the first verified a draft
    assert any(a > 0 for a, _ in mirror.accepted)
    t.free_stream(s)
    mirror.close()
    t.close()


def test_streaming_arch_one_shot_and_batch(model_dir, engine):
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0"})
    clips = [make_audio(80 + i, n) for i, n in enumerate([16000 * 2, 40000, 1500, 1000, 16000 * 5 + 123])]
    single = [t.transcribe_without_streaming(c) for c in clips]
    for c, lines in zip(clips, single):
        assert len(lines) == 1 and lines[0].is_complete
        m = GlueMirror(engine)
        assert lines[0].text_bytes == m.update(lines[0].audio_data, True)
        m.close()
    assert single[3][0].text_bytes == b""       # 1000 samples -> 512 after the VAD: not one whole 1280 chunk
    # (texts can be empty even for long clips: half of the 512-entry synthetic vocabulary is <0xNN> byte tokens,
    #  which tokens_to_text skips)
    batch = t.transcribe_batch_without_streaming(clips)
    assert [b[0].text_bytes for b in batch] == [s[0].text_bytes for s in single]
    t.close()
    # a batch larger than the device's slot pool (max_streams) runs in waves of that size: same transcripts
    t3 = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "max_streams": "3"})
    many = [clips[i % len(clips)] for i in range(11)]
    got = t3.transcribe_batch_without_streaming(many)
    assert [g[0].text_bytes for g in got] == [single[i % len(clips)][0].text_bytes for i in range(11)]
    assert len(t3.transcribe_batch_without_streaming(many[:7])) == 7     # and again: the slots were handed back
    t3.close()


def test_streaming_arch_options(model_dir, engine):
    audio = make_audio(90, 16000 * 3)
    outs = {}
    for spec in ("true", "false"):
        t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING,
                            {"vad_threshold": "0", "use_speculative_decoding": spec, "transcription_interval": "0.2"})
        s = t.create_stream()
        t.start_stream(s)
        m = GlueMirror(engine, speculative=(spec == "true"))
        for i in range(0, len(audio), 8000):
            t.add_audio(s, audio[i:i + 8000])
            lines = t.transcribe_stream(s)
            assert lines[0].text_bytes == m.update(lines[0].audio_data, False)
        t.stop_stream(s)
        lines = t.transcribe_stream(s)
        outs[spec] = lines[0].text_bytes
        assert outs[spec] == m.update(lines[0].audio_data, True)
        assert (len(m.accepted) > 0) == (spec == "true")
        m.close()
        t.close()
    # decode_incomplete_lines=false: open lines carry no text, the closing update does
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "decode_incomplete_lines": "false"})
    s = t.create_stream()
    t.start_stream(s)
    m = GlueMirror(engine, decode_incomplete=False)
    t.add_audio(s, audio)
    lines = t.transcribe_stream(s)
    assert lines[0].text_bytes == b"" == m.update(lines[0].audio_data, False)
    t.stop_stream(s)
    lines = t.transcribe_stream(s)
    assert lines[0].is_complete and lines[0].text_bytes == m.update(lines[0].audio_data, True)
    assert m.last[0] == CFG.bos and len(m.last) > 1      # the closing update did decode
    m.close()
    t.close()
    with pytest.raises(api.MoonshineError):      # offline weights directory for a streaming arch
        api.Transcriber(str(model_dir) + "_missing", api.ARCH_TINY_STREAMING, {"vad_threshold": "0"})


def compile_terms(terms, boost):
    """Transcriber::set_keyterms (transcriber.cpp:250-288) restated: both spellings of every term, encoded with the
    tokenizer's text -> ids direction (longest match here: the synthetic vocabulary has no raw-byte block)."""
    from oracle.biaser_ref import ContextBiaser, text_to_tokens_bpe

    vocab = synthetic_vocab(CFG.vocab)
    b = ContextBiaser(boost)
    for term in terms:
        for variant in ContextBiaser.variants_for_term(term):
            b.add_token_sequence(text_to_tokens_bpe(vocab, variant.encode()))
    return b


def test_streaming_arch_keyterms(model_dir, engine):
    audio = make_audio(95, 16000 * 3)
    terms = ["qjx", " wvut ", "abcab"]
    plain_t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0"})
    plain = plain_t.transcribe_without_streaming(audio)[0].text_bytes
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING,
                        {"vad_threshold": "0", "keyterms": ",".join(terms), "keyterm_boost": "5.0"})
    b = compile_terms(terms, 5.0)
    assert b.sequence_count == 6
    md = max(b.depth)
    engine.set_bias(b.children, b.depth, [float(b.bonus_for_depth(d)) for d in range(md + 2)])
    try:
        lines = t.transcribe_without_streaming(audio)
        m = GlueMirror(engine)
        assert lines[0].text_bytes == m.update(lines[0].audio_data, True)
        m.close()
        # runtime call: clearing the terms restores the unbiased transcript, setting them again the biased one
        biased = lines[0].text_bytes
        lib = api.lib()
        assert lib.moonshine_transcriber_set_keyterms(t.handle, b"") == 0
        assert t.transcribe_without_streaming(audio)[0].text_bytes == plain
        assert lib.moonshine_transcriber_set_keyterms(t.handle, ",".join(terms).encode()) == 0
        assert t.transcribe_without_streaming(audio)[0].text_bytes == biased
    finally:
        engine.set_bias(None)
    t.close()
    plain_t.close()


def test_streaming_arch_context(model_dir, engine):
    """The `context` option / moonshine_transcriber_set_context: key terms picked out of a passage by the subword
    count of the model's own tokenizer (reference transcriber.cpp:201-248), then compiled like explicit key terms."""
    from oracle.biaser_ref import extract_terms, text_to_tokens_bpe

    vocab = synthetic_vocab(CFG.vocab)

    def count(word: bytes) -> int:
        try:
            return len(text_to_tokens_bpe(vocab, word))
        except ValueError:
            return 0

    passage = "the qjx wvut and the qjx again; abcab's route to Évrémonde, 42 things"
    terms = [w.decode() for w in extract_terms(passage, 0, count)]
    assert "qjx" in terms and len(terms) >= 3   # (the synthetic vocabulary has no whole-word pieces: "the" qualifies too)
    audio = make_audio(96, 16000 * 3)
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "context": passage, "keyterm_boost": "5.0"})
    b = compile_terms(terms, 5.0)
    engine.set_bias(b.children, b.depth, [float(b.bonus_for_depth(d)) for d in range(max(b.depth) + 2)])
    try:
        lines = t.transcribe_without_streaming(audio)
        m = GlueMirror(engine)
        assert lines[0].text_bytes == m.update(lines[0].audio_data, True)
        m.close()
        # runtime call with a cap of one term
        lib = api.lib()
        assert lib.moonshine_transcriber_set_context(t.handle, passage.encode(), 1) == 0
        b1 = compile_terms(terms[:1], 5.0)
        engine.set_bias(b1.children, b1.depth, [float(b1.bonus_for_depth(d)) for d in range(max(b1.depth) + 2)])
        lines = t.transcribe_without_streaming(audio)
        m = GlueMirror(engine)
        assert lines[0].text_bytes == m.update(lines[0].audio_data, True)
        m.close()
    finally:
        engine.set_bias(None)
    t.close()


def test_streaming_arch_word_timestamps(model_dir, engine):
    """Option word_timestamps on a streaming architecture (reference core/transcriber.cpp:1028-1068): every decoded line
    carries words = the oracle's align_words applied to the engine's own cross-attention of the line's final tokens
    (inputs BOS, t1, ... = all tokens but the last; seconds per frame = segment duration / memory frames), shifted by the
    segment start; one-shot call, batch call and the stream API agree; without the option there are no words."""
    from oracle import word_align_ref as wa

    vocab = synthetic_vocab(CFG.vocab)
    clips = [make_audio(700 + i, n) for i, n in enumerate([16000 * 3, 40000, 16000 * 4 + 640])]
    t = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "word_timestamps": "true"})
    plain = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0"})
    try:
        single = [t.transcribe_without_streaming(c) for c in clips]
        batch = t.transcribe_batch_without_streaming(clips)
        for c, lines, blines in zip(clips, single, batch):
            assert len(lines) == 1 and lines[0].is_complete
            m = GlueMirror(engine)
            seg = lines[0].audio_data
            assert lines[0].text_bytes == m.update(seg, True)
            toks = m.last
            att = engine.cross_attention(m.slot, toks[:-1])
            spf = float(np.float32(np.float32(len(seg)) / np.float32(16000.0)) / np.float32(att.shape[2]))
            want = wa.align_words(att, toks, spf, vocab, host_ref.tokens_to_text)
            got = lines[0].words
            assert [w[0] for w in got] == [w["text"] for w in want]
            np.testing.assert_allclose([w[1] for w in got], [w["start"] for w in want], rtol=0, atol=1e-6)
            np.testing.assert_allclose([w[2] for w in got], [w["end"] for w in want], rtol=0, atol=1e-6)
            assert got == blines[0].words
            prev = -1.0
            for _, start, end, conf in got:
                assert end >= start >= prev and 0.0 <= conf <= 1.0 and end <= len(seg) / 16000 + 1e-3
                prev = start
            m.close()
            assert plain.transcribe_without_streaming(c)[0].words == []
            assert plain.transcribe_without_streaming(c)[0].text_bytes == lines[0].text_bytes   # the capture changes no text
        assert any(len(l[0].words) > 0 for l in single)
        # stream API: words on the open line too (the reference fills them on every update), final words == one-shot words
        s = t.create_stream()
        t.start_stream(s)
        ci = int(np.argmax([len(l[0].words) for l in single]))   # the clip whose line has the most words
        c = clips[ci]
        open_words = 0
        for k in range(0, len(c), 16000):
            t.add_audio(s, c[k:k + 16000])
            lines = t.transcribe_stream(s, api.FLAG_FORCE_UPDATE)
            open_words += len(lines[0].words)
        t.stop_stream(s)
        final = t.transcribe_stream(s, api.FLAG_FORCE_UPDATE)
        assert final[0].is_complete
        if final[0].text_bytes == single[ci][0].text_bytes:     # same ids (no near-tie resolved differently on the way)
            assert [w[0] for w in final[0].words] == [w[0] for w in single[ci][0].words]
            assert open_words > 0
        t.free_stream(s)
    finally:
        t.close()
        plain.close()


def test_streams_sharded_over_devices_equal_one_device(model_dir):
    """Options `devices` / `num_gpus` on a STREAMING architecture (streaming state is per segment, so streams
    shard as clips do): one model -- engine, slots, own copy of the weights -- per listed GPU, a stream's line state placed
    on the device holding the fewest lines, every device running its batch of an update on its own host thread, no
    collective.  The box has one GPU, so the list names it twice: two engines, two slot pools, the same code path as two
    GPUs.  Transcripts must equal the one-device run, for the batch call (incl. waves beyond one device's slot pool) and
    for concurrent live streams updated in lock-step."""
    clips = [make_audio(700 + i, n) for i, n in enumerate([32000, 40000, 1500, 52000, 16000 * 5 + 123, 24000, 61000])]
    one = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0"})
    want = [[l.text_bytes for l in r] for r in one.transcribe_batch_without_streaming(clips)]
    two = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "devices": "0,0", "max_streams": "2"})
    for _ in range(2):   # 7 clips on 2 x 2 slots: two waves; second call: slots handed back on both devices
        got = [[l.text_bytes for l in r] for r in two.transcribe_batch_without_streaming(clips)]
        assert got == want

    def live(t, n_streams):
        ids = [t.create_stream() for _ in range(n_streams)]
        for s in ids:
            t.start_stream(s)
        texts = [[] for _ in ids]
        for off in range(0, 48000, 8000):
            for k, s in enumerate(ids):
                t.add_audio(s, clips[k % len(clips)][off:off + 8000])
            for k, s in enumerate(ids):
                texts[k].append([l.text_bytes for l in t.transcribe_stream(s, api.FLAG_FORCE_UPDATE)])
        for k, s in enumerate(ids):
            t.stop_stream(s)
            texts[k].append([l.text_bytes for l in t.transcribe_stream(s, api.FLAG_FORCE_UPDATE)])
            t.free_stream(s)
        return texts

    two4 = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "devices": "0,0", "max_streams": "2"})
    assert live(two4, 4) == live(one, 4)        # 4 live streams = 2 per device
    # the shape of BASELINE.json's 8-GPU configuration on the one GPU of the box: eight entries = eight engines, eight slot
    # pools, eight host threads per update (per-device structure locks: their warm-up does not serialise)
    eight = api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "devices": "0,0,0,0,0,0,0,0", "max_streams": "1"})
    assert [[l.text_bytes for l in r] for r in eight.transcribe_batch_without_streaming(clips)] == want
    eight.close()
    with pytest.raises(api.MoonshineError):     # more GPUs than the box has: the load says so
        api.Transcriber(model_dir, api.ARCH_TINY_STREAMING, {"vad_threshold": "0", "num_gpus": "2"})
    for t in (one, two, two4):
        t.close()
