"""Fuzz-diff of libmoonshine.so's host byte / integer / sample helpers (msh_host_*, the code the MI355X Transcriber
runs) against the REFERENCE's own sources compiled into oracle/_ref/libmoonshine_ref_host.so (oracle/build_ref.py,
oracle/ref_shim.cpp): resampler.cpp, bin-tokenizer.cpp, word-alignment.cpp, context-biaser.cpp, context-extractor.cpp,
debug-utils.cpp (WAV).  Same arguments into both libraries; results must be byte-identical (floats bit-identical).

The reference library is test infrastructure: built here when /root/reference exists, shipped prebuilt to the GPU box
(where these CPU tests do not run anyway).  No reference library -> the tests are skipped, not passed.
"""
import ctypes as C
import os

import numpy as np
import pytest

from moonshine_amd.hip_api import load_library
from moonshine_amd.synth import encode_tokenizer_bin, synthetic_vocab
from oracle.build_ref import build_ref, ref_library_path

SPACE = "▁".encode()
vp, u64, i32, i64, f32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_int64, C.c_float

SIGS = {
    "tokens_to_text": (i64, [vp, u64, vp, u64, vp, u64]),
    "text_to_tokens": (i64, [vp, u64, vp, u64, C.c_char_p, i32, vp, u64]),
    "biaser_bonuses": (i64, [vp, vp, u64, f32, vp, u64, vp, u64]),
    "context_terms": (i64, [vp, u64, vp, u64, i32, vp, u64]),
    "dtw": (i64, [vp, i32, i32, vp, vp, u64]),
    "median_filter": (i32, [vp, u64, i32, i32]),
    "align_words": (i64, [vp, u64, vp, i32, i32, i32, vp, u64, f32, vp, u64, vp, u64]),
    "load_wav": (i64, [C.c_char_p, vp, u64, vp]),
    "save_wav": (i32, [C.c_char_p, vp, u64, i32]),
    "resample": (i64, [vp, u64, f32, f32, vp, u64]),
}


class Pair:
    """The same function in both libraries."""

    def __init__(self, ours, ref):
        self.fns = {}
        for name, (res, args) in SIGS.items():
            a, b = getattr(ours, "msh_host_" + name), getattr(ref, "ref_host_" + name)
            for f in (a, b):
                f.restype, f.argtypes = res, args
            self.fns[name] = (a, b)

    def both(self, name, make_args):
        """make_args() builds a fresh argument tuple + a function returning the observable outputs."""
        out = []
        for f in self.fns[name]:
            args, observe = make_args()
            rc = f(*args)
            out.append((rc, observe(rc)))
        return out


@pytest.fixture(scope="module")
def pair():
    path = build_ref() or ref_library_path()
    if path is None:
        pytest.skip("no oracle/_ref library (reference sources absent and nothing prebuilt)")
    return Pair(load_library(), C.CDLL(path))


def bpe_vocab(rng, n_pieces=600):
    """Control tokens, the 256 raw bytes, the word marker, then random 'merged' pieces over a small alphabet -- the
    layout the reference's BPE encoder needs (bin-tokenizer-test.cpp:17-45)."""
    v = [b"<unk>", b"<s>", b"</s>"] + [bytes([i]) for i in range(256)] + [SPACE]
    alpha = b"abcdefghij"
    seen = set(v)
    while len(v) < 260 + n_pieces:
        n = int(rng.integers(2, 6))
        p = bytes(rng.choice(list(alpha), n).tolist())
        if rng.random() < 0.3:
            p = SPACE + p
        if p not in seen:
            seen.add(p)
            v.append(p)
    return v


def arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def test_resampler_bit_identical(pair):
    rng = np.random.default_rng(0)
    rates = [(48000, 16000), (44100, 16000), (22050, 16000), (24000, 16000), (8000, 16000), (11025, 16000), (16000, 16000),
             (16000, 48000), (32000, 16000), (96000, 16000), (12345, 16000)]
    for k in range(120):
        ir, orate = rates[k % len(rates)]
        n = int(rng.integers(0, 6000)) if k % 7 else int(rng.integers(0, 4))
        x = arr(rng.standard_normal(n) * 0.3, np.float32)

        def mk():
            out = np.full(n * 8 + 64, np.nan, np.float32)
            return (x.ctypes.data, n, float(ir), float(orate), out.ctypes.data, out.size), lambda rc: out[:max(rc, 0)].tobytes()

        a, b = pair.both("resample", mk)
        assert a == b, (ir, orate, n)


def test_tokenizer_decode_and_encode_identical(pair):
    rng = np.random.default_rng(1)
    vocabs = [synthetic_vocab(4096), bpe_vocab(rng), [b"<unk>", b"<s>", b"</s>", b"_", b"a", b"b", b"ab", b"_ab", b"abb"]]
    for vi, vocab in enumerate(vocabs):
        blob = encode_tokenizer_bin(vocab)
        for trial in range(150):
            ids = arr(rng.integers(0, len(vocab), int(rng.integers(0, 30))), np.int32)

            def mk():
                buf = C.create_string_buffer(1 << 15)
                return (blob, len(blob), ids.ctypes.data, len(ids), C.addressof(buf), len(buf)), lambda rc: buf.raw[:max(rc, 0)]

            a, b = pair.both("tokens_to_text", mk)
            assert a == b, (vi, ids.tolist())
        # text -> ids, both encodings; texts are built from the vocabulary's own alphabet, raw bytes and the marker
        alpha = b"abcdefghij _" if vi else b"abcdefghijklmnopqrstuvwxyz "
        marker = b"_" if vi == 2 else SPACE
        for trial in range(200):
            n = int(rng.integers(0, 24))
            text = bytes(rng.choice(list(alpha), n).tolist()) if n else b""
            if trial % 9 == 0:
                text += "é中".encode() + bytes([int(rng.integers(1, 256))])
            for bpe in (0, 1):
                def mk():
                    out = np.full(256, -7, np.int32)
                    return (blob, len(blob), text, len(text), marker, bpe, out.ctypes.data, out.size), lambda rc: out[:max(rc, 0)].tolist()

                a, b = pair.both("text_to_tokens", mk)
                assert (a[0] < 0) == (b[0] < 0), (vi, text, bpe, a, b)
                if a[0] >= 0:
                    assert a == b, (vi, text, bpe)


REAL_TOKENIZER = os.path.join(os.path.dirname(__file__), "golden", "tiny_en_tokenizer.bin")
BECKETT = b"Ever tried. Ever failed. No matter. Try again. Fail again. Fail better."


def test_shipped_tiny_en_tokenizer_identical(pair):
    """The one model artefact the reference checkout holds: language-bindings/python/src/moonshine_voice/assets/tiny-en/
    tokenizer.bin (32,768 entries: <unk> <s> </s>, the 256 byte fallbacks, learned pieces, 768 <<ST_n>> specials), committed as
    the data fixture tests/golden/tiny_en_tokenizer.bin.  tokens_to_text over random ids incl. specials and byte fallbacks,
    text_to_tokens (longest match AND byte-pair replay) over English, raw bytes and the marker: byte-identical to the
    reference's bin-tokenizer.cpp (:46-66 file format, :277-404 encoders, :406-426 decoder) compiled in oracle/_ref."""
    blob = open(REAL_TOKENIZER, "rb").read()
    ref_copy = "/root/reference/language-bindings/python/src/moonshine_voice/assets/tiny-en/tokenizer.bin"
    if os.path.exists(ref_copy):
        assert open(ref_copy, "rb").read() == blob          # the fixture IS the reference's file
    rng = np.random.default_rng(11)
    V = 32768
    pools = [lambda n: rng.integers(0, V, n), lambda n: rng.integers(0, 259, n), lambda n: rng.integers(31990, V, n),
             lambda n: rng.integers(259, 32000, n)]
    for trial in range(400):
        ids = arr(pools[trial % 4](int(rng.integers(0, 70))), np.int32)
        if trial % 50 == 7:
            ids[len(ids) // 2:len(ids) // 2 + 1] = 31353     # the file's one EMPTY entry: the reference throws, both sides < 0
        if trial % 50 == 9:
            ids[:1] = 25299                                  # "<>": two bytes, NOT skipped as a special (needs > 2)

        def mk():
            buf = C.create_string_buffer(1 << 15)
            return (blob, len(blob), ids.ctypes.data, len(ids), C.addressof(buf), len(buf)), lambda rc: buf.raw[:max(rc, 0)]

        a, b = pair.both("tokens_to_text", mk)
        assert a == b, ids.tolist()
    words = (BECKETT.decode() + " the quick brown fox jumps over the lazy dog It was the best of times, it was the worst of times "
             "Kubernetes IPv6 naïve café 東京 don't e-mail 3.14159 $100 A.I. <s> </s> <unk> <<ST_5>> <>").split(" ")
    texts = [BECKETT, b"", b" ", b"  two  spaces ", SPACE, SPACE + b"Ever" + SPACE + b"tried.", b"\x00\x01\xff\xfe", "▁▁".encode()]
    for trial in range(300):
        n = int(rng.integers(1, 14))
        t = " ".join(words[int(rng.integers(len(words)))] for _ in range(n)).encode()
        if trial % 5 == 0:
            t += bytes(rng.integers(0, 256, int(rng.integers(1, 6))).tolist())
        if trial % 7 == 0:
            t = t.replace(b" ", SPACE, 1)
        texts.append(t)
    for text in texts:
        for bpe in (0, 1):
            def mk():
                out = np.full(1024, -7, np.int32)
                return (blob, len(blob), text, len(text), SPACE, bpe, out.ctypes.data, out.size), lambda rc: out[:max(rc, 0)].tolist()

            a, b = pair.both("text_to_tokens", mk)
            assert (a[0] < 0) == (b[0] < 0), (text, bpe, a, b)
            if a[0] >= 0:
                assert a == b, (text, bpe)
                # and back through BOTH decoders: the round trip of what the encoder produced
                ids = arr(a[1], np.int32)

                def mk2():
                    buf = C.create_string_buffer(1 << 15)
                    return (blob, len(blob), ids.ctypes.data, len(ids), C.addressof(buf), len(buf)), lambda rc: buf.raw[:max(rc, 0)]

                x, y = pair.both("tokens_to_text", mk2)
                assert x == y, (text, bpe)
    # known answer: the sentence beckett.wav says survives ids -> text under both encodings
    for bpe in (0, 1):
        def mk():
            out = np.full(256, -7, np.int32)
            return (blob, len(blob), BECKETT, len(BECKETT), SPACE, bpe, out.ctypes.data, out.size), lambda rc: out[:max(rc, 0)].tolist()

        a, b = pair.both("text_to_tokens", mk)
        assert a == b and a[0] > 0
        ids = arr([1] + a[1] + [2], np.int32)               # <s> ... </s> as the greedy loop emits them

        def mk2():
            buf = C.create_string_buffer(1 << 12)
            return (blob, len(blob), ids.ctypes.data, len(ids), C.addressof(buf), len(buf)), lambda rc: buf.raw[:max(rc, 0)]

        x, y = pair.both("tokens_to_text", mk2)
        assert x == y and x[1] == BECKETT, (bpe, x)


def test_context_biaser_identical(pair):
    rng = np.random.default_rng(2)
    V = 300
    for trial in range(150):
        n_seqs = int(rng.integers(0, 9))
        seqs = [rng.integers(0, 12 if trial % 2 else V, int(rng.integers(1, 6))).tolist() for _ in range(n_seqs)]
        flat = arr([t for s in seqs for t in s] or [0], np.int32)
        lens = arr([len(s) for s in seqs] or [0], np.int32)
        prefix = arr(rng.integers(0, 12 if trial % 2 else V, int(rng.integers(0, 8))), np.int32)
        boost = float(rng.choice([0.5, 2.0, 4.0]))
        base = arr(rng.standard_normal(V), np.float32)

        def mk():
            out = base.copy()
            return (flat.ctypes.data, lens.ctypes.data, n_seqs, boost, prefix.ctypes.data, len(prefix), out.ctypes.data, V), lambda rc: out.tobytes()

        a, b = pair.both("biaser_bonuses", mk)
        assert a == b, (seqs, prefix.tolist(), boost)


def test_context_extractor_identical(pair):
    rng = np.random.default_rng(3)
    vocab = bpe_vocab(rng, 900)
    blob = encode_tokenizer_bin(vocab)
    words = ["abc", "Kubernetes", "ab", "jig", "Abe", "cafe", "hidea", "bad", "IPv6", "x", "gadj", "beef's", "dig-ace", "école", "façade",
             "BACH", "the", "decaf", "jab", "abcdefghij"]
    seps = [" ", ", ", ". ", "\n", "; ", " - ", "  ", "! ", " (", ") "]
    for trial in range(80):
        n = int(rng.integers(0, 60))
        text = "".join(words[int(rng.integers(len(words)))] + seps[int(rng.integers(len(seps)))] for _ in range(n)).encode()
        max_terms = int(rng.choice([0, -1, 1, 3, 10, 200]))

        def mk():
            buf = C.create_string_buffer(1 << 15)
            return (blob, len(blob), text, len(text), max_terms, C.addressof(buf), len(buf)), lambda rc: buf.raw[:max(rc, 0)]

        a, b = pair.both("context_terms", mk)
        assert a == b, (text, max_terms)


def test_dtw_and_median_filter_identical(pair):
    rng = np.random.default_rng(4)
    for trial in range(120):
        n, m = int(rng.integers(1, 24)), int(rng.integers(1, 80))
        cost = arr(rng.standard_normal((n, m)) if trial % 3 else np.round(rng.standard_normal((n, m))), np.float32)   # ties too

        def mk():
            ti, tj = np.full(n + m + 4, -1, np.int32), np.full(n + m + 4, -1, np.int32)
            return (cost.ctypes.data, n, m, ti.ctypes.data, tj.ctypes.data, ti.size), lambda rc: (ti.tolist(), tj.tolist())

        a, b = pair.both("dtw", mk)
        assert a == b, (n, m)
    for trial in range(120):
        rows, w = int(rng.integers(1, 6)), int(rng.integers(1, 40))
        width = int(rng.choice([1, 3, 5, 7, 9]))
        x = arr(rng.standard_normal((rows, w)), np.float32)

        def mk():
            d = x.copy()
            return (d.ctypes.data, rows, w, width), lambda rc: d.tobytes()

        a, b = pair.both("median_filter", mk)
        assert a == b, (rows, w, width)


def test_align_words_identical(pair):
    rng = np.random.default_rng(5)
    vocab = synthetic_vocab(2048)
    blob = encode_tokenizer_bin(vocab)
    for trial in range(60):
        heads, steps, frames = int(rng.integers(1, 9)), int(rng.integers(1, 20)), int(rng.integers(2, 120))
        att = rng.random((heads, steps, frames)).astype(np.float32) ** 4
        # a moving ridge, as real cross-attention has
        for s in range(steps):
            att[:, s, min(frames - 1, s * frames // steps)] += 1.0
        att /= att.sum(-1, keepdims=True)
        att = arr(att, np.float32)
        body = rng.integers(259, 2048, steps - 1).tolist() if steps > 1 else []
        toks = arr([1] + body + ([2] if trial % 2 else []), np.int32)[: steps + 1]
        spf = float(rng.choice([0.02, 0.024, 0.0241]))

        def mk():
            text = C.create_string_buffer(1 << 14)
            times = np.full(3 * 64, np.nan, np.float32)
            return ((blob, len(blob), att.ctypes.data, heads, steps, frames, toks.ctypes.data, len(toks), spf, C.addressof(text), len(text),
                     times.ctypes.data, 64), lambda rc: (text.value, times[: 3 * max(rc, 0)].tobytes()))

        a, b = pair.both("align_words", mk)
        assert a == b, (heads, steps, frames, toks.tolist())


def test_wav_io_identical(pair, tmp_path):
    rng = np.random.default_rng(6)
    for trial in range(12):
        n = int(rng.integers(0, 5000))
        x = arr(np.clip(rng.standard_normal(n) * 0.5, -1.2, 1.2), np.float32)     # includes clipping
        rate = int(rng.choice([16000, 44100, 8000]))
        paths = [str(tmp_path / f"ours_{trial}.wav").encode(), str(tmp_path / f"ref_{trial}.wav").encode()]
        rcs = [f(p, x.ctypes.data, n, rate) for f, p in zip(pair.fns["save_wav"], paths)]
        assert rcs[0] == rcs[1] == 0
        assert open(paths[0], "rb").read() == open(paths[1], "rb").read()
        # each loader on the OTHER writer's file
        res = []
        for f, p in zip(pair.fns["load_wav"], paths[::-1]):
            out = np.full(n + 16, np.nan, np.float32)
            r = C.c_int32(0)
            cnt = f(p, out.ctypes.data, out.size, C.addressof(r))
            res.append((cnt, r.value, out[:max(cnt, 0)].tobytes()))
        assert res[0] == res[1] and res[0][0] == n and res[0][1] == rate
    # malformed files: both refuse
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
    r = C.c_int32(0)
    assert all(f(str(bad).encode(), None, 0, C.addressof(r)) < 0 for f in pair.fns["load_wav"])


def test_sanitize_text_identical(pair):
    """Transcriber::sanitize_text (core/transcriber.cpp:1489-1543, lifted into oracle/_ref at build time) against
    msh_host_sanitize_utf8 on random byte strings: valid UTF-8 of every length, truncated sequences at the end and in the
    middle, stray continuation bytes, overlong-looking starts, 0xF5-0xFF -- byte for byte."""
    ours = load_library().msh_host_sanitize_utf8
    ref = C.CDLL(ref_library_path()).ref_host_sanitize_utf8
    for f in (ours, ref):
        f.restype, f.argtypes = i64, [C.c_char_p, u64, C.c_char_p, u64]
    if ref(b"a", 1, C.create_string_buffer(8), 8) < 0:
        pytest.skip("oracle/_ref was built without the sanitize_text extraction")
    rng = np.random.default_rng(11)
    pieces = [b"a", b"\xc3\xa9", b"\xe2\x82\xac", b"\xf0\x9f\x98\x80", b"\x80", b"\xbf", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\xf5", b"\xff",
              b"\xc0\xaf", b"\xed\xa0\x80", b" "]
    cases = [b"", b"\xf0\x9f", b"\xe2", b"plain ascii"]
    for _ in range(1500):
        if rng.random() < 0.5:
            raw = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), int(rng.integers(0, 12))))
        else:
            raw = bytes(rng.integers(1, 256, int(rng.integers(0, 20))).tolist())   # (no NUL: the reference takes a C string)
        cases.append(raw)
    for raw in cases:
        a, b = C.create_string_buffer(len(raw) + 8), C.create_string_buffer(len(raw) + 8)
        na, nb = ours(raw, len(raw), a, len(a)), ref(raw, len(raw), b, len(b))
        assert na == nb and a.raw[:na] == b.raw[:nb], raw


def _fnv(x: np.ndarray) -> int:
    h = 1469598103934665603
    for byte in x.tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h - (1 << 64) if h >= (1 << 63) else h


def test_vad_state_machine_identical(pair):
    """VoiceActivityDetector (core/voice-activity-detector.cpp:69-199, compiled as it is; its SileroVad::predict stubbed with
    the supplied per-hop probabilities) against the library's detector fed the same probabilities
    (msh_host_vad_segments_from_probs, the host half of the device-VAD path) and, for threshold 0, against
    msh_host_vad_segments with chunked feeding and other sample rates: segment starts, lengths, completion flags and the
    segments' audio (hash) must be identical -- probability ring, look-behind, max-segment fade, start / continue / end."""
    ours_lib = load_library()
    refl = C.CDLL(ref_library_path())
    ref = refl.ref_host_vad_segments
    ref.restype = i64
    ref.argtypes = [f32, i32, i32, u64, u64, vp, u64, i32, u64, vp, u64, vp, u64]
    fp = ours_lib.msh_host_vad_segments_from_probs
    fp.restype = i64
    fp.argtypes = [C.c_char_p, u64, f32, i32, i32, u64, u64, u64, vp, u64, vp, u64, vp, u64]
    f0 = ours_lib.msh_host_vad_segments
    f0.restype = i64
    f0.argtypes = [vp, u64, f32, i32, i32, u64, u64, u64, vp, u64, i32, u64, vp, u64]
    rng = np.random.default_rng(5)
    # the library's hook wants a Silero weights blob (the detector it builds is the Transcriber's); with the probabilities
    # supplied the network itself never runs
    import tempfile

    from moonshine_amd.synth import make_silero_weights, save_safetensors

    with tempfile.TemporaryDirectory() as d:
        wp = os.path.join(d, "silero_vad.safetensors")
        save_safetensors(wp, make_silero_weights(2))
        blob = open(wp, "rb").read()

    def ours_rows(bounds, k, audio16):
        rows = []
        for i in range(k):
            s, cnt, comp = int(bounds[3 * i]), int(bounds[3 * i + 1]), int(bounds[3 * i + 2])
            rows.append((s, cnt, comp, _fnv(audio16[s:s + cnt])))
        return rows

    checked = 0
    for trial in range(60):
        hop = 512   # (with Silero on, this build pins the hop to the network's 512-sample window and says so at load)
        n = int(rng.integers(0, 40)) * hop + int(rng.integers(0, hop))
        n += int(rng.integers(0, 8)) * 16000 if trial % 3 == 0 else 0
        audio = (rng.standard_normal(n) * 0.1).astype(np.float32)
        window = int(rng.choice([1, 4, 16, 32]))
        look = int(rng.choice([512, 4096, 8192]))   # (the reference shifts its look-behind ring by one hop in place: UB below one hop)
        max_seg = int(rng.choice([0, 3 * 16000, 15 * 16000, 1600]))
        thr = float(rng.choice([0.5, 0.3, 0.7]))
        hops = n // hop
        # speech-like probability tracks: runs of high / low values with noise, so that segments start and end
        probs = np.clip(np.repeat(rng.random(hops // 7 + 2), 7)[:hops] * 1.2 - 0.1 + rng.normal(0, 0.1, hops), 0, 1).astype(np.float32)
        outr = np.zeros(4 * 256, np.int64)
        kr = ref(thr, window, hop, look, max_seg, audio.ctypes.data, n, 16000, 0, probs.ctypes.data, hops, outr.ctypes.data, 256)
        outo = np.zeros(3 * 256, np.int64)
        ko = fp(blob, len(blob), thr, window, hop, look, max_seg, 0, audio.ctypes.data, n, probs.ctypes.data, hops, outo.ctypes.data, 256)
        assert ko == kr, (trial, ko, kr)
        want = [tuple(int(v) for v in outr[4 * i:4 * i + 4]) for i in range(kr)]
        assert ours_rows(outo, ko, audio) == want, trial
        checked += kr
    assert checked > 40   # the tracks do cut segments
    # threshold 0 (all audio is speech): chunked feeding, other sample rates (the detector resamples), tiny max_segment
    for trial in range(40):
        rate = int(rng.choice([16000, 16000, 48000, 24000, 8000]))
        n = int(rng.integers(0, 6 * rate))
        audio = (rng.standard_normal(n) * 0.1).astype(np.float32)
        chunk = int(rng.choice([0, 160, 1000, 4096, 7777]))
        look = int(rng.choice([512, 4096]))
        max_seg = int(rng.choice([0, 16000, 15 * 16000]))
        outr = np.zeros(4 * 256, np.int64)
        kr = ref(0.0, 32, 512, look, max_seg, audio.ctypes.data, n, rate, chunk, None, 0, outr.ctypes.data, 256)
        outo = np.zeros(3 * 256, np.int64)
        ko = f0(None, 0, 0.0, 32, 512, look, max_seg, 0, audio.ctypes.data, n, rate, chunk if chunk else max(n, 1), outo.ctypes.data, 256)
        assert ko == kr, (trial, rate, n, chunk, ko, kr)
        for i in range(kr):   # the audio of a resampled segment lives inside the detector: compare starts / lengths / flags
            assert tuple(int(v) for v in outo[3 * i:3 * i + 3]) == tuple(int(v) for v in outr[4 * i:4 * i + 3]), (trial, i)
