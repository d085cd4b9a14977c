"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, struct layouts match
the reference ABI, and the host-side byte / integer work (tokenizer, UTF-8 repair, resampler, VAD
segmentation, option parsing, handle / error rules, stream flags) is bit-exact with the oracle
restatement of the reference (oracle/host_ref.py)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from moonshine_amd import api
from moonshine_amd.hip_api import LIB_PATH, load_library
from oracle import host_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def test_library_exports_every_declared_symbol():
    """The product library exports every symbol of include/moonshine-c-api.h and include/moonshine_hip.h and NONE of the
    development hooks; those (include/moonshine_hip_dev.h, msh_test_*) live in libmoonshine_dev.so, built from the same
    objects + csrc/dev_hooks.cpp."""
    lib = C.CDLL(LIB_PATH)
    dev = C.CDLL(os.path.join(os.path.dirname(LIB_PATH), "libmoonshine_dev.so"))

    def declared_in(h):
        txt = open(os.path.join(INCLUDE, h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"#define[^\n]*", "", txt)
        return re.findall(r"(?:MSH_EXPORT|MOONSHINE_EXPORT)[^;(]*?\b(\w+)\s*\(", txt)

    declared, hooks = [], []
    for h in sorted(os.listdir(INCLUDE)):
        (hooks if h == "moonshine_hip_dev.h" else declared).extend(declared_in(h))
    assert len(declared) >= 40 and len(hooks) >= 8
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert hasattr(dev, name), f"{name} missing from the development library"
    for name in hooks:
        assert name.startswith("msh_test_")
        assert hasattr(dev, name), f"{name} declared in moonshine_hip_dev.h but not exported by libmoonshine_dev.so"
        assert not hasattr(lib, name), f"development hook {name} is exported by the product library"
    assert set(api.C_API_SYMBOLS) <= set(declared)


def test_sysfs_cpu_list_parser():
    """host_utils.cpp parse_cpu_list: what pins a lane's host thread to its GPU's NUMA node when one process drives several GPUs
    (options num_gpus / devices)."""
    lib = load_library()
    lib.msh_host_parse_cpu_list.restype = C.c_int64
    lib.msh_host_parse_cpu_list.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    for text, want in [(b"0-3,8-9\n", [0, 1, 2, 3, 8, 9]), (b"5", [5]), (b"", []), (b"0-63,128-191", list(range(64)) + list(range(128, 192)))]:
        out = np.full(256, -1, np.int32)
        n = lib.msh_host_parse_cpu_list(text, out.ctypes.data, out.size)
        assert n == len(want) and out[:n].tolist() == want


def test_struct_layout_matches_reference_abi(tmp_path):
    """sizeof / offsetof of the public structs as gcc sees include/moonshine-c-api.h must equal the
    ctypes mirror (which pins the sizes the reference binding pins: 24 / 40 / 88 / 16)."""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "moonshine-c-api.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu %zu\\n", sizeof(struct transcript_word_t), sizeof(struct speaker_span_t), sizeof(struct transcript_line_t), sizeof(struct transcript_t), sizeof(struct moonshine_option_t));\n'
        'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(struct transcript_line_t, audio_data_count), offsetof(struct transcript_line_t, start_time), offsetof(struct transcript_line_t, id), offsetof(struct transcript_line_t, is_complete), offsetof(struct transcript_line_t, speaker_spans), offsetof(struct transcript_line_t, last_transcription_latency_ms), offsetof(struct transcript_line_t, words), offsetof(struct transcript_line_t, word_count));\n'
        "return 0;}\n"
    )
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", INCLUDE, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert [int(x) for x in out[0].split()] == [24, 40, 88, 16, 16]
    L = api.TranscriptLineC
    want = [getattr(L, f).offset for f in ("audio_data_count", "start_time", "id", "is_complete", "speaker_spans", "last_transcription_latency_ms", "words", "word_count")]
    assert [int(x) for x in out[1].split()] == want


def _tok_text(blob: bytes, ids) -> bytes:
    lib = load_library()
    ids = np.asarray(ids, np.int32)
    buf = C.create_string_buffer(1 << 16)
    n = lib.msh_host_tokens_to_text(blob, len(blob), ids.ctypes.data, len(ids), buf, len(buf))
    assert n >= 0
    return buf.raw[:n]


def test_tokenizer_and_sanitize_bit_exact():
    vocab = host_ref.synthetic_vocab(4096)
    blob = host_ref.encode_tokenizer_bin(vocab)
    assert host_ref.decode_tokenizer_bin(blob) == vocab
    rng = np.random.default_rng(0)
    lib = load_library()
    special = [0, 1, 2, 3, 258, 997, 1013, 1021, 1031, 1994, 2026, 2042, 3063]  # <..> specials, long / multi-byte / broken pieces
    for trial in range(200):
        n = int(rng.integers(0, 40))
        ids = rng.integers(0, 4096, n).tolist() + [special[trial % len(special)]]
        rng.shuffle(ids)
        want = host_ref.tokens_to_text(vocab, ids)
        got = _tok_text(blob, ids)
        assert got == want
        buf = C.create_string_buffer(len(got) + 8)
        m = lib.msh_host_sanitize_utf8(got, len(got), buf, len(buf))
        assert buf.raw[:m] == host_ref.sanitize_text(want)
    # random byte strings through the sanitizer, including truncated sequences at the very end
    for trial in range(300):
        raw = bytes(rng.integers(0, 256, int(rng.integers(0, 24))).tolist())
        raw = raw.replace(b"\x00", b"\x01")  # the reference operates on a C string
        buf = C.create_string_buffer(len(raw) + 8)
        m = lib.msh_host_sanitize_utf8(raw, len(raw), buf, len(buf))
        assert buf.raw[:m] == host_ref.sanitize_text(raw), raw
    # an empty entry is an invalid token (reference throws): negative status
    bad = host_ref.encode_tokenizer_bin([b"<unk>", b"", b"ab"])
    ids = np.asarray([2, 1], np.int32)
    buf = C.create_string_buffer(64)
    assert lib.msh_host_tokens_to_text(bad, len(bad), ids.ctypes.data, 2, buf, 64) < 0
    assert lib.msh_host_tokens_to_text(b"", 0, ids.ctypes.data, 2, buf, 64) < 0


def test_shipped_tiny_en_tokenizer_known_answers():
    """tests/golden/tiny_en_tokenizer.bin = the reference's own language-bindings/python/.../assets/tiny-en/tokenizer.bin
    (32,768 entries), the one model artefact its checkout holds.  Without the compiled reference (tests/test_ref_host_diff.py
    diffs against that): the library's decoder against the oracle's on random ids, the layout SURVEY Appendix A.13 describes,
    and the sentence test-assets/beckett.wav says as a known-answer round trip under both encodings."""
    blob = open(os.path.join(os.path.dirname(__file__), "golden", "tiny_en_tokenizer.bin"), "rb").read()
    vocab = host_ref.decode_tokenizer_bin(blob)
    assert len(vocab) == 32768 and vocab[:3] == [b"<unk>", b"<s>", b"</s>"]
    assert vocab[3:259] == [bytes([i]) for i in range(256)]
    assert vocab[32000] == b"<<ST_0>>" and vocab[32767] == b"<<ST_767>>"
    assert host_ref.encode_tokenizer_bin(vocab) == blob
    lib = load_library()
    rng = np.random.default_rng(5)
    for trial in range(200):
        ids = [int(t) for t in rng.integers(0, 32768, int(rng.integers(0, 60))) if t != 31353]   # 31353 is the empty entry
        assert _tok_text(blob, ids) == host_ref.tokens_to_text(vocab, ids)
    ids = np.asarray([5, 31353, 7], np.int32)
    buf = C.create_string_buffer(64)
    assert lib.msh_host_tokens_to_text(blob, len(blob), ids.ctypes.data, 3, buf, 64) < 0            # reference: "Invalid token"
    sentence = b"Ever tried. Ever failed. No matter. Try again. Fail again. Fail better."
    lib.msh_host_text_to_tokens.restype = C.c_int64
    lib.msh_host_text_to_tokens.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int32, C.c_void_p, C.c_uint64]
    marker = "\u2581".encode()
    got = {}
    for bpe in (0, 1):
        out = np.full(128, -1, np.int32)
        n = lib.msh_host_text_to_tokens(blob, len(blob), sentence, len(sentence), marker, bpe, out.ctypes.data, out.size)
        assert n == 19
        ids = out[:n].tolist()
        assert host_ref.tokens_to_text(vocab, [1] + ids + [2]) == sentence
        assert _tok_text(blob, [1] + ids + [2]) == sentence
        got[bpe] = ids
    # the spellings both encoders pick (no word marker is put in front of the text; longest match takes the byte-fallback
    # entry of a lone "E" / ".", the byte-pair replay the learned one) -- the same ids the compiled reference gives
    pieces = [b"E", b"ver", marker + b"tried", b".", marker + b"Ever", marker + b"failed", b".", marker + b"No", marker + b"matter", b".",
              marker + b"Try", marker + b"again", b".", marker + b"Fail", marker + b"again", b".", marker + b"Fail", marker + b"better", b"."]
    for bpe in (0, 1):
        assert [vocab[t] for t in got[bpe]] == pieces
    assert got[0][:4] == [72, 369, 1898, 49] and got[1][:4] == [29923, 369, 1898, 29889]


@pytest.mark.parametrize("in_rate,out_rate,n", [(48000, 16000, 4801), (44100, 16000, 3000), (24000, 16000, 777), (8000, 16000, 500), (11025, 16000, 333), (16000, 16000, 100)])
def test_resampler_bit_exact(in_rate, out_rate, n):
    lib = load_library()
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32)
    want = host_ref.resample_ref(x, in_rate, out_rate)
    m = lib.msh_host_resample(x.ctypes.data, n, in_rate, out_rate, None, 0)
    assert m == want.shape[0]
    out = np.zeros(m, np.float32)
    lib.msh_host_resample(x.ctypes.data, n, in_rate, out_rate, out.ctypes.data, m)
    np.testing.assert_array_equal(out, want)


def _load_skip(options=None):
    opts = {"skip_transcription": "true", "vad_threshold": "0"}
    opts.update(options or {})
    return api.Transcriber("", api.ARCH_BASE, opts)


def test_default_options_without_a_model_load_and_say_what_is_missing():
    """A drop-in caller with the reference's default options (vad_threshold 0.5) and no model (skip_transcription) loads
    -- the reference embeds its VAD, this build reads silero_vad.safetensors -- and the first call that needs the
    network fails with an error code instead of the load (keep the NONE-source path loadable)."""
    t = api.Transcriber("", api.ARCH_BASE, {"skip_transcription": "true"})
    with pytest.raises(api.MoonshineError):
        t.transcribe_without_streaming(np.zeros(16000, np.float32))
    with pytest.raises(api.MoonshineError):
        t.create_stream()
    t.close()


@pytest.mark.parametrize("n", [160000, 159414, 511, 512, 16000 * 31 + 7, 600])
def test_vad_threshold0_segments_match_reference_rules(n):
    """Hop truncation / look-behind / never-split behaviour through the public API
    with no model loaded: text is NULL, audio is the hop-truncated clip."""
    t = _load_skip()
    x = np.random.default_rng(n).standard_normal(n).astype(np.float32) * 0.1
    lines = t.transcribe_without_streaming(x)
    want = host_ref.vad_segments_threshold0(n)
    assert len(lines) == len(want)
    for l, (s, e, complete) in zip(lines, want):
        assert l.text is None and l.is_complete == complete and l.is_new and l.is_updated
        np.testing.assert_array_equal(l.audio_data, x[s:e])
        assert abs(l.start_time - s / 16000) < 1e-6 and abs(l.duration - (e - s) / 16000) < 1e-6
    t.close()


def test_vad_other_sample_rate_and_options():
    t = _load_skip({"vad_max_segment_duration": "5", "return_audio_data": "false"})
    x = np.random.default_rng(1).standard_normal(48000 * 3).astype(np.float32) * 0.1
    lines = t.transcribe_without_streaming(x, sample_rate=48000)
    want = host_ref.vad_segments_threshold0(48000, max_segment_samples=80000)
    assert len(lines) == len(want) == 1
    assert lines[0].audio_data is None  # return_audio_data=false
    assert abs(lines[0].duration - (want[0][1] - want[0][0]) / 16000) < 1e-6
    t.close()


def test_option_parsing_and_error_rules():
    lib = api.lib()
    assert lib.moonshine_get_version() == 30000
    assert lib.moonshine_error_to_string(0) == b"Success"
    assert lib.moonshine_error_to_string(-2) == b"Invalid handle"
    assert lib.moonshine_error_to_string(-3) == b"Invalid argument"
    assert lib.moonshine_error_to_string(-77) == b"Unknown error"
    with pytest.raises(api.MoonshineError) as e:  # unknown option names fail the load (reference c-api.cpp:193-196)
        _load_skip({"no_such_option": "1"})
    assert e.value.code == -1
    with pytest.raises(api.MoonshineError):  # bools accept only true/false/1/0
        _load_skip({"return_audio_data": "yes"})
    with pytest.raises(api.MoonshineError):  # vad_threshold > 0 without Silero weights fails the LOAD (no silent fallback)
        api.Transcriber("", api.ARCH_BASE, {"skip_transcription": "true", "vad_threshold": "0.5"}).transcribe_without_streaming(np.zeros(1000, np.float32))
    t = _load_skip({"VAD_HOP_SIZE": "256", "Log_Api_Calls": "false"})  # names are case-insensitive
    assert len(t.transcribe_without_streaming(np.zeros(1024, np.float32))) == 1
    out = C.POINTER(api.TranscriptC)()
    assert lib.moonshine_transcribe_without_streaming(12345, None, 0, 16000, 0, C.byref(out)) == -2
    assert lib.moonshine_transcribe_without_streaming(-1, None, 0, 16000, 0, C.byref(out)) == -2
    assert lib.moonshine_create_stream(9999, 0) == -2
    assert lib.moonshine_transcriber_set_keyterms(t.handle, b"") == 0
    # no model at all (skip_transcription): nothing to tokenize against, accepted (reference transcriber.cpp:266-277)
    assert lib.moonshine_transcriber_set_keyterms(t.handle, b"Kubernetes, etcd") == 0
    assert lib.moonshine_transcriber_set_keyterms(4242, b"x") == -2
    assert lib.moonshine_transcriber_set_context(t.handle, b"Madame Defarge knits", 0) == 0   # no model: accepted, no terms
    assert lib.moonshine_transcriber_set_context(4242, b"x", 0) == -2
    assert lib.moonshine_load_transcriber_from_memory(None, 0, None, 0, None, 0, None, 0, 1, None, 0, 30000) == -3
    t.close()
    t.close()  # double free is harmless
    lib.moonshine_free_transcriber(4242)


def test_loading_a_model_without_a_gpu_fails_loudly(tmp_path):
    """There is no CPU fallback: on a box without an MI355X a real model load returns an error."""
    load = load_library()
    if load.msh_device_count() > 0:
        pytest.skip("GPU present")
    (tmp_path / "tokenizer.bin").write_bytes(host_ref.encode_tokenizer_bin(host_ref.synthetic_vocab(300)))
    (tmp_path / "model.safetensors").write_bytes(b"\x02\x00\x00\x00\x00\x00\x00\x00{}")
    with pytest.raises(api.MoonshineError) as e:
        api.Transcriber(str(tmp_path), api.ARCH_BASE, {"vad_threshold": "0"})
    assert e.value.code == -1
    with pytest.raises(api.MoonshineError):  # .ort-only directory: clear refusal
        (tmp_path / "encoder_model.ort").write_bytes(b"x")
        os.remove(tmp_path / "model.safetensors")
        api.Transcriber(str(tmp_path), api.ARCH_BASE, {"vad_threshold": "0"})
    with pytest.raises(api.MoonshineError):
        api.Transcriber(str(tmp_path / "missing"), api.ARCH_BASE, {"vad_threshold": "0"})
    with pytest.raises(api.MoonshineError):  # streaming architectures are a later row
        api.Transcriber(str(tmp_path), 5, {"vad_threshold": "0"})


def test_stream_flag_semantics_without_model():
    """Line bookkeeping of the stream API (reference transcriber.cpp:775-891, 1648-1726 and the guarantees of
    moonshine-c-api.h:159-200): lines are only added, ids are stable, only new audio >= the interval (or
    FORCE_UPDATE) triggers an update, stop marks the open line complete."""
    t = _load_skip({"transcription_interval": "0.5"})
    s = t.create_stream()
    with pytest.raises(api.MoonshineError):
        t.add_audio(s, np.zeros(100, np.float32))  # not started
    t.start_stream(s)
    x = np.random.default_rng(3).standard_normal(16000 * 2).astype(np.float32) * 0.1
    t.add_audio(s, x[:4000])
    assert t.transcribe_stream(s) == []  # 0.25 s < interval: cached (empty) transcript
    lines = t.transcribe_stream(s, api.FLAG_FORCE_UPDATE)
    assert len(lines) == 1 and lines[0].is_new and lines[0].is_updated and not lines[0].is_complete
    first_id = lines[0].line_id
    np.testing.assert_array_equal(lines[0].audio_data, x[:3584])  # 7 whole hops
    t.add_audio(s, x[4000:16000])
    lines = t.transcribe_stream(s)
    assert len(lines) == 1 and lines[0].line_id == first_id and not lines[0].is_new and lines[0].is_updated
    assert lines[0].audio_data.shape[0] == (16000 // 512) * 512
    lines = t.transcribe_stream(s)  # nothing new: flags cleared
    assert not lines[0].is_updated and not lines[0].is_new
    t.stop_stream(s)
    lines = t.transcribe_stream(s)
    assert lines[0].is_complete and lines[0].line_id == first_id
    t.start_stream(s)  # restarting clears the transcript
    assert t.transcribe_stream(s) == []
    t.free_stream(s)
    with pytest.raises(api.MoonshineError):
        t.transcribe_stream(s)
    t.close()


# ---- known-answer cases of the reference's own tests, restated (they need the reference's test assets, which
#      exist in the build container only; skipped elsewhere) ----
REF_ASSETS = "/root/reference/test-assets"


def _read_wav(path):
    import wave

    with wave.open(path) as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        return (pcm.astype(np.float32) / np.float32(32768.0)), w.getframerate()


def test_reference_sanitize_known_answers():
    """core/transcriber-test.cpp:510-533 (test-invalid-utf8 / test-valid-utf8)."""
    bad = bytes([0xA3, 0x0A, 0xF5, 0x78])
    for fn in (host_ref.sanitize_text, _cxx_sanitize):
        out = fn(bad)
        assert out[0] < 0x80 and len(out) == len(bad)
        assert fn(b"Hello, world!") == b"Hello, world!"


def _cxx_sanitize(b: bytes) -> bytes:
    lib = load_library()
    out = C.create_string_buffer(len(b) + 8)
    n = lib.msh_host_sanitize_utf8(b, len(b), out, len(b) + 8)
    assert n >= 0
    return out.raw[:n]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ASSETS, "beckett.wav")), reason="reference test assets not present")
def test_reference_vad_threshold0_known_answer():
    """core/voice-activity-detector-test.cpp:122-163 (vad-threshold-0 on beckett.wav, hop 256): one complete segment
    that loses at most one hop, starting at ~0 and ending at the clip's duration."""
    audio, rate = _read_wav(os.path.join(REF_ASSETS, "beckett.wav"))
    hop = 256
    t = _load_skip({"vad_hop_size": str(hop)})
    lines = t.transcribe_without_streaming(audio, sample_rate=rate)
    assert len(lines) == 1 and lines[0].is_complete
    n = lines[0].audio_data.shape[0]
    assert audio.shape[0] - hop <= n <= audio.shape[0]
    eps = hop / rate
    assert 0 <= lines[0].start_time < eps
    assert abs(lines[0].start_time + lines[0].duration - audio.shape[0] / rate) <= eps
    np.testing.assert_array_equal(lines[0].audio_data, audio[:n])
    segs = host_ref.vad_segments_threshold0(audio.shape[0], hop=hop)
    assert len(segs) == 1 and segs[0][1] - segs[0][0] == n and segs[0][2]
    t.close()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ASSETS, "beckett.wav")), reason="reference test assets not present")
@pytest.mark.parametrize("out_rate", [16000, 96000, 8000])
def test_reference_resampler_known_answer(out_rate):
    """core/resampler-test.cpp:11-36: max, min and mean survive resampling (0.5 % / 0.5 % / 0.1 %) -- on the
    reference's 16 kHz assets re-labelled 48 kHz so that both directions are exercised."""
    audio, _ = _read_wav(os.path.join(REF_ASSETS, "beckett.wav"))
    lib = load_library()
    n = lib.msh_host_resample(audio.ctypes.data, audio.shape[0], 48000.0, float(out_rate), None, 0)
    out = np.zeros(n, np.float32)
    assert lib.msh_host_resample(audio.ctypes.data, audio.shape[0], 48000.0, float(out_rate), out.ctypes.data, n) == n
    np.testing.assert_array_equal(out, host_ref.resample_ref(audio, 48000, out_rate))
    if out_rate > 48000:  # (the re-labelled clip is not band-limited for the "48 kHz" rate, so the box filter of the
        #                     down direction does flatten its peaks; the reference's asset was recorded at 48 kHz)
        assert abs(out.max() - audio.max()) <= 0.005 * abs(audio.max())
        assert abs(out.min() - audio.min()) <= 0.005 * abs(audio.min())
    assert abs(out.mean() - audio.mean()) <= 1e-3


def test_c_program_links_against_the_public_header_and_fails_loudly_without_a_gpu(tmp_path):
    """tests/c/word_timestamps_flow.c (the reference's word-timestamp test flow) builds with plain gcc against
    include/moonshine-c-api.h + libmoonshine.so; on a box without a GPU the load is refused with an error code -- there
    is no CPU fallback behind the C API."""
    import torch

    from moonshine_amd.build import LIB

    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "wt_flow")
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I", INCLUDE, os.path.join(here, "c", "word_timestamps_flow.c"), "-o", exe, LIB,
                    f"-Wl,-rpath,{os.path.dirname(LIB)}"], check=True)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by tests/test_gpu_capi.py")
    r = subprocess.run([exe, str(tmp_path), str(tmp_path / "none.wav")], capture_output=True, text=True)
    assert r.returncode == 1 and "REQUIRE failed" in r.stderr and "handle >= 0" in r.stderr


def _wav_bytes(samples_i16: np.ndarray, rate: int, channels: int = 1, fmt: int = 1, bits: int = 16, extra_chunks=(), fmt_extra: bytes = b"",
               claim_data_bytes: int | None = None) -> bytes:
    import struct

    data = samples_i16.astype("<i2").tobytes() if bits == 16 else samples_i16.astype("u1").tobytes()
    fmt_chunk = struct.pack("<HHIIHH", fmt, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits) + fmt_extra
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk
    for cid, payload in extra_chunks:
        body += cid + struct.pack("<I", len(payload)) + payload
    body += b"data" + struct.pack("<I", len(data) if claim_data_bytes is None else claim_data_bytes) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_reader_and_writer(tmp_path):
    """msh_host_load_wav / msh_host_save_wav against the rules of the reference's load_wav_data / save_wav_data
    (core/moonshine-utils/debug-utils.cpp:52-250): 16-bit PCM only, sample / 32768, channels left interleaved, chunks
    before `data` skipped, a data size larger than the file clamped; and against Python's own `wave` module."""
    import wave

    lib = load_library()
    lib.msh_host_load_wav.restype = C.c_int64
    lib.msh_host_load_wav.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.msh_host_save_wav.restype = C.c_int32
    lib.msh_host_save_wav.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int32]

    def load(path):
        rate = C.c_int32(0)
        n = lib.msh_host_load_wav(str(path).encode(), None, 0, C.byref(rate))
        if n < 0:
            return None, 0
        out = np.zeros(n, np.float32)
        assert lib.msh_host_load_wav(str(path).encode(), out.ctypes.data, n, C.byref(rate)) == n
        return out, rate.value

    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32768, 4001, dtype=np.int64).astype(np.int16)
    pcm[:3] = [-32768, 32767, 0]
    # a file written by Python's wave module
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(pcm.tobytes())
    got, rate = load(p)
    assert rate == 22050
    np.testing.assert_array_equal(got, pcm.astype(np.float32) / np.float32(32768.0))
    # LIST chunk before the data, extended fmt chunk, stereo (left interleaved)
    p2 = tmp_path / "b.wav"
    p2.write_bytes(_wav_bytes(pcm[:4000], 48000, channels=2, extra_chunks=[(b"LIST", b"x" * 26)], fmt_extra=b"\x00\x00"))
    got, rate = load(p2)
    assert rate == 48000 and got.shape[0] == 4000
    np.testing.assert_array_equal(got, pcm[:4000].astype(np.float32) / np.float32(32768.0))
    # the header claims more data than the file holds: clamped to what is there
    p3 = tmp_path / "c.wav"
    p3.write_bytes(_wav_bytes(pcm[:100], 16000, claim_data_bytes=1 << 30))
    got, _ = load(p3)
    assert got.shape[0] == 100
    # refused: 8-bit, float format, not RIFF, no data chunk, empty data, missing file
    bad = {"d.wav": _wav_bytes(pcm[:10], 16000, bits=8), "e.wav": _wav_bytes(pcm[:10], 16000, fmt=3), "f.wav": b"RIFX" + b"\0" * 60,
           "g.wav": _wav_bytes(pcm[:10], 16000)[:36], "h.wav": _wav_bytes(pcm[:0], 16000)}
    for name, blob in bad.items():
        (tmp_path / name).write_bytes(blob)
        assert load(tmp_path / name)[0] is None, name
    assert load(tmp_path / "missing.wav")[0] is None
    # writer: clamp to [-1, 1], truncate towards zero after x 32768, readable by `wave`, round-trips through the reader
    x = np.asarray([0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -2.0, 0.99999, 1e-6, -3.0517578125e-05], np.float32)
    p4 = tmp_path / "out.wav"
    assert lib.msh_host_save_wav(str(p4).encode(), x.ctypes.data, len(x), 16000) == 0
    with wave.open(str(p4), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, len(x))
        raw = np.frombuffer(w.readframes(len(x)), "<i2")
    want = np.clip(x.astype(np.float32) * np.float32(32768.0), -32768.0, 32767.0).astype(np.int16)  # astype truncates towards zero
    np.testing.assert_array_equal(raw, want)
    got, rate = load(p4)
    np.testing.assert_array_equal(got, want.astype(np.float32) / np.float32(32768.0))


def test_reference_python_binding_binds_every_symbol_unedited():
    """The reference's own ctypes binding resolves EVERY function of moonshine-c-api.h when it loads the library
    (language-bindings/python/src/moonshine_voice/moonshine_api.py:860-1121), including the TTS / G2P / embedding /
    catalog calls that are out of scope here and exported as MOONSHINE_ERROR_UNKNOWN stubs.  Loading our libmoonshine.so
    through that unedited module must work.  Needs /root/reference (build container only)."""
    src = "/root/reference/language-bindings/python/src"
    if not os.path.isdir(src):
        pytest.skip("reference checkout not present")
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "import moonshine_voice.moonshine_api as m\n"
        "l = m._MoonshineLib()._lib\n"
        "assert l.moonshine_get_version() == 30000\n"
        "out = ctypes.c_void_p()\n"
        "assert l.moonshine_get_stt_catalog(ctypes.byref(out)) == -1 and not out.value\n"
        "assert l.moonshine_create_tts_synthesizer_from_files(b'en', None, 0, None, 0, 30000) == -1\n"
        "print('bound', l._name)\n" % src
    )
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(LIB_PATH) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "bound libmoonshine.so" in r.stdout


def test_effective_cpus_follows_affinity_and_cgroup_quota():
    """Default host thread counts come from msh_host_effective_cpus: the affinity mask, cut down to the cgroup CPU quota
    (a container that sees 256 cores but is granted 16 CPUs must not start 128 VAD threads: measured, that halves the rate)."""
    import math
    import os

    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    lib = api.lib()
    lib.msh_host_effective_cpus.restype = __import__("ctypes").c_int32
    assert lib.msh_host_effective_cpus() == max(1, n)
