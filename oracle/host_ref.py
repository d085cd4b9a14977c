"""Restatement of the reference's host-side byte/integer work on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pure Python / numpy; each function
cites the reference lines it follows.  Bit-exact parity is required of the C++
host code against these (tests/test_host_parity.py).
"""
from __future__ import annotations

import math

import numpy as np

SPACE = "\u2581".encode("utf-8")  # sentencepiece word-boundary marker


# --- tokenizer.bin ---------------------------------------------------------
from moonshine_amd.synth import encode_tokenizer_bin, synthetic_vocab, write_synthetic_tokenizer  # noqa: E402,F401


def decode_tokenizer_bin(blob: bytes) -> list[bytes]:
    """reference core/bin-tokenizer/bin-tokenizer.cpp:46-66."""
    toks: list[bytes] = []
    i = 0
    while i < len(blob):
        first = blob[i]
        i += 1
        if first == 0:
            toks.append(b"")
            continue
        if first < 128:
            n = first
        else:
            second = blob[i]
            i += 1
            n = second * 128 + first - 128
        if i + n > len(blob):
            raise ValueError("truncated tokenizer.bin")
        toks.append(blob[i : i + n])
        i += n
    if not toks:
        raise ValueError("no tokens")
    return toks


def tokens_to_text(vocab: list[bytes], tokens) -> bytes:
    """reference core/bin-tokenizer/bin-tokenizer.cpp:406-426: concatenate token
    bytes, skip ``<...>`` specials (len > 2), replace the space marker, trim."""
    out = bytearray()
    for t in tokens:
        b = vocab[int(t)]
        if len(b) == 0:
            raise ValueError(f"Invalid token {t}")
        if len(b) > 2 and b[:1] == b"<" and b[-1:] == b">":
            continue
        out += b
    res = bytes(out).replace(SPACE, b" ")
    # trim(): reference core/moonshine-utils/string-utils.cpp:21-29, default
    # whitespace set " \t" (string-utils.h:12)
    return res.strip(b" \t")


def sanitize_text(text: bytes) -> bytes:
    """reference core/transcriber.cpp:1489-1543: replace each byte that does not
    start a structurally valid UTF-8 sequence by '?'."""
    out = bytearray()
    i = 0
    n = len(text)

    def cont(j):
        return (text[j] & 0xC0) == 0x80

    while i < n:
        c = text[i]
        rem = n - i
        if c < 0x80:
            out.append(c)
            i += 1
        elif (c & 0xE0) == 0xC0:
            if rem < 2 or not cont(i + 1):
                out.append(0x3F)
                i += 1
            else:
                out += text[i : i + 2]
                i += 2
        elif (c & 0xF0) == 0xE0:
            if rem < 3 or not cont(i + 1) or not cont(i + 2):
                out.append(0x3F)
                i += 1
            else:
                out += text[i : i + 3]
                i += 3
        elif (c & 0xF8) == 0xF0:
            if rem < 4 or not cont(i + 1) or not cont(i + 2) or not cont(i + 3):
                out.append(0x3F)
                i += 1
            else:
                out += text[i : i + 4]
                i += 4
        else:
            out.append(0x3F)
            i += 1
    return bytes(out)


# --- VAD segmentation with the Silero model bypassed -----------------------
def vad_segments_threshold0(
    n_samples: int,
    hop: int = 512,
    look_behind: int = 8192,
    max_segment_samples: int = 240000,
) -> list[tuple[int, int, bool]]:
    """Segment boundaries (start_sample, end_sample, is_complete) produced by the
    reference VAD when ``vad_threshold == 0`` (no Silero call), for one
    ``transcribe_without_streaming`` call on 16 kHz audio.

    Follows reference core/voice-activity-detector.cpp:69-96 (whole hops only,
    remainder dropped), :125-199 (threshold 0 => probability 1, multiplied by the
    max-length fade factor (len - fade)/fade once len > fade; voice iff p > 0)
    and :62-67 (stop() closes the open segment).  The first voiced hop starts
    the segment from the look-behind buffer (:171-179)."""
    segs: list[tuple[int, int, bool]] = []
    fade = (max_segment_samples * 2) // 3
    processed = 0
    prev_voice = False
    cur_start = 0
    cur_len = 0
    n_hops = n_samples // hop
    for _ in range(n_hops):
        processed += hop
        p = 1.0
        if max_segment_samples and cur_len > fade:
            p = p * (np.float32(cur_len - fade) / np.float32(fade))
        voice = p > 0.0
        if voice and not prev_voice:
            lb = min(look_behind, processed)
            cur_len = lb
            cur_start = processed - lb
        elif (not voice) and prev_voice:
            cur_len += hop
            segs.append((cur_start, cur_start + cur_len, True))
            cur_len = 0
        elif voice and prev_voice:
            cur_len += hop
        prev_voice = voice
    if prev_voice:
        segs.append((cur_start, cur_start + cur_len, True))
    return segs


def max_decode_len(n_samples: int, max_tokens_per_second: float = 6.5) -> int:
    """reference core/moonshine-model.cpp:347-349 (float32 arithmetic)."""
    dur = np.float32(n_samples) / np.float32(16000.0)
    return int(math.ceil(float(np.float32(dur * np.float32(max_tokens_per_second)))))


def resample_ref(audio: np.ndarray, in_rate: float, out_rate: float) -> np.ndarray:
    """reference core/resampler.cpp:5-86 in float32: identity at equal rates; box average over
    [floor(i*r), floor((i+1)*r)] (inclusive, clamped) when decimating; linear interpolation with the last
    sample held when interpolating.  Output length = trunc(n * out_rate / in_rate) in float arithmetic."""
    audio = np.asarray(audio, np.float32)
    if in_rate == out_rate:
        return audio.copy()
    n_in = audio.shape[0]
    n_out = int(np.float32(np.float32(n_in) * np.float32(out_rate)) / np.float32(in_rate))
    ratio = np.float32(in_rate) / np.float32(out_rate)
    out = np.zeros(n_out, np.float32)
    if in_rate > out_rate:
        for i in range(n_out):
            lo = int(np.float32(i) * ratio)
            hi = min(int(np.float32(i + 1) * ratio), n_in - 1)
            acc = np.float32(0.0)
            for j in range(lo, hi + 1):
                acc = np.float32(acc + audio[j])
            cnt = hi - lo + 1
            out[i] = acc / np.float32(cnt) if cnt > 0 else 0.0
    else:
        for i in range(n_out):
            pos = np.float32(i) * ratio
            idx = int(pos)
            frac = np.float32(pos - np.float32(idx))
            if idx >= n_in - 1:
                out[i] = audio[n_in - 1]
            else:
                out[i] = np.float32(audio[idx] + frac * np.float32(audio[idx + 1] - audio[idx]))
    return out
