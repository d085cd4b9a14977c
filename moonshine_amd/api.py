"""ctypes binding of the public C API (include/moonshine-c-api.h).

The shape follows the reference's own Python binding (reference
language-bindings/python/src/moonshine_voice/moonshine_api.py:60-160 ctypes structs, :864-1121
prototypes, transcriber.py:191-230 the Transcriber wrapper) so tests written against the reference
binding read the same here.  Only the transcriber subset exists in this library.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .hip_api import LIB_PATH

MOONSHINE_HEADER_VERSION = 30000
ARCH_TINY, ARCH_BASE = 0, 1
ARCH_TINY_STREAMING, ARCH_BASE_STREAMING, ARCH_SMALL_STREAMING, ARCH_MEDIUM_STREAMING = 2, 3, 4, 5
FLAG_FORCE_UPDATE = 1
MOONSHINE_ERROR_NONE, MOONSHINE_ERROR_UNKNOWN, MOONSHINE_ERROR_INVALID_HANDLE, MOONSHINE_ERROR_INVALID_ARGUMENT = 0, -1, -2, -3


class TranscriptWordC(C.Structure):
    _fields_ = [("text", C.POINTER(C.c_char)), ("start", C.c_float), ("end", C.c_float), ("confidence", C.c_float)]


class SpeakerSpanC(C.Structure):
    _fields_ = [("start_time", C.c_float), ("duration", C.c_float), ("speaker_id", C.c_uint64), ("speaker_index", C.c_uint32),
                ("start_char", C.c_uint64), ("end_char", C.c_uint64)]


class TranscriptLineC(C.Structure):
    _fields_ = [
        ("text", C.c_char_p), ("audio_data", C.POINTER(C.c_float)), ("audio_data_count", C.c_size_t), ("start_time", C.c_float),
        ("duration", C.c_float), ("id", C.c_uint64), ("is_complete", C.c_int8), ("is_updated", C.c_int8), ("is_new", C.c_int8),
        ("has_text_changed", C.c_int8), ("have_speakers_changed", C.c_int8), ("speaker_spans", C.POINTER(SpeakerSpanC)),
        ("speaker_span_count", C.c_uint64), ("last_transcription_latency_ms", C.c_uint32), ("words", C.POINTER(TranscriptWordC)),
        ("word_count", C.c_uint64),
    ]


class TranscriptC(C.Structure):
    _fields_ = [("lines", C.POINTER(TranscriptLineC)), ("line_count", C.c_uint64)]


class OptionC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("value", C.c_char_p)]


# sizes the reference binding pins (moonshine_api.py:137-147)
assert (C.sizeof(TranscriptWordC), C.sizeof(SpeakerSpanC), C.sizeof(TranscriptLineC), C.sizeof(TranscriptC)) == (24, 40, 88, 16)

C_API_SYMBOLS = [
    "moonshine_get_version", "moonshine_error_to_string", "moonshine_free_buffer", "moonshine_transcript_to_string",
    "moonshine_transcriber_set_keyterms", "moonshine_transcriber_set_context", "moonshine_load_transcriber_from_files",
    "moonshine_load_transcriber_from_memory", "moonshine_load_transcriber_from_memory_files", "moonshine_free_transcriber",
    "moonshine_transcribe_without_streaming", "moonshine_create_stream", "moonshine_free_stream", "moonshine_start_stream",
    "moonshine_stop_stream", "moonshine_transcribe_add_audio_to_stream", "moonshine_transcribe_stream",
    "moonshine_transcribe_batch_without_streaming", "moonshine_transcribe_batch_without_streaming_pcm16",
]

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    l = C.CDLL(LIB_PATH)
    i32, u32, u64, P = C.c_int32, C.c_uint32, C.c_uint64, C.POINTER
    l.moonshine_get_version.restype = i32
    l.moonshine_error_to_string.restype = C.c_char_p
    l.moonshine_error_to_string.argtypes = [i32]
    l.moonshine_transcript_to_string.restype = C.c_char_p
    l.moonshine_transcript_to_string.argtypes = [P(TranscriptC)]
    l.moonshine_load_transcriber_from_files.restype = i32
    l.moonshine_load_transcriber_from_files.argtypes = [C.c_char_p, u32, P(OptionC), u64, i32]
    l.moonshine_load_transcriber_from_memory_files.restype = i32
    l.moonshine_load_transcriber_from_memory_files.argtypes = [P(C.c_char_p), P(C.c_void_p), P(u64), u64, u32, P(OptionC), u64, i32]
    l.moonshine_load_transcriber_from_memory.restype = i32
    l.moonshine_free_transcriber.argtypes = [i32]
    l.moonshine_free_transcriber.restype = None
    l.moonshine_transcribe_without_streaming.restype = i32
    l.moonshine_transcribe_without_streaming.argtypes = [i32, P(C.c_float), u64, i32, u32, P(P(TranscriptC))]
    l.moonshine_transcribe_batch_without_streaming.restype = i32
    l.moonshine_transcribe_batch_without_streaming.argtypes = [i32, P(P(C.c_float)), P(u64), u64, i32, u32, P(P(TranscriptC))]
    l.moonshine_transcribe_batch_without_streaming_pcm16.restype = i32
    l.moonshine_transcribe_batch_without_streaming_pcm16.argtypes = [i32, P(P(C.c_int16)), P(u64), u64, i32, u32, P(P(TranscriptC))]
    for name in ("moonshine_create_stream",):
        getattr(l, name).restype = i32
        getattr(l, name).argtypes = [i32, u32]
    for name in ("moonshine_free_stream", "moonshine_start_stream", "moonshine_stop_stream"):
        getattr(l, name).restype = i32
        getattr(l, name).argtypes = [i32, i32]
    l.moonshine_transcribe_add_audio_to_stream.restype = i32
    l.moonshine_transcribe_add_audio_to_stream.argtypes = [i32, i32, P(C.c_float), u64, i32, u32]
    l.moonshine_transcribe_stream.restype = i32
    l.moonshine_transcribe_stream.argtypes = [i32, i32, u32, P(P(TranscriptC))]
    l.moonshine_transcriber_set_keyterms.restype = i32
    l.moonshine_transcriber_set_keyterms.argtypes = [i32, C.c_char_p]
    l.moonshine_transcriber_set_context.restype = i32
    l.moonshine_transcriber_set_context.argtypes = [i32, C.c_char_p, i32]
    _lib = l
    return l


class MoonshineError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(f"moonshine error {code}: {lib().moonshine_error_to_string(code).decode()}")
        self.code = code


@dataclass
class TranscriptLine:
    text: str | None
    text_bytes: bytes | None
    start_time: float
    duration: float
    line_id: int
    is_complete: bool
    is_updated: bool
    is_new: bool
    has_text_changed: bool
    audio_data: np.ndarray | None
    last_transcription_latency_ms: int
    words: list = None  # [(text bytes, start s, end s, confidence)] with the word_timestamps option, else []


def _options(options: dict | None):
    items = list((options or {}).items())
    arr = (OptionC * max(len(items), 1))()
    for i, (k, v) in enumerate(items):
        arr[i].name = str(k).encode()
        arr[i].value = str(v).encode()
    return arr, len(items)


def _parse(tp) -> list[TranscriptLine]:
    out = []
    t = tp.contents
    for i in range(t.line_count):
        l = t.lines[i]
        audio = None
        if l.audio_data and l.audio_data_count:
            audio = np.ctypeslib.as_array(l.audio_data, shape=(l.audio_data_count,)).copy()
        out.append(TranscriptLine(
            text=None if l.text is None else l.text.decode("utf-8", errors="replace"), text_bytes=l.text, start_time=l.start_time,
            duration=l.duration, line_id=l.id, is_complete=bool(l.is_complete), is_updated=bool(l.is_updated), is_new=bool(l.is_new),
            has_text_changed=bool(l.has_text_changed), audio_data=audio, last_transcription_latency_ms=l.last_transcription_latency_ms,
            words=[(C.cast(l.words[k].text, C.c_char_p).value, float(l.words[k].start), float(l.words[k].end), float(l.words[k].confidence)) for k in range(l.word_count)]))
    return out


class Transcriber:
    """Owns one transcriber handle.  Mirrors the reference's Python Transcriber surface."""

    def __init__(self, model_path: str, model_arch: int = ARCH_BASE, options: dict | None = None):
        arr, n = _options(options)
        h = lib().moonshine_load_transcriber_from_files(model_path.encode(), model_arch, arr, n, MOONSHINE_HEADER_VERSION)
        if h < 0:
            raise MoonshineError(h)
        self.handle = h

    @classmethod
    def from_memory_files(cls, files: dict[str, bytes | None], model_arch: int = ARCH_BASE, options: dict | None = None):
        names = (C.c_char_p * len(files))()
        mem = (C.c_void_p * len(files))()
        sizes = (C.c_uint64 * len(files))()
        keep = []
        for i, (k, v) in enumerate(files.items()):
            names[i] = k.encode()
            if v is not None:
                buf = C.create_string_buffer(v, len(v))
                keep.append(buf)
                mem[i] = C.cast(buf, C.c_void_p)
                sizes[i] = len(v)
        arr, n = _options(options)
        h = lib().moonshine_load_transcriber_from_memory_files(names, mem, sizes, len(files), model_arch, arr, n, MOONSHINE_HEADER_VERSION)
        if h < 0:
            raise MoonshineError(h)
        self = cls.__new__(cls)
        self.handle = h
        return self

    def close(self):
        if getattr(self, "handle", -1) >= 0:
            lib().moonshine_free_transcriber(self.handle)
            self.handle = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transcribe_without_streaming(self, audio, sample_rate: int = 16000, flags: int = 0) -> list[TranscriptLine]:
        a = np.ascontiguousarray(audio, dtype=np.float32)
        out = C.POINTER(TranscriptC)()
        rc = lib().moonshine_transcribe_without_streaming(self.handle, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], sample_rate, flags, C.byref(out))
        if rc != 0:
            raise MoonshineError(rc)
        return _parse(out)

    def transcribe_batch_without_streaming(self, clips, sample_rate: int = 16000, flags: int = 0) -> list[list[TranscriptLine]]:
        if len(clips) > 0 and all(np.asarray(c).dtype == np.int16 for c in clips):   # 16-bit PCM: the additive _pcm16 entry point
            arrs = [np.ascontiguousarray(c, dtype=np.int16) for c in clips]
            n = len(arrs)
            ptrs = (C.POINTER(C.c_int16) * n)(*[a.ctypes.data_as(C.POINTER(C.c_int16)) for a in arrs])
            lens = (C.c_uint64 * n)(*[a.shape[0] for a in arrs])
            outs = (C.POINTER(TranscriptC) * n)()
            rc = lib().moonshine_transcribe_batch_without_streaming_pcm16(self.handle, ptrs, lens, n, sample_rate, flags, outs)
            if rc != 0:
                raise MoonshineError(rc)
            return [_parse(outs[i]) for i in range(n)]
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        n = len(arrs)
        ptrs = (C.POINTER(C.c_float) * n)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
        lens = (C.c_uint64 * n)(*[a.shape[0] for a in arrs])
        outs = (C.POINTER(TranscriptC) * n)()
        rc = lib().moonshine_transcribe_batch_without_streaming(self.handle, ptrs, lens, n, sample_rate, flags, outs)
        if rc != 0:
            raise MoonshineError(rc)
        return [_parse(outs[i]) for i in range(n)]

    # streams
    def create_stream(self) -> int:
        s = lib().moonshine_create_stream(self.handle, 0)
        if s < 0:
            raise MoonshineError(s)
        return s

    def _call(self, fn, *args):
        rc = fn(self.handle, *args)
        if rc < 0:
            raise MoonshineError(rc)
        return rc

    def free_stream(self, s):
        self._call(lib().moonshine_free_stream, s)

    def start_stream(self, s):
        self._call(lib().moonshine_start_stream, s)

    def stop_stream(self, s):
        self._call(lib().moonshine_stop_stream, s)

    def add_audio(self, s, audio, sample_rate: int = 16000):
        a = np.ascontiguousarray(audio, dtype=np.float32)
        self._call(lib().moonshine_transcribe_add_audio_to_stream, s, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], sample_rate, 0)

    def transcribe_stream(self, s, flags: int = 0) -> list[TranscriptLine]:
        out = C.POINTER(TranscriptC)()
        self._call(lib().moonshine_transcribe_stream, s, flags, C.byref(out))
        return _parse(out)
