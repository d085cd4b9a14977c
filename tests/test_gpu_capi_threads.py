"""Thread-safety of the C API on the GPU (reference contract core/moonshine-c-api.h:64-67: the API may be called from
several threads; work on one transcriber is serialised).  ctypes releases the GIL for the duration of a call, so these
threads really run inside libmoonshine.so at the same time.  Results must equal the ones a single thread gets."""
import threading

import numpy as np
import pytest

from moonshine_amd import api
from moonshine_amd.synth import ARCHS, make_audio, write_model_dir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("tiny_model_threads"))
    write_model_dir(d, ARCHS["tiny"], seed=3)
    return d


def _run_threads(fns):
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:  # noqa: BLE001 -- reported below, in the test's thread
            errs.append(e)

    ts = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a thread is stuck inside the library"
    if errs:
        raise errs[0]


def test_transcribers_in_parallel_threads(tiny_dir):
    """Three transcribers (own engines, one GPU), one thread each, many calls: texts equal the single-thread ones."""
    clips = [make_audio(100 + i, 16000 + 3000 * i) for i in range(6)]
    ref_t = api.Transcriber(tiny_dir, api.ARCH_TINY, {"vad_threshold": "0"})
    want = [[l.text_bytes for l in ref_t.transcribe_without_streaming(c)] for c in clips]
    ref_t.close()
    trs = [api.Transcriber(tiny_dir, api.ARCH_TINY, {"vad_threshold": "0"}) for _ in range(3)]
    got = [[None] * len(clips) for _ in trs]

    def work(k):
        def f():
            for rep in range(4):
                for i, c in enumerate(clips):
                    got[k][i] = [l.text_bytes for l in trs[k].transcribe_without_streaming(c)]
                    assert got[k][i] == want[i], (k, rep, i)
        return f

    _run_threads([work(k) for k in range(len(trs))])
    for t in trs:
        t.close()


def test_streams_of_one_transcriber_in_parallel_threads(tiny_dir):
    """One transcriber, one stream per thread (its own transcript), a producer thread per stream feeding audio while the
    consumer transcribes: the final lines equal a single-threaded run of the same stream."""
    tr = api.Transcriber(tiny_dir, api.ARCH_TINY, {"vad_threshold": "0"})
    audios = [make_audio(200 + i, 40000 + 4000 * i) for i in range(3)]

    def single(a):
        s = tr.create_stream()
        tr.start_stream(s)
        tr.add_audio(s, a)
        tr.transcribe_stream(s, api.FLAG_FORCE_UPDATE)   # the update consumes the audio (an update after stop would not)
        tr.stop_stream(s)
        lines = tr.transcribe_stream(s)
        out = [(l.text_bytes, l.is_complete) for l in lines]
        tr.free_stream(s)
        return out

    want = [single(a) for a in audios]
    got = [None] * len(audios)

    def stream_worker(k):
        def f():
            a = audios[k]
            s = tr.create_stream()
            tr.start_stream(s)
            fed = threading.Event()

            def producer():
                for off in range(0, len(a), 4000):   # an audio callback: small pieces, never blocks for long
                    tr.add_audio(s, a[off:off + 4000])
                fed.set()

            p = threading.Thread(target=producer)
            p.start()
            while not fed.is_set():
                tr.transcribe_stream(s)           # updates while audio is still arriving
            p.join()
            tr.transcribe_stream(s, api.FLAG_FORCE_UPDATE)   # whatever arrived since the last update
            tr.stop_stream(s)
            lines = tr.transcribe_stream(s)
            got[k] = [(l.text_bytes, l.is_complete) for l in lines]
            tr.free_stream(s)
        return f

    _run_threads([stream_worker(k) for k in range(len(audios))])
    assert got == want
    tr.close()


def test_free_transcriber_while_another_thread_is_inside_a_call(tiny_dir):
    """moonshine_free_transcriber racing a batch call: no crash, the call either completes or reports an error, and the
    handle is invalid afterwards (the reference keeps the object alive for the call through its map lock; here a
    shared_ptr does, c_api.cpp lookup()).  The transcripts of a freed transcriber are gone with it (moonshine-c-api.h:
    valid until the next call or free), so the racing call's outputs are not read here -- only its status."""
    import time

    C = api.C
    clips = [np.ascontiguousarray(make_audio(300 + i, 24000), dtype=np.float32) for i in range(48)]
    n = len(clips)
    ptrs = (C.POINTER(C.c_float) * n)(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in clips])
    lens = (C.c_uint64 * n)(*[a.shape[0] for a in clips])
    for delay in (0.0, 0.005, 0.03):   # free before the call got going, early in it, in the middle of it
        tr = api.Transcriber(tiny_dir, api.ARCH_TINY, {"vad_threshold": "0", "batch_clips": "8"})
        h = tr.handle
        tr.handle = -1   # this test frees the handle itself
        started = threading.Event()
        result = {}
        outs = (C.POINTER(api.TranscriptC) * n)()

        def caller():
            started.set()
            result["rc"] = api.lib().moonshine_transcribe_batch_without_streaming(h, ptrs, lens, n, 16000, 0, outs)

        t = threading.Thread(target=caller)
        t.start()
        started.wait()
        time.sleep(delay)
        api.lib().moonshine_free_transcriber(h)
        t.join(timeout=300)
        assert not t.is_alive()
        assert result["rc"] in (api.MOONSHINE_ERROR_NONE, api.MOONSHINE_ERROR_INVALID_HANDLE, api.MOONSHINE_ERROR_UNKNOWN)
        out = C.POINTER(api.TranscriptC)()
        a = np.zeros(16000, np.float32)
        rc = api.lib().moonshine_transcribe_without_streaming(h, a.ctypes.data_as(C.POINTER(C.c_float)), 16000, 16000, 0, C.byref(out))
        assert rc == api.MOONSHINE_ERROR_INVALID_HANDLE
