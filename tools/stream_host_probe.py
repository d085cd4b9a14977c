"""Where a config-5 update's decode time goes on the HOST side: wall time of decoder_reset and decode_full per update against the
GPU time between the engine's own events (msh_stream_query 12 / 13), with the engine's phase prints (MSH_STREAM_TIMING=1)."""
import os, sys, time, tempfile
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from moonshine_amd.hip_api import StreamEngine
from moonshine_amd.synth import STREAMING_ARCHS, make_audio, write_streaming_model_dir

cfg = STREAMING_ARCHS["medium_streaming"]
S = 64
with tempfile.TemporaryDirectory() as d:
    write_streaming_model_dir(d, cfg, seed=0)
    eng = StreamEngine(os.path.join(d, "model.safetensors"), cfg.streaming_config_json(), device=0, max_slots=S, max_memory_frames=512)
audio = [make_audio(1000 + i, 160000) for i in range(S)]
slots = [eng.open() for _ in range(S)]
upd = 8000
def step(timed):
    for s in slots:
        eng.reset(s)
    processed, last = 0, [[] for _ in range(S)]
    rows = []
    for u in range(20):
        n = (u + 1) * upd
        cc = (n - processed) // 1280
        if cc:
            eng.process_audio(slots, [a[processed:processed + cc * 1280] for a in audio]); processed += cc * 1280
        eng.encode(slots, [u == 19] * S)
        g0 = (eng.query(0, 12), eng.query(0, 13), eng.query(0, 10))
        t0 = time.perf_counter(); eng.decoder_reset(slots)
        t1 = time.perf_counter()
        if u == 0:
            toks, acc = eng.decode_full(slots, max_tokens=[13] * S)
        else:
            toks, acc = eng.decode_full(slots, drafts=last)
        t2 = time.perf_counter()
        g1 = (eng.query(0, 12), eng.query(0, 13), eng.query(0, 10))
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (g1[0] - g0[0]) / 1e3, (g1[1] - g0[1]) / 1e3, g1[2] - g0[2]))
        last = toks
    if timed:
        for u, r in enumerate(rows):
            print(f"update {u:2d}: decoder_reset {r[0]:6.2f} ms  decode_full {r[1]:7.2f} ms wall = AR events {r[2]:7.2f} + verify events {r[3]:6.2f} + {r[1] - r[2] - r[3]:6.2f} outside; {r[4]} AR passes", flush=True)
        tot = [sum(r[i] for r in rows) for i in (0, 1, 2, 3)]
        print("sum: reset %.1f  decode_full %.1f  AR %.1f  verify %.1f  outside %.1f ms (outside = GPU work queued by the update's frontend / encoder calls, which return without waiting for it)" % (tot[0], tot[1], tot[2], tot[3], tot[1] - tot[2] - tot[3]))
step(False)
step(True)
