"""Probe: K batches of 256 clips through F engines in flight (one host thread each, own stream / workspace) on one GPU.
F = 1 is the serial bench loop; F = 2 overlaps the encoder of one batch with the decode loop of the other."""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

B, STEPS, DEC = 256, int(os.environ.get("K", "8")), 65
cfg = ARCHS["base"]
with tempfile.TemporaryDirectory() as d:
    w = make_weights(cfg, 0)
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    engs = []
    for _ in range(3):
        e = Engine(0)
        e.load_weights_file(path)
        engs.append(e)
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
for e in engs:
    for _ in range(2):
        ref = e.transcribe_tokens(device_ptrs=ptrs, forced_steps=DEC)
for F in (1, 2, 3, 1, 2):
    nxt = [0]
    lock = threading.Lock()
    outs = [None] * STEPS
    def worker(e):
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= STEPS:
                return
            outs[i] = e.transcribe_tokens(device_ptrs=ptrs, forced_steps=DEC)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(engs[k],)) for k in range(F)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(o == ref for o in outs)
    print(f"in_flight={F} steps={STEPS} ms_per_step={dt/STEPS*1e3:.2f} audio_s_per_s={B*10*STEPS/dt:.0f} identical_tokens={ok}", flush=True)
