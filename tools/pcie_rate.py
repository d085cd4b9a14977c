"""PCIe-inclusive rate of the offline path: the same 256 x 10 s batch as bench.py, but with the clips handed over as
host buffers (what the C API's batch call does), so every step pays the host -> device copy.  For DESIGN.md."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

cfg = ARCHS["base"]
eng = Engine(0)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, make_weights(cfg, 0), {"arch": cfg.name, "heads": str(cfg.heads)})
    eng.load_weights_file(path)
clips = [make_audio(i, 160000) for i in range(256)]
for _ in range(2):
    eng.transcribe_tokens(clips, forced_steps=65)
t0 = time.perf_counter()
K = 5
for _ in range(K):
    eng.transcribe_tokens(clips, forced_steps=65)
dt = (time.perf_counter() - t0) / K
print(f"host-buffer batch of 256 x 10 s: {dt * 1e3:.2f} ms per step, {2560.0 / dt:.0f} audio-seconds/sec (PCIe-inclusive)")
