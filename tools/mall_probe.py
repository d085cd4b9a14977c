"""Probe: does the cross-attention kernel run faster when its K / V come from the memory-side cache (MALL) instead of HBM?
Runs the kernel over all layers in turn (1.4 GB working set: HBM) and over layer 0 again and again (177 MB at 256 clips,
88 MB at 128: cache-resident if read lines are kept)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, ".")
import torch
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

cfg = ARCHS["base"]
torch.cuda.set_device(0)
eng = Engine(0)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.safetensors")
    save_safetensors(path, make_weights(cfg, 0), {"arch": cfg.name, "heads": str(cfg.heads)})
    eng.load_weights_file(path)
for B in (256, 128, 64):
    audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
    ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
    eng.transcribe_tokens(device_ptrs=ptrs, forced_steps=8)
    mb = B * 2 * 415 * 416 * 2 / 1e6
    a = eng.profile_cross_attention_ms(25)
    b = eng.profile_cross_attention_ms(-25)
    print(f"B={B}: {mb:.0f} MB per launch; all layers {a*1e3:.2f} us ({mb/a/1e3:.0f} GB/s)   same layer {b*1e3:.2f} us ({mb/b/1e3:.0f} GB/s)", flush=True)
