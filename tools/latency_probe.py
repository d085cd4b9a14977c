"""Batch-1 latency of one 10 s clip (the bench's latency_ms leg alone: p50 of 50 runs, encode / decode split) + the per-kernel
chain costs at batch 1.  Knobs are read from the environment (MSH_XSPLIT_M, ...)."""
import os, statistics, sys, tempfile, time
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
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
eng.set_cross_mode("kv")
audio = torch.from_numpy(np.stack([make_audio(0, 160000)])).cuda()
one = [(audio[0].data_ptr(), 160000)]
steps = int(os.environ.get("STEPS", "65"))
for _ in range(3):
    ids = eng.transcribe_tokens(device_ptrs=one, forced_steps=steps)
tot, enc_t, dec_t = [], [], []
for _ in range(50):
    a = time.perf_counter(); eng.encode(device_ptrs=one); eng.synchronize()
    b = time.perf_counter(); eng.decode(forced_steps=steps)
    c = time.perf_counter()
    tot.append((c - a) * 1e3); enc_t.append((b - a) * 1e3); dec_t.append((c - b) * 1e3)
print(f"latency p50 total {statistics.median(tot):.3f} ms  encode {statistics.median(enc_t):.3f}  decode {statistics.median(dec_t):.3f}  "
      f"({statistics.median(dec_t) / steps * 1e3:.1f} us per step incl. replay overhead)  ids[:12] {ids[0][:12]}", flush=True)
REPS = 4
eng.profile_reset()
eng.profile_decode_chain(REPS)
rows, per_step = {}, {}
for p in eng.profile():
    if p["name"].startswith("chain_") and p["launches"] > 0:
        k = p["name"][6:]
        rows[k] = p["ms"] / p["launches"] * 1e3            # us per launch
        per_step[k] = p["ms"] / REPS * 1e3                 # us per decode step (all launches of the group)
tot_us = sum(v for k, v in per_step.items() if k != "empty_step")
print(f"B=1 chain: step {tot_us:.1f} us  " + "  ".join(f"{k[4:] if k.startswith('dec_') else k}={rows[k]:.2f}x{round(per_step[k] / rows[k])}" for k in sorted(rows)), flush=True)

# the encoder of the single clip, kernel by kernel (HIP-event scopes: ~4.8 us of overhead inside each figure)
eng.profile_reset()
eng.profile_enable(True)
for _ in range(5):
    eng.encode(device_ptrs=one)
    eng.synchronize()
eng.profile_enable(False)
rows = [(p["name"], p["ms"] / 5 * 1e3, p["launches"] // 5) for p in eng.profile() if not p["name"].startswith("chain_") and p["launches"] > 0]
print("B=1 encoder (us per encode, launches): " + "  ".join(f"{n}={u:.0f}/{l}" for n, u, l in rows) + f"  sum={sum(u for _, u, _ in rows):.0f}", flush=True)
