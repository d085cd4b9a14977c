"""Teacher-forced eager decode on two engines at once: which (step, clip) logits differ from the serial run, by how much."""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

B, DEC = int(os.environ.get("B", "256")), int(os.environ.get("DEC", "12"))
cfg = ARCHS["base"]
with tempfile.TemporaryDirectory() as d:
    w = make_weights(cfg, 0)
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    engs = []
    for _ in range(2):
        e = Engine(0)
        e.load_weights_file(path)
        engs.append(e)
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
for e in engs:
    e.encode(device_ptrs=ptrs); e.synchronize()
toks, _ = engs[0].decode(forced_steps=DEC)
teacher = np.asarray(toks, np.int32)
ref = []
for e in engs:
    _, lg = e.decode(forced_steps=DEC, teacher=teacher, want_logits=DEC)
    ref.append(lg.copy())
print("serial engines agree:", np.array_equal(ref[0], ref[1]), flush=True)
_, lg = engs[0].decode(forced_steps=DEC, teacher=teacher, want_logits=DEC)
print("serial repeat agrees:", np.array_equal(ref[0], lg), flush=True)
res = [[], []]
def work(k):
    for _ in range(int(os.environ.get("REPS", "3"))):
        _, lg = engs[k].decode(forced_steps=DEC, teacher=teacher, want_logits=DEC)
        res[k].append(lg.copy())
th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
for t in th: t.start()
for t in th: t.join()
for k in (0, 1):
    for r, lg in enumerate(res[k]):
        d = np.abs(lg - ref[k]).max(axis=2)          # [steps, clips]
        bad = np.argwhere(d > 0)
        if len(bad) == 0:
            print(f"engine {k} rep {r}: bit-identical"); continue
        first = {}
        for s, c in bad:
            first.setdefault(int(c), int(s))
        print(f"engine {k} rep {r}: {len(first)} clips differ; (clip: first step, max|d| at that step, logit scale)", flush=True)
        for c, s in sorted(first.items())[:40]:
            print(f"    clip {c:3d} step {s:2d} max|d| {d[s, c]:.3e}  max|logit| {np.abs(ref[k][s, c]).max():.2f}  n_cols_differ {(lg[s, c] != ref[k][s, c]).sum()}")
