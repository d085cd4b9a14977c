"""Localise what differs when two engines run concurrently on one GPU (see inflight_probe.py)."""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

B, DEC = 256, 65
cfg = ARCHS["base"]
with tempfile.TemporaryDirectory() as d:
    w = make_weights(cfg, 0)
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    engs = []
    for _ in range(2):
        e = Engine(0)
        e.load_weights_file(path)
        e.set_keep_encoder_output(True)
        engs.append(e)
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
CH = [0, 1, 100, 255]
def enc_sig(e):
    return [e.encoder_output(c).copy() for c in CH]
refs = []
for e in engs:
    e.encode(device_ptrs=ptrs)
    s = enc_sig(e)
    t, _ = e.decode(forced_steps=DEC)
    refs.append((s, t))
print("engines agree serially:", all(np.array_equal(a, b) for a, b in zip(refs[0][0], refs[1][0])), refs[0][1] == refs[1][1], flush=True)

def run_pair(fa, fb, reps=4):
    res = [[], []]
    def wa():
        for _ in range(reps): res[0].append(fa())
    def wb():
        for _ in range(reps): res[1].append(fb())
    ta, tb = threading.Thread(target=wa), threading.Thread(target=wb)
    ta.start(); tb.start(); ta.join(); tb.join()
    return res

def do_enc(e):
    def f():
        e.encode(device_ptrs=ptrs); e.synchronize()
        return enc_sig(e)
    return f
def do_dec(e):
    def f():
        return e.decode(forced_steps=DEC)[0]
    return f
def ndiff_tok(t, ref):
    return sum(1 for a, b in zip(t, ref) if a != b)

# 1. encode || encode
r = run_pair(do_enc(engs[0]), do_enc(engs[1]))
for k in (0, 1):
    bad = [[float(np.abs(a - b).max()) for a, b in zip(s, refs[k][0])] for s in r[k]]
    print("enc||enc  engine", k, "max abs diff per rep/clip", bad, flush=True)
# 2. decode || decode (encoders redone serially first)
for e in engs:
    e.encode(device_ptrs=ptrs); e.synchronize()
r = run_pair(do_dec(engs[0]), do_dec(engs[1]))
for k in (0, 1):
    print("dec||dec  engine", k, "clips differing per rep", [ndiff_tok(t, refs[k][1]) for t in r[k]], flush=True)
    for t in r[k]:
        d = [(i, next(j for j in range(len(a)) if a[j] != b[j])) for i, (a, b) in enumerate(zip(t, refs[k][1])) if a != b]
        print("    (clip, first differing step):", d[:16], flush=True)
if os.environ.get("DIAG_SHORT"):
    sys.exit(0)
# 3. encode(A) || decode(B)
r = run_pair(do_enc(engs[0]), do_dec(engs[1]))
print("enc(A)||dec(B): A enc diffs", [[float(np.abs(a - b).max()) for a, b in zip(s, refs[0][0])] for s in r[0]], flush=True)
print("enc(A)||dec(B): B clips differing", [ndiff_tok(t, refs[1][1]) for t in r[1]], flush=True)
t = r[1][-1]
for i, (a, b) in enumerate(zip(t, refs[1][1])):
    if a != b:
        j = next(k for k in range(len(a)) if a[k] != b[k])
        print("  first differing clip", i, "first differing step", j, a[j - 2:j + 3], b[j - 2:j + 3]); break
