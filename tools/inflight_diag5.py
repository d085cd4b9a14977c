"""Signature of the glitches a co-running DMA GEMM causes in the decode: root events in the K/V cache, element by element."""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from moonshine_amd.hip_api import Engine, load_library
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

B, DEC = 256, int(os.environ.get("DEC", "24"))
cfg = ARCHS["base"]
L, H, DH = cfg.dec_layers, cfg.heads, cfg.hidden // cfg.heads
with tempfile.TemporaryDirectory() as d:
    w = make_weights(cfg, 0)
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    e = Engine(0)
    e.load_weights_file(path)
lib = load_library()
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
e.encode(device_ptrs=ptrs); e.synchronize()
toks, _ = e.decode(forced_steps=DEC)
teacher = np.asarray(toks, np.int32)
SMAX = (DEC + 7) // 8 * 8
def f32(a):
    return (a.astype(np.uint32) << 16).view(np.float32)
def caches():
    return [e.debug_read(n).view(np.uint16).reshape(L, B, H, SMAX, DH)[:, :, :, :DEC].copy() for n in ("cache_k", "cache_v")]
e.decode(forced_steps=DEC, teacher=teacher)
ref = caches()
stop = threading.Event()
def aggressor():
    while not stop.is_set():
        lib.msh_test_gemm_microbench(256, 32768, 416, 416, 0, int(os.environ.get("ABL", "0")), 400)
th = threading.Thread(target=aggressor); th.start()
time.sleep(0.05)
n_ev = 0
for rep in range(int(os.environ.get("REPS", "40"))):
    e.decode(forced_steps=DEC, teacher=teacher)
    got = caches()
    badk = np.argwhere(got[0] != ref[0]); badv = np.argwhere(got[1] != ref[1])
    if len(badk) == 0 and len(badv) == 0:
        continue
    allbad = [(tuple(x), 0) for x in badk] + [(tuple(x), 1) for x in badv]
    # root = smallest (pos, layer); K and V of the same (pos, layer) both count
    p0 = min(x[0][3] for x in allbad); l0 = min(x[0][0] for x in allbad if x[0][3] == p0)
    root = [x for x in allbad if x[0][3] == p0 and x[0][0] == l0]
    n_ev += 1
    kcols = sorted(set(x[0][2] * DH + x[0][4] for x in root if x[1] == 0)); vcols = sorted(set(x[0][2] * DH + x[0][4] for x in root if x[1] == 1))
    clips = sorted(set(x[0][1] for x in root))
    desc = f"rep {rep}: root pos {p0} layer {l0} clips {clips[0]}..{clips[-1]} ({len(clips)}) Kcols {kcols[:6]}{'...' if len(kcols) > 6 else ''} ({len(kcols)}) Vcols {vcols[:6]}{'...' if len(vcols) > 6 else ''} ({len(vcols)})"
    if len(kcols) + len(vcols) <= 4:
        (idx, which) = root[0]
        c = idx[2] * DH + idx[4]
        vals = [(f32(got[which])[l0, m, idx[2], p0, idx[4]], f32(ref[which])[l0, m, idx[2], p0, idx[4]]) for m in clips[:4]]
        desc += f"  n%16={(c + (416 if which == 0 else 832)) % 16} d={idx[4]}  (got, ref) {[(round(float(a), 3), round(float(b), 3)) for a, b in vals]}"
    print(desc, flush=True)
stop.set(); th.join()
print("events:", n_ev)
