"""What a ragged sequence of batch shapes costs the decode loop (every new shape captured and instantiated two
graphs, ~600 kernel nodes, under a process-wide mutex).  40 free-running batches whose clip count and longest clip change
from call to call, drawn from 6 distinct shapes: wall time per call with the graph cache (engine.cpp DecodeGroup::graphs),
with one step per replay (MSH_DEC_GRAPH_STEPS=1: the cheapest capture), and the number of graphs instantiated.
Run on the GPU box: python tools/ragged_graph_probe.py"""
import os
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, time, json, tempfile
sys.path.insert(0, %r)
import numpy as np
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors
cfg = ARCHS["base"]; w = make_weights(cfg, 0)
d = tempfile.mkdtemp(); p = os.path.join(d, "model.safetensors")
save_safetensors(p, w, {"arch": cfg.name, "heads": str(cfg.heads)})
e = Engine(0, dev=True); e.load_weights_file(p)
rng = np.random.default_rng(0)
shapes = [(256, 160000), (192, 120000), (224, 96000), (256, 64000), (128, 144000), (160, 80000)]
pool = {n: make_audio(7, n) for _, n in shapes}
batches = {s: [pool[s[1]]] * s[0] for s in shapes}
e.transcribe_tokens(batches[shapes[0]], forced_steps=-1)      # workspaces settle on the largest shape
c0 = e.graph_captures()
order = [shapes[int(i)] for i in rng.integers(0, len(shapes), 40)]
t = []
for s in order:
    a = time.perf_counter(); e.transcribe_tokens(batches[s]); t.append((time.perf_counter() - a) * 1e3)
first = {}
for s, ms in zip(order, t):
    first.setdefault(s, []).append(ms)
print(json.dumps({"captures_first": c0, "captures_total": e.graph_captures(), "ms_total": round(sum(t), 1),
                  "per_shape_ms_first_second_later": {f"{s[0]}x{s[1] / 16000:.0f}s": [round(v[0], 1), round(v[1], 1) if len(v) > 1 else None, round(float(np.median(v[2:])), 1) if len(v) > 2 else None] for s, v in first.items()}}))
''' % ROOT
for label, env in [("graph cache, 8 steps per replay", {}), ("graph cache, 1 step per replay", {"MSH_DEC_GRAPH_STEPS": "1"})]:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
    print(label, "->", r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else ("FAILED: " + r.stderr[-800:]))
