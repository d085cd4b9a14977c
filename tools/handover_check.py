import os, sys, time, json
sys.path.insert(0, ".")
import numpy as np
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors
cfg = ARCHS["base"]
w = make_weights(cfg, 0)
save_safetensors("/tmp/base.safetensors", w, {"arch": "base"})
clips = [make_audio(700 + i, 160000) for i in range(256)]
res = {}
for knob in ("1", "0"):
    os.environ["MSH_ENC_LN_HANDOVER"] = knob
    e = Engine(0); e.load_weights_file("/tmp/base.safetensors")
    e.set_keep_encoder_output(True)
    e.encode(clips)
    enc = np.stack([e.encoder_output(i) for i in (0, 50, 255)])
    ids = e.transcribe_tokens(clips, forced_steps=20)
    e.profile_enable(True); e.profile_reset()
    for _ in range(3): e.encode(clips)
    e.synchronize()
    prof = {p["name"]: (round(p["ms"] / max(p["launches"], 1), 4), p["launches"]) for p in e.profile()}
    res[knob] = (enc, ids, prof)
    e.close()
a, b = res["1"], res["0"]
print("encoder max abs diff handover vs not:", float(np.abs(a[0] - b[0]).max()), " rel rms", float(np.sqrt(((a[0]-b[0])**2).mean()) / np.sqrt((b[0]**2).mean())))
print("ids equal on", sum(x == y for x, y in zip(a[1], b[1])), "of", len(a[1]))
for k in ("enc_qkv_panel", "enc_oproj_mlp_fused", "enc_attention"):
    print(k, "handover", a[2].get(k), " old", b[2].get(k))
