"""Encoder of a 256 x 10 s batch (and a ragged one) with MSH_ENC_SPLIT=0 / 1: encoder output and ids must be identical."""
import os, sys, tempfile
os.environ["MSH_DEV_KNOBS"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, write_model_dir
cfg = ARCHS["base"]
with tempfile.TemporaryDirectory() as d:
    write_model_dir(d, cfg, seed=0)
    e = Engine(0)
    e.load_weights_file(os.path.join(d, "model.safetensors"))
e.set_cross_mode("absorbed")
e.set_keep_encoder_output(True)
for name, clips in (("uniform", [make_audio(100 + i, 160000) for i in range(256)]),
                    ("ragged", [make_audio(900 + i, 160000 - 1280 * (i % 40)) for i in range(256)])):
    outs = []
    for sp in ("0", "1"):
        os.environ["MSH_ENC_SPLIT"] = sp
        ids = e.transcribe_tokens(clips, forced_steps=12)
        enc = [e.encoder_output(c) for c in (0, 100, 127, 128, 200, 255)]
        outs.append((ids, enc))
    same_ids = outs[0][0] == outs[1][0]
    same_enc = all(np.array_equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    print(name, "ids equal", same_ids, "encoder output equal", same_enc)
    assert same_ids and same_enc
print("ok")
