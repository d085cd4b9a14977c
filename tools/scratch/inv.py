import os, sys, tempfile
import numpy as np
sys.path.insert(0, ".")
os.environ["MSH_DEV_KNOBS"] = "1"
from moonshine_amd.hip_api import Engine
from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

def engine(arch, seed):
    cfg = ARCHS[arch]
    w = make_weights(cfg, seed)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "model.safetensors")
    save_safetensors(path, w, {"arch": cfg.name, "heads": str(cfg.heads)})
    e = Engine(0, dev=True)
    e.load_weights_file(path)
    return e

clips = [make_audio(300 + i, 16000 + 37 * i) for i in range(200)]
for ko in ("0", "1"):
    for rs in ("0", "1"):
        os.environ["MSH_CONV_KORDER"] = ko
        os.environ["MSH_GN_ROWSUMS"] = rs
        e = engine("tiny", 3)
        e.set_keep_encoder_output(True)
        e.encode(clips[:8])
        small = [e.encoder_output(i) for i in range(8)]
        st_small = e.debug_read("gn_stats").view(np.float32).reshape(-1, 2).copy()
        e.encode(clips)
        big = [e.encoder_output(i) for i in range(8)]
        st_big = e.debug_read("gn_stats").view(np.float32).reshape(-1, 2)[:8].copy()
        eq = [bool(np.array_equal(a, b)) for a, b in zip(small, big)]
        print(f"KORDER={ko} ROWSUMS={rs}: encoder outputs equal (8 alone vs among 200): {eq}; stats equal: {bool(np.array_equal(st_small, st_big))}", flush=True)
        if not np.array_equal(st_small, st_big):
            print("   stats small", st_small[:3].tolist(), "big", st_big[:3].tolist())
        e.close()
