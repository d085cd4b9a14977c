"""Per-kernel cost inside a replayed decode graph (msh_profile_decode_chain) at small batch sizes."""
import os, sys, tempfile
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
for B in [int(x) for x in (sys.argv[1:] or ["1", "16", "64"])]:
    audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
    ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
    eng.transcribe_tokens(device_ptrs=ptrs, forced_steps=65)
    eng.profile_reset()
    eng.profile_decode_chain(4)
    rows = {}
    for p in eng.profile():
        if p["name"].startswith("chain_") and p["launches"] > 0:
            rows[p["name"][6:]] = p["ms"] / p["launches"] * 1e3
    layer = sum(v for k, v in rows.items() if k not in ("dec_final_layernorm", "dec_lm_head_gemm", "dec_argmax_advance", "empty_step"))
    print(f"B={B}: layer (sum of its kernels' chain costs) {layer:.1f} us  " + "  ".join(f"{k[4:] if k.startswith('dec_') else k}={v:.2f}" for k, v in sorted(rows.items())), flush=True)
