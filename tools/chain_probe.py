"""Per-kernel cost inside a replayed decode graph (msh_profile_decode_chain) at small batch sizes."""
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
for B in [int(x) for x in (sys.argv[1:] or ["1", "16", "64"])]:
    audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
    ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
    eng.transcribe_tokens(device_ptrs=ptrs, forced_steps=65)
    eng.profile_reset()
    eng.profile_decode_chain(4)
    rows = {}
    for p in eng.profile():
        if p["name"].startswith("chain_") and p["launches"] > 0:
            rows.setdefault(p["name"][6:].split("#")[0], []).append(p["ms"] / p["launches"] * 1e3)
    rows = {k: sum(v) / len(v) for k, v in rows.items()}
    n = {"dec_qkv_gemm": 8, "dec_self_attention": 8, "dec_proj_resid_gemm": 16, "dec_crossq_gemm": 8, "dec_cross_attention": 8,
         "dec_fc1_swiglu_gemm": 8, "dec_fc2_resid_gemm": 8, "dec_final_layernorm": 1, "dec_lm_head_gemm": 1, "dec_argmax_advance": 1}
    tot = sum(rows.get(k, 0.0) * c for k, c in n.items())
    print(f"B={B}: step {tot:.1f} us  " + "  ".join(f"{k[4:]}={v:.2f}" for k, v in sorted(rows.items())), flush=True)
