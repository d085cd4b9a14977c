"""What SEQUENCES of decode kernels cost inside a replayed graph (MSH_CHAIN_MASKS, Engine::profile_decode_chain), against the
sum of their members timed alone: a kernel that is cheap back to back with itself can be dear behind or in front of
another (cold operands, the next kernel's start).  Groups per layer: bit 0 qkv, 1 self-attention, 2 o-proj, 3 cross-q,
4 cross-attention, 5 context / cross o-proj, 6 fc1, 7 fc2.   python tools/chain_masks.py [batch]"""
import os, sys, tempfile
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import numpy as np
sys.path.insert(0, ".")
MASKS = ["0x08", "0x10", "0x20", "0x18", "0x0c", "0x30", "0x38", "0x3c", "0x03", "0xc0", "0xff"]
os.environ.setdefault("MSH_CHAIN_MASKS", ",".join(MASKS))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
audio = torch.from_numpy(np.stack([make_audio(i, 160000) for i in range(B)])).cuda()
ptrs = [(audio[i].data_ptr(), 160000) for i in range(B)]
eng.transcribe_tokens(device_ptrs=ptrs, forced_steps=65)
REPS = 4
for rnd in range(2):
    eng.profile_reset()
    eng.profile_decode_chain(REPS)
    prof = {p["name"]: p for p in eng.profile() if p["launches"] > 0}
    names = ["dec_qkv_gemm", "dec_self_attention", "dec_proj_resid_gemm", "dec_crossq_gemm", "dec_cross_attention", "dec_ctx_resid_gemm",
             "dec_fc1_swiglu_gemm", "dec_fc2_resid_gemm"]
    alone = []
    for n in names:
        p = prof.get("chain_" + n)
        alone.append(p["ms"] / p["launches"] * 1e3 if p else float("nan"))
    print(f"round {rnd}  alone (us per launch): " + "  ".join(f"{n[4:]}={v:.2f}" for n, v in zip(names, alone)))
    for m in os.environ["MSH_CHAIN_MASKS"].split(","):
        p = prof.get("chainmask_" + m)
        if not p:
            continue
        mask = int(m, 0)
        members = [i for i in range(8) if (mask >> i) & 1]
        per_layer = p["ms"] / (2 * REPS * 8) * 1e3
        want = sum(alone[i] for i in members)
        print(f"  mask {m:>5} [{' + '.join(names[i][4:] for i in members)}]: {per_layer:.2f} us per layer in sequence, sum of members alone {want:.2f}  ({per_layer - want:+.2f})")
