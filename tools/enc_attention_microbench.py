"""Encoder self-attention kernels alone (k_attn.hip, development library): ms per launch of every variant / ablation on the
benchmark's shape (256 clips x 415 frames, D = 416, 8 heads), and the resident kernel's output against the streaming one's.

    MSH_DEV_KNOBS=1 python tools/enc_attention_microbench.py [n_clips] [T]
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from moonshine_amd.hip_api import load_dev_library  # noqa: E402

lib = load_dev_library()
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 415
D, H = 416, 8
rows = (T + 7) // 8 * 8
flops = 4.0 * n_clips * H * T * T * 52
names = {0: "streaming, 2 x 4 waves", 1: "streaming, 7 waves", 2: "streaming, 2 tiles per wave", 100: "resident (product)",
         50: "resident, 12 waves x 3 tiles", 51: "resident, 16 waves x 2 tiles",
         101: "resident, no global loads", 102: "resident, loads waited for but zeroed", 104: "resident, no exp2", 108: "resident, no MFMA", 113: "resident, no loads / exp2 / MFMA",
         116: "resident, staging + epilogue only"}
outs = {}
for v in (0, 100, 50, 51):
    o = np.zeros((n_clips * rows, D), np.uint16)
    ms = lib.msh_test_enc_attention(v, n_clips, T, D, H, 0, o.ctypes.data)
    assert ms >= 0
    outs[v] = o
def f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


for v in (100, 50, 51):
    print(f"variant {v} vs streaming: max abs diff", float(np.abs(f32(outs[0]) - f32(outs[v])).max()), " differing bf16 values",
          int((outs[0] != outs[v]).sum()), "of", outs[0].size)
for rnd in range(2):
    for v in names:
        ms = lib.msh_test_enc_attention(v, n_clips, T, D, H, 20, None)
        print(f"variant {v:4d}  {ms * 1000:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s   {names[v]}", flush=True)
