"""Encoder self-attention kernels alone (k_attn.hip, development library): ms per launch of every variant / ablation on the
benchmark's shape (256 clips x 415 frames, D = 416, 8 heads).

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
names = {0: "product (8 waves x 4 tiles, one chunk)", 1: "product, chunk-loop instantiation", 50: "12 waves x 3 tiles", 51: "16 waves x 2 tiles",
         101: "no global loads", 102: "loads waited for but zeroed", 104: "no exp2", 108: "no MFMA", 112: "no exp2, no MFMA"}
outs = {}
for v in (0, 1, 50, 51):
    o = np.zeros((n_clips * rows, D), np.uint16)
    ms = lib.msh_test_enc_attention(v, n_clips, T, D, H, 0, o.ctypes.data)
    assert ms >= 0
    outs[v] = o
for v in (1, 50, 51):
    print(f"variant {v} == variant 0, bit for bit:", bool((outs[0] == outs[v]).all()))
for rnd in range(2):
    for v in names:
        ms = lib.msh_test_enc_attention(v, n_clips, T, D, H, 20, None)
        print(f"variant {v:4d}  {ms * 1000:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s   {names[v]}", flush=True)
