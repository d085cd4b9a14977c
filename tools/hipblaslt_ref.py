"""Vendor-library reference point for the encoder GEMM shapes: torch.matmul (hipBLASLt / rocBLAS) in bf16 on
the same M, N, K as the tiled kernel's microbenchmark.  Not part of the product; a yardstick for DESIGN.md."""
import time

import torch

R = 107520
shapes = [("conv2", 2 * R, 832, 2912), ("fc1", R, 1664, 416), ("fc2", R, 416, 1664), ("qkv", R, 1248, 416),
          ("oproj", R, 416, 416), ("crosskv", R, 6656, 416), ("lm_head", 256, 32768, 416)]
dev = torch.device("cuda")
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:8s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s (torch.matmul bf16, no epilogue)", flush=True)
