"""Tiled-GEMM microbenchmark on the GPU box: configurations x ablations on the encoder shapes."""
import ctypes as C
import sys

sys.path.insert(0, ".")
from moonshine_amd.hip_api import load_dev_library as load_library

lib = load_library()
lib.msh_test_gemm_microbench.restype = C.c_float
lib.msh_test_gemm_microbench.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
R = 107520
shapes = [("conv2", 2 * R, 832, 2912, 1248), ("fc1", R, 1664, 416, 416), ("fc2", R, 416, 1664, 1664), ("qkv", R, 1248, 416, 416)]
cfgs = {0: "4w 128x208 s3", 1: "8w 256x208 s4", 2: "4w 256x208 s4", 3: "4w 128x208 s2", 4: "4w 128x208 s4"}
abls = {0: "full", 1: "noDMA", 2: "noMFMA", 3: "noDMA+noMFMA", 5: "noDMA+noLDSread", 6: "DMA only"}
for name, M, N, K, lda in shapes:
    fl = 2.0 * M * N * K
    for cfg in ([0, 1, 2, 3, 4] if name in ("conv2", "fc1") else [0, 1]):
        row = []
        for abl in ([0, 1, 2, 3, 5, 6] if cfg in (0, 1) else [0]):
            ms = lib.msh_test_gemm_microbench(M, N, K, lda, cfg, abl, 5)
            row.append(f"{abls[abl]}={ms:.3f}ms({fl / ms / 1e9:.0f}TF)")
        print(f"{name:6s} cfg{cfg} [{cfgs[cfg]}]  " + "  ".join(row), flush=True)

# A-stationary kernel (K = 416 shapes only)
for name, M, N, K, lda in shapes + [("oproj", R, 416, 416, 416), ("crosskv", R, 6656, 416, 416)]:
    if K != 416:
        continue
    fl = 2.0 * M * N * K
    row = []
    for cfg in (0, 5, 6):
        ms = lib.msh_test_gemm_microbench(M, N, K, lda, cfg, 0, 5)
        row.append(f"cfg{cfg}={ms:.3f}ms({fl / ms / 1e9:.0f}TF)")
    print(f"{name:8s} astat  " + "  ".join(row), flush=True)
