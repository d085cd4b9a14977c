"""Probe: do co-resident workgroups of the tiled GEMM run main loop and epilogue in lockstep?"""
import ctypes as C
import sys

sys.path.insert(0, ".")
from moonshine_amd.hip_api import load_library

lib = load_library()
lib.msh_test_gemm_microbench.restype = C.c_float
lib.msh_test_gemm_microbench.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
R = 107520
shapes = [("fc1", R, 1664, 416, 416), ("qkv", R, 1248, 416, 416), ("fc2", R, 416, 1664, 1664), ("conv2", 2 * R, 832, 2912, 1248)]
abls = {0: "full (staged stores)", 16: "direct stores", 8: "no epilogue", 0x10400: "b3x1", 0x20400: "b3x2", 0x40400: "b3x4",
        0x10C00: "b11x1", 0x20C00: "b11x2", 0x40C00: "b11x4", 0x10100: "b0x1", 0x20100: "b0x2"}
for name, M, N, K, lda in shapes:
    fl = 2.0 * M * N * K
    row = []
    for abl, label in abls.items():
        ms = lib.msh_test_gemm_microbench(M, N, K, lda, 0, abl, 5)
        row.append(f"{label}={ms:.3f}ms({fl / ms / 1e9:.0f}TF)")
    if K == 416:
        ms = lib.msh_test_gemm_microbench(M, N, K, lda, 5, 0, 5)
        row.append(f"astat(direct stores)={ms:.3f}ms({fl / ms / 1e9:.0f}TF)")
    print(f"{name:6s} " + "  ".join(row), flush=True)
