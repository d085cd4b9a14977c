"""GPU: the fused encoder MLP kernel (k_mlp.hip) on the benchmark's row count, next to the two tiled GEMMs it replaces
(fc1 + GELU, fc2 + residual; LayerNorm not included on that side), same box, uniform random data."""
import ctypes as C
import os
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.hip_api import load_dev_library as load_library  # noqa: E402

lib = load_library()
lib.msh_test_mlp_microbench.restype = C.c_float
lib.msh_test_mlp_microbench.argtypes = [C.c_int32] * 5
lib.msh_test_gemm_microbench.restype = C.c_float
lib.msh_test_gemm_microbench.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
for name, R, D, F in (("base 256x10s", 108544, 416, 1664), ("base 32 clips", 13568, 416, 1664), ("tiny 256x10s", 108544, 288, 1152)):
    fl = 4.0 * R * D * F
    best = min(lib.msh_test_mlp_microbench(R, D, F, 10, 0) for _ in range(3))
    line = f"{name:14s} fused MLP {best:.3f} ms = {fl / best / 1e9:.0f} TFLOP/s = {fl / best / 1e9 / 2500:.3f} of 2.5 PF"
    if D == 416:
        a = min(lib.msh_test_gemm_microbench(R, F, D, D, 5, 0, 10) for _ in range(2))
        b = min(lib.msh_test_gemm_microbench(R, D, F, F, 0, 0, 10) for _ in range(2))
        line += f" | tiled fc1 {a:.3f} + fc2 {b:.3f} = {a + b:.3f} ms ({fl / (a + b) / 1e9:.0f} TFLOP/s, plain bf16 store epilogues)"
    print(line, flush=True)

D, F = 416, 1664
for R in (32768, 65536, 98304, 108544):   # whole rounds of 256 panels, then the benchmark's 3.31 rounds
    ms = min(lib.msh_test_mlp_microbench(R, D, F, 10, 0) for _ in range(3))
    print(f"R = {R:6d} ({R / 128 / 256:.2f} rounds of 256 panels) {ms:.3f} ms = {4.0 * R * D * F / ms / 1e9:.0f} TFLOP/s", flush=True)
R = 32768
names = {1: "no DMA", 2: "no GELU", 3: "no DMA, no GELU", 4: "fc1 on two accumulators", 8: "6-deep fragment ring", 16: "DMAs issued together", 32: "prologue + epilogue only", 67: "MFMAs + barrier only", 195: "MFMAs only", 199: "MFMAs only, fc1 on two accumulators", 322: "weight stream only (DMA + waits + barrier)"}
for abl, nm in names.items():
    ms = min(lib.msh_test_mlp_microbench(R, D, F, 10, abl) for _ in range(2))
    print(f"ablation {abl:2d} ({nm:24s}) {ms:.3f} ms = {4.0 * R * D * F / ms / 1e9:.0f} TFLOP/s", flush=True)
