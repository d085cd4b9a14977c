import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.hip_api import load_library
lib = load_library()
for dyn in (0, 1024, 2048, 4096, 8192, 16384, 24576, 31744):
    bad = lib.msh_test_gemm_microbench(4096, 1664, 416, 416, 10, dyn, 3)
    print(f"dynamic LDS {dyn:6d} B -> workgroup allocation {64512 + dyn:6d} B: mismatching outputs over 3 launches: {int(bad)}", flush=True)
