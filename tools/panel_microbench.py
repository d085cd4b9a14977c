#!/usr/bin/env python3
"""ms per launch of the encoder QKV panel kernel (k_panel.hip) and its ablations: python tools/panel_microbench.py [abl ...]
(each ablation runs in its own process: MSH_PANEL_ABL is read once)."""
import ctypes as C
import os
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from moonshine_amd.hip_api import load_dev_library as load_library
    lib = load_library()
    lib.msh_test_qkv_panel.restype = C.c_float
    lib.msh_test_qkv_panel.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    for R in [int(x) for x in sys.argv[2].split(",")]:
        ms = lib.msh_test_qkv_panel(R, 416, 20, None, None, None, None, None)
        print(f"abl {os.environ.get('MSH_PANEL_ABL', '0'):>3}  R = {R:7d} ({R / 128 / 256:.2f} rounds of 256 panels)  {ms:.3f} ms = "
              f"{2.0 * R * 416 * 1248 / ms / 1e9:.0f} TFLOP/s")
    sys.exit(0)
NAMES = {0: "product", 1: "no stores", 2: "no finish arithmetic", 3: "no finish, no stores", 4: "no DMA",
         8: "no LayerNorm loads", 16: "no RoPE factor loads", 32: "full vmcnt(0) at the mid-stage wait"}
for a in ([int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 4, 8, 16, 32]):
    env = dict(os.environ, MSH_PANEL_ABL=str(a))
    rows = "32768,65536,106496" if a == 0 else "65536"
    out = subprocess.run([sys.executable, __file__, "--one", rows], env=env, capture_output=True, text=True)
    print(out.stdout.rstrip() + f"   [{NAMES.get(a, '?')}]" if out.returncode == 0 else out.stderr[-300:])
