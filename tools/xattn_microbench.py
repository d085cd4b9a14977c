"""Rate of the absorbed cross-attention kernel alone (k_xattn.hip) at the benchmark shape; MSH_XATTN_ABL / _SLOTS / _TR select
the variant (developer tool, see tools/gpu_r4c.sh)."""
import os
os.environ.setdefault("MSH_DEV_KNOBS", "1")   # developer switches are honoured only with this set
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.hip_api import load_dev_library as load_library

D, T, M = 416, int(os.environ.get("XA_T", "415")), int(os.environ.get("XA_M", "256"))
rng = np.random.default_rng(1)
rows = (T + 7) // 8 * 8 + 8
enc = rng.standard_normal((M * rows, D)).astype(np.float32)
qt = (rng.standard_normal((M, 8 * D)) * (2.0 / np.sqrt(D))).astype(np.float32)
Ts = np.full(M, T, np.int32)
starts = (np.arange(M) * rows).astype(np.int32)
out = np.zeros((M, 8 * D), np.float32)
lib = load_library()
ms = lib.msh_test_cross_absorbed(qt.ctypes.data, enc.ctypes.data, enc.shape[0], Ts.ctypes.data, starts.ctypes.data, M, D, out.ctypes.data, 200)
mb = M * (T * D * 2 + 8 * D * 6) / 1e6
print(f"abl={os.environ.get('MSH_XATTN_ABL', '0')} cfg={os.environ.get('MSH_XATTN_CFG', '84')} M={M} T={T}: {ms * 1e3:.2f} us per launch, {mb / ms / 1e3:.2f} TB/s")
