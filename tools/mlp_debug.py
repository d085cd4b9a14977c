"""GPU debugging aid for k_mlp.hip: probes that isolate the residual path, the fc2 path and the LayerNorm / fc1 path."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonshine_amd.hip_api import load_library  # noqa: E402
from oracle import moonshine_ref as ref  # noqa: E402

lib = load_library()
fp = C.POINTER(C.c_float)
lib.msh_test_mlp_run.restype = C.c_int32
lib.msh_test_mlp_run.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, fp, fp]


def run(h, w1, g, b1, w2, b2):
    out = np.ascontiguousarray(h, np.float32).copy()
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w1, g, b1, w2, b2)]
    assert lib.msh_test_mlp_run(out.ctypes.data_as(fp), out.shape[0], out.shape[1], w1.shape[0], *[a.ctypes.data_as(fp) for a in arrs]) == 0
    return out


def want(h, w1, g, b1, w2, b2):
    y = ref.layer_norm_nobias(h, g)
    return (h + ref.gelu(y @ w1.T + b1) @ w2.T + b2).astype(np.float32)


D, F, R = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 40
rng = np.random.default_rng(0)
h = rng.standard_normal((R, D)).astype(np.float32)
w1 = (rng.standard_normal((F, D)) / np.sqrt(D)).astype(np.float32)
w2 = (rng.standard_normal((D, F)) / np.sqrt(F)).astype(np.float32)
b1 = (rng.standard_normal(F) * 0.1).astype(np.float32)
b2 = (rng.standard_normal(D) * 0.1).astype(np.float32)
g = np.ones(D, np.float32)
z = np.zeros_like
np.set_printoptions(precision=3, suppress=True, linewidth=200)
for name, args in [("A residual only (w2 = 0, b2 = 0)", (h, w1, g, b1, z(w2), z(b2))),
                   ("B b2 only", (h, w1, g, b1, z(w2), b2)),
                   ("C fc2 of a constant (w1 = 0, b1 = 1)", (h, z(w1), g, np.ones_like(b1), w2, z(b2))),
                   ("D fc1 bias path (w1 = 0)", (h, z(w1), g, b1, w2, b2)),
                   ("E full", (h, w1, g, b1, w2, b2))]:
    got, wnt = run(*args), want(*args)
    err = np.abs(got - wnt)
    print(f"{name}: max err {err.max():.4f}; rows with err > 0.05: {np.where(err.max(1) > 0.05)[0][:12].tolist()}; cols: {np.where(err.max(0) > 0.05)[0][:24].tolist()}")
    if err.max() > 0.05:
        r = int(np.argmax(err.max(1)))
        print("   row", r, "got ", got[r, :16])
        print("   row", r, "want", wnt[r, :16])
