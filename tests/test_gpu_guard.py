"""Memory-safety regression: a slice of the GPU suite again under the library's guard allocator (MSH_GUARD_ALLOC=1:
every device buffer ends on an unmapped page, see moonshine_amd/csrc/msh_common.h device_alloc).  A kernel that reads or
writes past the end of a buffer -- as the staged GEMM epilogue once did with the streaming encoder's fc1 bias (N = 3072
on 208-wide tiles) -- aborts the child process with a GPU memory fault.  tools/gpu_guard.sh runs the WHOLE suite, smoke()
and the default bench this way (profiles/r2v_guard_allocator_run.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    "tests/test_gpu_streaming.py::test_stream_matches_reference_graphs",      # ragged N on the tiled GEMM (medium dims)
    "tests/test_gpu_parity.py::test_base_ragged_batch_vs_oracle",             # ragged clips, offline encoder + decoder
    "tests/test_gpu_capi.py::test_batch_call_equals_single_calls",            # growing buffers across calls
]


@pytest.mark.gpu
def test_suite_slice_under_guard_allocator():
    env = dict(os.environ, MSH_GUARD_ALLOC="1", MSH_GUARD_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", *CASES, "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.gpu
def test_device_allocator_self_test():
    code = ("import sys; sys.path.insert(0, '.'); from moonshine_amd.hip_api import load_dev_library; "
            "sys.exit(0 if load_dev_library().msh_test_device_alloc() == 0 else 3)")
    for extra in ({}, {"MSH_GUARD_ALLOC": "1"}, {"MSH_GUARD_ALLOC": "1", "MSH_GUARD_ALIGN": "16"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, **extra), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, (extra, r.stderr[-1500:])
