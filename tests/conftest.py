import os
import sys

import pytest

# lanes ("batches in flight") overlap only on separate HIP hardware queues; the pool size is read from the environment when
# the HIP runtime initialises, and the library leaves the environment to its host (include/moonshine_hip.h msh_set_hw_queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the library honours its developer switches (MSH_ENC_MLP, MSH_XSPLIT_M, ...: kernel variants the tests compare) only when
# this is set too -- a production process cannot pick one up from a stray variable (csrc/msh_common.h dev_getenv)
os.environ.setdefault("MSH_DEV_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def pytest_sessionfinish(session, exitstatus):
    """Write the margins the parity tests measured (tests/margins.py) where the GPU run scripts pick them up."""
    import margins

    path = os.environ.get("MSH_PARITY_MARGINS", os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    try:
        margins.dump(path)
    except OSError:
        pass
