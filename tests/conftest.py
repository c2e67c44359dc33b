import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))             # tests/wavio.py: the tests' own WAV reader
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_simple():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "model_simple_seed1234.npz")))


@pytest.fixture(scope="session")
def golden_full():
    """3-conv WakewordModel outputs produced by the reference's own class text (tests/golden/make_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "model_full_seed1234.npz")))


@pytest.fixture(scope="session")
def golden_grads():
    """loss + parameter gradients of the reference SimpleWakewordModel (tests/golden/make_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "grads_simple_seed1234.npz")))


@pytest.fixture(scope="session")
def golden_grads_full():
    """loss + parameter gradients of the reference's own 3-conv WakewordModel class (tests/golden/make_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "grads_full_seed1234.npz")))
