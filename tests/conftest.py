import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ops_gold():
    import numpy as np
    return dict(np.load(os.path.join(GOLD, "ops.npz")))


@pytest.fixture(scope="session")
def tiny_gold():
    import numpy as np
    return dict(np.load(os.path.join(GOLD, "qwen2vl_tiny.npz")))


@pytest.fixture(scope="session", autouse=True)
def _torch_initialises_the_gpu_first():
    """On a GPU box torch's HIP runtime comes up before the first call into libmllm_hip.so: in the other order (a file whose first tests only use the library, run on its own)
    torch later reports "No HIP GPUs are available".  No effect without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
