import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "baseline(rank): oracle / property comparison of a BASELINE.json config; collected first")


def pytest_collection_modifyitems(config, items):
    """The driver runs the GPU suite with -x: the comparisons on BASELINE.json's own configurations (C1 .. C5) come first, in config
    order, so that no later failure can leave one of them unexercised.  Stable for everything else."""
    def rank(item):
        mk = item.get_closest_marker("baseline")
        return (0, mk.args[0] if mk and mk.args else 0) if mk else (1, 0)
    items.sort(key=rank)


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first():
    """On a GPU box: bring torch's HIP context up before the first test.  The distributed-schedule tests wrap the library's chain
    stream in a torch stream; when torch's lazy initialisation happened for the first time AFTER the 50 GB config-5 test of the same
    process (`pytest tests/test_gpu_parity.py -k "config5 or distributed"`) it reported "No HIP GPUs are available" on every box
    tried, while the same sequence in a plain script (tools/dbg_c5_then_torch.py) initialises fine.  Order-independent this way."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "pyref_qr.npz"))


@pytest.fixture(scope="session")
def golden_precision():
    """Inputs made by the reference's generate_matrix (python/utils.py:13-24) + its own fp32 / fp64 errors on them."""
    return np.load(os.path.join(ROOT, "tests", "golden", "pyref_precision.npz"))


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    return pyoracle


# The reference's synthetic sweep, Cuda/qr.cu:1762-1783 (m, n, r)
REF_SWEEP = [(6, 4, 2), (6, 4, 1), (6, 4, 3), (12, 8, 4), (12, 8, 5), (12, 8, 6), (12, 8, 2), (12, 8, 8),
             (12, 8, 3), (24, 16, 8), (24, 16, 12), (60, 40, 8), (60, 40, 16), (80, 80, 16), (97, 90, 16),
             (100, 80, 16), (128, 80, 16), (129, 80, 16), (240, 160, 16), (600, 400, 16)]
