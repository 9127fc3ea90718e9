import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
