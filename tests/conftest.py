import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Fixtures are plain arrays: numpy.load with allow_pickle=False (its default)."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def theta_dict(theta):
    """theta[12] = pi, eta, epsilon, gamma[3], mu[3], sigma[3]."""
    theta = np.asarray(theta, dtype=np.float64)
    return dict(pi=float(theta[0]), eta=float(theta[1]), epsilon=float(theta[2]),
                gamma=theta[3:6].copy(), mu=theta[6:9].copy(), sigma=theta[9:12].copy())
