import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def tables():
    import numpy as np
    d = os.path.join(ROOT, "confidence_bootstrapping_amd", "data")
    return np.load(os.path.join(d, "so3_exp_score_norms.npy")), np.load(os.path.join(d, "torus_score_norm.npy"))


@pytest.fixture(scope="session")
def score_model():
    from confidence_bootstrapping_amd.utils import make_score_model
    model, args = make_score_model(seed=0)
    return model, args
