import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def lenet15_real():
    import numpy as np
    from gpd_amd import synth
    real = dict(np.load(os.path.join(ROOT, "tests", "golden", "lenet15_params.npz")))
    # ip1 / 128: logits of the size a trained LeNet produces (|score| < 20), where "within 1e-4" can be decided for the split path
    return synth.lenet_weights(15, real=real, trained_magnitude=True)


@pytest.fixture(scope="session")
def cloud30k():
    from gpd_amd import synth
    return synth.make_cloud(1234, 30000)
