import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))


@pytest.fixture(scope="session")
def kat():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "kat.npz"))


@pytest.fixture(scope="session")
def cornell_host():
    from ray_tracing_amd.api import HostScene
    return HostScene("cornell_box", 1.0)


@pytest.fixture(scope="session")
def cornell_gpu(cornell_host):
    from ray_tracing_amd.api import Scene
    s = Scene(cornell_host, 0)
    yield s
    s.close()
