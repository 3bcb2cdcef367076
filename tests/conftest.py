import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import orc as _orc
    _orc.build()
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def golden_buffers():
    return np.load(os.path.join(GOLDEN, "cornell_buffers.npz"))


@pytest.fixture(scope="session")
def golden_256():
    return np.load(os.path.join(GOLDEN, "cornell_256.npz"))


@pytest.fixture(scope="session")
def cornell_oracle_scene(orc, golden_buffers):
    b = golden_buffers
    return orc.Scene(b["primitives"], b["lights"], b["spectra"], b["cie"], b["camera"])


@pytest.fixture(scope="session")
def renderer():
    """One context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    from computeraytracer_amd import Renderer
    r = Renderer(0)
    yield r
    r.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
