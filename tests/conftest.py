import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure)."""
    o = entry.load_oracle()
    o.lib()  # builds oracle/_build on first use
    return o


@pytest.fixture(scope="session")
def pkg():
    p = entry.load_package()
    if not os.path.exists(p.library_path()):
        p.build_library()
    return p


@pytest.fixture(scope="session")
def ctx(pkg):
    """A GPU context.  No fallback: on a GPU box a missing/unloadable library is a failure."""
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def fixture_rgb():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "tulips_medium640_rgb.png")).convert("RGB"))


@pytest.fixture(scope="session")
def fixture_rgba(fixture_rgb):
    return np.dstack([fixture_rgb, np.full(fixture_rgb.shape[:2], 255, np.uint8)])


@pytest.fixture(scope="session")
def golden_weights():
    return json.load(open(os.path.join(GOLDEN, "gauss_weights_ref.json")))


@pytest.fixture(scope="session")
def regression():
    return json.load(open(os.path.join(GOLDEN, "oracle_regression.json")))


def rand_rgba(h, w, seed, alpha=255, n=None):
    rng = np.random.default_rng(seed)
    shape = (h, w, 4) if n is None else (n, h, w, 4)
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    if alpha is not None:
        a[..., 3] = alpha
    return a
