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
def oracle():
    """CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def fx():
    import flux3d_jl_amd
    return flux3d_jl_amd


@pytest.fixture(scope="session")
def gpu_fx(fx):
    """The package on a box with a GPU; a missing device is a hard failure for -m gpu tests."""
    assert fx.functional(), "no HIP device visible: -m gpu tests must run on the GPU box"
    return fx


@pytest.fixture(scope="session")
def known():
    import json
    with open(os.path.join(GOLDEN, "ref_known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture
def fx_option(fx):
    """set(name, value): fx3d_set_option for the duration of one test (restored at teardown).  The library reads no
    environment variable on its launch path: the tests that exercise the kernels' alternative code paths go through the
    explicit option API (include/flux3d_hip.h "variant switches")."""
    from flux3d_jl_amd import _lib
    saved = {}

    def set(name, value):
        saved.setdefault(name, _lib.get_option(name))
        _lib.set_option(name, int(value))
    yield set
    for name, v in saved.items():
        _lib.set_option(name, v)
