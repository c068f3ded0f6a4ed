import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def ref():
    from oracle import sonic_ref
    return sonic_ref


@pytest.fixture(scope="session")
def sonic():
    """The product package on a GPU box; fails loudly (never skips) if the extension or GPU is missing."""
    import sonic_amd
    from sonic_amd import _lib
    _lib.check(_lib.lib().sonic_init(0))
    return sonic_amd


@pytest.fixture
def rng():
    return random.Random(0xC0FFEE)


NCPU = os.cpu_count() or 1
