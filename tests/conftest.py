import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): the checker, never the thing under test in -m gpu."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if stale) and loads the product library."""
    from allwave_amd import build, ffi
    build.build_hip()
    return ffi.load()


@pytest.fixture(scope="session", params=["auto", "one_wave"])
def engine(hip_lib, request):
    """The GPU engine, twice: with its own choice of kernel flavour per batch (small batches, long
    sequences and very unequal pairs get four waves per pair) and pinned to the one-wave-per-pair
    throughput kernel that bench.py measures."""
    from allwave_amd import ffi
    e = ffi.Engine(device=0, flags=ffi.AWV_F_ONE_WAVE if request.param == "one_wave" else 0)
    yield e
    e.close()
