import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding
    return binding.load()


@pytest.fixture(scope="session")
def host():
    """The shipped host interface (librt_host.so above librt_amd.so)."""
    from raytracer_amd import api
    return api.load()


@pytest.fixture(autouse=True)
def _fresh_handles(request):
    yield
    # Handles are process-global per library; drop them between tests.
    from oracle import binding
    if binding._API is not None:
        binding._API.reset()
    from raytracer_amd import api
    if api._API is not None:
        api._API.reset()
