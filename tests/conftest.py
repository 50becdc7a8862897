import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gpu_ctx():
    """One librtmi context on device 0. Fails (never skips) when the library or GPU is missing:
    a GPU test that silently passes without the HIP path proves nothing."""
    from raytracing_amd import lib
    ctx = lib.Context(0)
    yield ctx
    ctx.close()
