import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.Api("orc")


@pytest.fixture(scope="session")
def reference():
    """The real reference (oracle/_ref/); only in the build container or where _ref travelled."""
    import orc
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    return orc.Api("ref")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    import gpu_api
    return gpu_api.GpuApi()
