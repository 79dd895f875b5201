import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library on a machine with a GPU.  A gpu-marked test on a box
    without a usable device is a FAILURE, never a skip or a fallback."""
    import q3lib
    lib = q3lib.hip_lib()
    if lib.q3_device_count() <= 0:
        pytest.fail("gpu test started without a usable HIP device")
    return lib


@pytest.fixture(scope="session")
def orc():
    import q3lib
    return q3lib.oracle_lib()


@pytest.fixture(scope="session")
def host():
    import q3lib
    return q3lib.host_lib()
