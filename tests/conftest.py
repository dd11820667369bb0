import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """libsonarfe context on device 0 (GPU tests only).  Fails loudly if the HIP path is missing."""
    from sonar_slam_amd import _lib
    return _lib.default_context()


@pytest.fixture(scope="session")
def shipped_cfar():
    from sonar_slam_amd.CFAR import CFAR
    return CFAR(40, 10, 0.1, 10)  # bruce_slam/config/feature.yaml:3-7
