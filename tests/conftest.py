import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def vo():
    """The product package (directory name is not a Python identifier)."""
    return importlib.import_module("visual_odometry_pipeline_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import vo_oracle
    vo_oracle.build()
    return vo_oracle


@pytest.fixture(scope="session")
def ctx(vo):
    """A GPU context; creating it fails loudly when no GPU or no built library is present."""
    c = vo.Context(0)
    yield c
    c.close()
