import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG_NAME = '3d-pose-estimation-with-previleged-information_amd'
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    """The product package (its directory name is not a Python identifier, so import it by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope='session')
def synth():
    return importlib.import_module(PKG_NAME + '.synth')


def golden_path(name):
    return os.path.join(GOLDEN, name)
