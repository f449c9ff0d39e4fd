import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    return oracle_lib


@pytest.fixture(scope="session")
def ctx():
    # torch bundles its own HIP runtime: when a test uses both, torch must initialise it first (PyTorch wheels bundle their own HIP runtime; one runtime must serve both)
    import torch

    torch.cuda.is_available()
    import provekit_amd

    c = provekit_amd.Context(0)
    yield c
    c.close()
