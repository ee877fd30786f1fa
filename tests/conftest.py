import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The tracked RaycastSdf chooses between the plain march and the march through the class tables from the share of the volume
# the tables cover (raycast.hip, class_view).  The suite wants the table march exercised wherever a summary is passed, whatever
# that share is: force it (read once, when libkfx.so is first used).  tests/test_gpu_summary.py checks the unforced choice in
# a process of its own.
os.environ.setdefault("KFX_RAYCAST_SUMMARY", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def roo():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kangaroo_amd import roo as r
    from kangaroo_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    return r
