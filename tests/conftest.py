import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    # the CPU oracle (checker) and libesim.so (product) are built in-tree; both travel to the GPU box
    if not os.path.exists(os.path.join(ROOT, "oracle", "libesim_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "epidemicsimulator_amd", "libesim.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "epidemicsimulator_amd", "csrc")])


_build_once()


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
