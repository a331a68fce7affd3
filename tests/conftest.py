import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (C restatement, oracle/liboracle.so), built on demand."""
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def engine():
    """An engine context on cuda:0 through the C ABI; gpu tests only."""
    if not _have_gpu():
        pytest.skip("no GPU visible")
    import eccoxide_amd

    eng = eccoxide_amd.Engine(0)
    yield eng
    eng.close()
