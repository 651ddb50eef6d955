import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def zk_ctx():
    """libzkmi context on cuda:0 -- GPU tests only; fails loudly if the HIP library is missing."""
    # torch first: it ships its own libamdhip64.so.7 and whichever HIP runtime a process loads first
    # is the one every later library gets; tests that hand torch device tensors to the C-ABI need
    # both sides on the same runtime (bench.py imports torch before the context for the same reason)
    import torch  # noqa: F401
    from gnark_crypto_primitives_amd import lib
    ctx = lib.Context(0)
    yield ctx
    ctx.close()
