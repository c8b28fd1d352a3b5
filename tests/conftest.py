import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


import contextlib


@contextlib.contextmanager
def f32_split(mode):
    """VROD_F32_SPLIT for the handles created inside: "0" = fp32 MFMA pass, "1" = bf16 split
    pass forced, None = the library's default (split while memory allows)."""
    old = os.environ.get("VROD_F32_SPLIT")
    if mode is None:
        os.environ.pop("VROD_F32_SPLIT", None)
    else:
        os.environ["VROD_F32_SPLIT"] = mode
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("VROD_F32_SPLIT", None)
        else:
            os.environ["VROD_F32_SPLIT"] = old
