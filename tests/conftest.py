import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """POEM_NATIVE_BT=<path of tools/lab/segv_bt.so>: native backtrace on SIGSEGV / SIGABRT (there is no gdb on the GPU
    boxes; Python's faulthandler, which pytest enables, shows the Python frames only).  Loaded here -- after pytest's own
    faulthandler -- so that its handlers are the ones in place while the tests run."""
    so = os.environ.get("POEM_NATIVE_BT")
    if so:
        import ctypes
        ctypes.CDLL(so)


def pytest_collection_modifyitems(config, items):
    """``pytest tests`` on a box without a GPU skips the ``gpu``-marked tests instead of failing in them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
