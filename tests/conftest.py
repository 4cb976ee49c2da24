import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # DAL3_TEST_LIB=<path>: run this test session against ANOTHER BUILD of the library — the host-sanitizer build of
    # `make asan-host` (tools/asan_host.sh). A switch of the test harness only: the product binding (_hip.py) reads no
    # environment variable and always loads the in-tree lib3dal_hip.so.
    alt = os.environ.get("DAL3_TEST_LIB")
    if alt:
        import importlib
        hip = importlib.import_module("3dal_pytorch_amd._hip")
        hip.LIB_PATH = os.path.abspath(alt)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
