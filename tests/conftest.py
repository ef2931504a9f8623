import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size case")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class _Knobs:
    """Sets the engine's knobs through charls_amd_debug_set_knob and clears every one it touched at the end of the test."""

    def __init__(self):
        self.touched = set()

    def set(self, name, value):
        from charls_amd import capi
        name = name.replace("CHARLS_AMD_", "")
        capi.set_knob(name, value)
        self.touched.add(name)

    def clear(self, name):
        self.set(name, None)

    def restore(self):
        from charls_amd import capi
        for name in self.touched:
            capi.set_knob(name, None)


@pytest.fixture
def knobs():
    k = _Knobs()
    yield k
    k.restore()
