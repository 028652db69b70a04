import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def emu():
    """Route rehrseg_amd.ops through the CPU emulation of the C-ABI (test only)."""
    import emu_backend
    from rehrseg_amd import ops
    old = ops.set_backend(emu_backend)
    yield emu_backend
    ops.set_backend(old)
