import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_sessionstart(session):
    """Build liberlamsa_hip.so (hipcc cross-compiles for gfx950 without a GPU) and the oracle if a fresh
    checkout has not been built yet; the product package itself never builds or falls back."""
    lib = os.path.join(ROOT, "erlamsa_amd", "liberlamsa_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
