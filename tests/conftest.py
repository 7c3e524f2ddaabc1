import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def plugin():
    """A live libbgs context on cuda:0. GPU tests only — fails loudly (no fallback) when the
    HIP extension or the device is missing."""
    from bevy_gaussian_splatting_amd import GaussianSplattingPlugin

    p = GaussianSplattingPlugin(0)
    yield p
    p.close()
