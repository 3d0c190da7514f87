import os
import sys

import pytest
import torch  # noqa: F401  (first: torch brings its own HIP runtime; if libplr.so loads the system one before it, torch then finds "no HIP GPUs")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle splits the rows of a pass over std::threads (default 1); its results do not depend on the count (test_oracle_is_thread_count_invariant).
    # Single-threaded, the oracle frames of the full-size parity tests were 3/4 of the GPU suite's 18 minutes.
    import pyoracle
    pyoracle.set_threads(max(1, min(os.cpu_count() or 1, 128)))


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def backend():
    """One RenderBackend per session (plr_setup is a process-wide singleton like gRenderBackend)."""
    from plainrenderer_amd import RenderBackend
    be = RenderBackend(1920, 1080, device=0)
    be.setMathMode(False)  # the parity tests demand bit identity unless they switch to the fast kernels themselves
    yield be
    be.shutdown()
