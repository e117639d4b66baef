import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tc():
    """One TaskContext (device context + compiled-operator cache) for the GPU session."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import arrow_ballista_amd as g
    return g.TaskContext(device=0)


@pytest.fixture
def mirror_layer(monkeypatch):
    """`node.execute()` through the Python restatement of the executor instead of the native one (plan.py: GPUQ_PLAN_LAYER): for tests
    about the mirror itself -- its refusals, its late-materialised views (`.via`), its per-node metrics."""
    monkeypatch.setenv("GPUQ_PLAN_LAYER", "mirror")
