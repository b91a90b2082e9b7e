import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The placement search of the wide weights changes where arrays live, never what they hold: the suite runs it
# shallow (2 -> at most 10 candidate arrays per weight) to keep the full-size tests' set-up short; the dedicated test
# (test_gpu_fused_fwd.py::test_placement_probe_...) and bench.py under test remove / override this.
os.environ.setdefault("FX_PLACEMENT_TRIES", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _graphs_die_between_tests(request):
    """hipGraph lifetime: objects of a finished GPU test that own captured graphs (models with level-1 tape
    graphs, plans) are collected HERE, with the device idle, not by the collector run torch makes on entry to the next test's capture."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
