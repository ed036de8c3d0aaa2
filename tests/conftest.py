import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly on a GPU box; on a CPU-only host they are skipped
    unless explicitly selected with -m gpu (then they fail, as they should)."""
    import torch
    if torch.cuda.is_available():
        return
    selected = config.getoption("-m") or ""
    if "gpu" in selected and "not gpu" not in selected:
        return
    skip = pytest.mark.skip(reason="no GPU in this host")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
