import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """spin-bound retries of the ranks-share-one-GPU tests (tests/test_hip_sharded.py::_spawn) are never silent"""
    mod = sys.modules.get("test_hip_sharded")
    retried = getattr(mod, "RETRIED", None) if mod else None
    if retried:
        terminalreporter.section("exchange retries (shared test GPU)")
        for test_id, msg in retried:
            terminalreporter.line(f"{test_id}: {msg}")
