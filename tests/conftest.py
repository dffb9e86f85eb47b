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
def tuning():
    """Setter of the library's tuning overrides (lanes per problem, resident waves, LDS use); automatic again afterwards."""
    from ilqr_amd import capi

    def set_(**kw):
        capi.set_tuning(**kw)

    yield set_
    capi.set_tuning()


def pytest_terminal_summary(terminalreporter):
    import golden_util as gu
    if gu.PINNED_LOG:
        terminalreporter.write_line("solves covered by the 1e-6 parity assertions (pinned / total):")
        for label, n, tot in gu.PINNED_LOG:
            terminalreporter.write_line(f"  {label:60s} {n:5d} / {tot:<5d} {n / max(1, tot):.3f}")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
