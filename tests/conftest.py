import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from volumeraytracer_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a device must fail loudly, not skip: only skip gpu tests when they
    # were not asked for explicitly (plain `pytest tests/`).
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU here; run with -m gpu on the MI355X box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
