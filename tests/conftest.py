import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def scale_rel_err(actual, expected):
    """Scale-relative error used throughout:
    max|a - b| / max|b| per series; exact zeros in `expected` are fine."""
    import numpy as np

    a = np.asarray(actual, dtype=np.float64)
    b = np.asarray(expected, dtype=np.float64)
    scale = np.max(np.abs(b))
    if scale == 0.0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a GPU-less host: the gpu-marked tests are skipped, not errors (the
    product path has no CPU fallback, so there is nothing for them to run)."""
    try:
        from transport_analysis_amd import _lib

        have_gpu = _lib.device_count() >= 1
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no usable GPU: the HIP path has no CPU fallback")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
