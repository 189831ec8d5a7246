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
    product path has no CPU fallback, so there is nothing for them to run).  "GPU-less" means no
    kernel driver node: on a box WITH a GPU a library that does not load, or that sees no
    device, is an error -- a broken build must not read as 255 skips."""
    if not any("gpu" in item.keywords for item in items):
        return
    if os.path.exists("/dev/kfd"):
        from transport_analysis_amd import _lib  # load / build errors propagate

        if _lib.device_count() >= 1:
            return
        if os.environ.get("TA_REQUIRE_GPU") == "1":
            raise pytest.UsageError("/dev/kfd exists but libta_hip.so sees no HIP device")
    skip = pytest.mark.skip(reason="no usable GPU: the HIP path has no CPU fallback")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
