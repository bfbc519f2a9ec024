import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The library normally travels prebuilt (python __graft_entry__.py); a checkout without it is built once here.
    The product itself never builds on import — it fails loudly (LrpLibraryMissing)."""
    try:
        from lrp_imagecaptioning_amd import _capi
        if not os.path.exists(_capi.LIB_PATH):
            from lrp_imagecaptioning_amd.build import build_library
            build_library()
    except Exception as e:                                  # (no hipcc: the tests that need the library say so)
        sys.stderr.write("conftest: could not build liblrp_hip.so: %s\n" % e)


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def rel_l1(a, b):
    """sum|a-b| / sum|b| — the parity metric of BASELINE.json / SURVEY.md §8d."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).sum()
    return float(np.abs(a - b).sum() / den) if den else float(np.abs(a - b).sum())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
