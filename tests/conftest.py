import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests skip (instead of failing at launch) where there is no GPU."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """Build and load libnsff_hip.so.  `make` is incremental, so an edited .hip source is never tested against a
    stale library; where there is no compiler (the GPU box could lack one) the shipped library is used as is."""
    import shutil
    import __graft_entry__ as entry
    from nsff_pl_amd import _lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(_lib.LIB_PATH) or (shutil.which("make") and os.path.exists(hipcc)):
        entry.build()
    return _lib.load()
