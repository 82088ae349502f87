import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a HIP device: the gpu-marked tests are skipped, not failed."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP GPU in this process (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "coma_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def hip_lib():
    """The loaded C-ABI library; building it first if the .so is absent (hipcc cross-compiles)."""
    from coma_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_lib(verbose=False)
    return _lib.lib()
