import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build_library():
    """Build (incrementally) libsafereach.so; hipcc cross-compiles without a GPU.  _build.py is loaded by path
    because importing the package fails (by design) while the library is missing."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_sr_build", os.path.join(ROOT, "safe_exploration_amd", "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    if not os.path.exists(os.path.join(ROOT, "safe_exploration_amd", "libsafereach.so")):
        _build_library()            # fresh checkout: the .so is git-ignored


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def lib_built():
    """Build (incrementally) the HIP library once per session; hipcc cross-compiles without a GPU."""
    return _build_library()
