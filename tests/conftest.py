import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load


def experiments_build():
    """The library under test was built with `make EXP=1`: the A/B switches of csrc/common.h (tune_int / tune_set) read the
    environment.  The product build compiles them to their defaults; tests of an alternative launch shape skip there."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import _capi
    return bool(_capi.lib().scanerf_experiments_enabled())


def need_experiments(what):
    if not experiments_build():
        pytest.skip(f"{what}: only in a library built with `make EXP=1` (the product build has no environment switches)")


def need_symbol(name):
    """Skip a test whose entry point is one of the OPTIONAL test / measurement symbols a lean build may not export."""
    import scanerf_amd  # noqa: F401
    from scanerf_amd import _capi
    if not _capi.has_symbol(name):
        pytest.skip(f"this build of libscanerf_hip.so does not export {name} (optional test entry point)")
