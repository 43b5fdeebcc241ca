import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


try:  # the reference's hloc (h5py, pycolmap ...) is not installed here: a protocol stand-in for the plugin tests
    import hloc.utils.base_model  # noqa: F401
except Exception:  # noqa: BLE001
    for _m in [m for m in sys.modules if m == "hloc" or m.startswith("hloc.")]:
        del sys.modules[_m]
    sys.path.insert(0, os.path.join(ROOT, "tests", "hloc_stub"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle_sd():
    import loftr_oracle as O
    return O.make_state_dict(seed=0)
