import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(GOLDEN / (name + ".npz"))

    return load


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """The library reads its DN_* switches once; tests that monkeypatch them call dn_reload_knobs() themselves, and this puts the
    cached copy back in step with the restored environment afterwards (autouse fixtures are torn down after monkeypatch)."""
    yield
    from supervised_dispnet_amd import _lib
    if _lib._lib is not None:
        _lib._lib.dn_reload_knobs()
