import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_data")
MODELS = os.path.join(ROOT, "tests", "golden", "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_pieces_built():
    """The tests load in-tree native artefacts (git-ignored: HIP library, host mirror and its test mains, the
    plain-C oracle).  On a fresh checkout build them once (hipcc cross-compiles without a GPU)."""
    needed = [os.path.join(ROOT, "historian_amd", "lib", "libhistorian_hip.so"),
              os.path.join(ROOT, "historian_amd", "lib", "libhistorian_host.so"),
              os.path.join(ROOT, "historian_amd", "bin", "testmerge"),
              os.path.join(ROOT, "historian_amd", "bin", "hxrecon"),
              os.path.join(ROOT, "oracle", "_build", "liboracle_fill.so"),
              os.path.join(ROOT, "oracle", "_build", "liboracle_fill_map.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def models_dir():
    return MODELS
