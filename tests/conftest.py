import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The C ABI and the host mirror default to the fastest policy with the reference's best paths (HX_LSE_TRUNC).  The parity
# tests are about bit-identity first: processes they start (bin/hxrecon, the test mains) run the bit-exact policy unless a
# test names another one (HX_FILL_MODE in the environment it passes).
os.environ.setdefault("HX_FILL_MODE", "exact")

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
