import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles
    without a GPU) and the CPU oracle once, only if they are missing."""
    from cmad_amd import build
    if not os.path.exists(build.LIB):
        build.build()
    import oracle_lib
    if not os.path.exists(os.path.join(ROOT, "oracle", "libcmad_oracle.so")):
        oracle_lib.build(force=True)
