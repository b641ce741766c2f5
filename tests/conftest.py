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
    """Build the HIP library (hipcc cross-compiles without a GPU) and the CPU oracle when they are missing OR older than
    their sources (both builders compare mtimes and return at once when nothing changed): a test run never validates a
    stale binary after an edit to the kernels or the oracle."""
    import sys
    from cmad_amd import build
    if build.is_stale():            # say so: the rebuild takes minutes and replaces the in-tree library with the CURRENT sources' build
        print("conftest: libcmad_hip.so is missing or older than cmad_amd/csrc / include -- rebuilding it (hipcc, ~4 min)", file=sys.stderr, flush=True)
    build.build()
    import oracle_lib
    oracle_lib.build()
