import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the checker's problems are small: a team of a few threads beats one of every core of a large box
os.environ.setdefault("LBMPM_ORACLE_THREADS", str(min(8, os.cpu_count() or 1)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


DEV_KNOBS = ("LBMPM_RK3D_TILE", "LBMPM_RK3D_CHUNK", "LBMPM_RK3D_FILL", "LBMPM_RK3D_BOUNDARY", "LBMPM_RK3D_XCC", "LBMPM_RK3D_SLAB_SCHEDULE",
             "LBMPM_RK2D_SHAPE", "LBMPM_IPC_LAND")


@pytest.fixture
def knobs(monkeypatch):
    """knobs(env): set environment switches of the library for the rest of the test.  Tuning knobs (DEV_KNOBS) exist in the development
    build only (openlbmpm_amd/build.py::build_dev_if_stale, built by __graft_entry__.build()): contexts created after a call that names
    one come from that library; the product library is back when the test ends."""
    from openlbmpm_amd import _lib, build

    def apply(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if any(k in DEV_KNOBS for k in env):
            _lib.use_library(build.build_dev_if_stale(verbose=False))
    yield apply
    _lib.use_library(None)
