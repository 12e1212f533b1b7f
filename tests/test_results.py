"""Result files: the reference's group / dataset names in a real HDF5 file written through the
HDF5 C library (ctypes) when PyTables / h5py are absent, `.npz` only as the last resort."""
import os

import numpy as np
import pytest

from openlbmpm_amd import _hdf5
from openlbmpm_amd.results import ResultFile, load_results, read_planes, find_result_file

needs_hdf5 = pytest.mark.skipif(not _hdf5.available(), reason="no HDF5 C library on this machine")
GROUPS = (("FluidMacro", "MacroData"), ("FluidVelocity", "MacroVelocity"))


@needs_hdf5
def test_libhdf5_round_trip(tmp_path, monkeypatch):
    monkeypatch.setenv("LBMPM_RESULT_BACKEND", "libhdf5")
    out = ResultFile(str(tmp_path), "SimulationResults", GROUPS)
    assert out.path.endswith(".h5")
    rng = np.random.default_rng(0)
    a = rng.standard_normal((7, 5)); b = rng.standard_normal((3, 4, 9)); c = (rng.random((6, 6)) > 0.5).astype(np.uint8)
    out.write("FluidMacro", "FluidDensityType0in0", a)
    out.write("FluidMacro", "FluidDensityType1in0", b)
    out.write("FluidVelocity", "mask", c)
    out.write("FluidVelocity", "steps", np.arange(5, dtype=np.int64))
    with open(out.path, "rb") as fh:
        assert fh.read(8) == b"\x89HDF\r\n\x1a\n"                     # the HDF5 superblock signature
    back = load_results(out.path)
    assert set(back) == {"/FluidMacro/FluidDensityType0in0", "/FluidMacro/FluidDensityType1in0", "/FluidVelocity/mask",
                         "/FluidVelocity/steps"}
    assert np.array_equal(back["/FluidMacro/FluidDensityType0in0"], a) and back["/FluidMacro/FluidDensityType1in0"].shape == (3, 4, 9)
    assert np.array_equal(back["/FluidMacro/FluidDensityType1in0"], b)
    assert back["/FluidVelocity/mask"].dtype == np.uint8 and np.array_equal(back["/FluidVelocity/mask"], c)
    assert back["/FluidVelocity/steps"].dtype == np.int64


@needs_hdf5
def test_libhdf5_errors(tmp_path, monkeypatch):
    monkeypatch.setenv("LBMPM_RESULT_BACKEND", "libhdf5")
    out = ResultFile(str(tmp_path), "r", GROUPS)
    out.write("FluidMacro", "x", np.zeros(3))
    with pytest.raises(_hdf5.Hdf5Error):
        out.write("FluidMacro", "x", np.zeros(3))                    # exists already
    with pytest.raises(_hdf5.Hdf5Error):
        out.write("NoSuchGroup", "x", np.zeros(3))
    with pytest.raises(_hdf5.Hdf5Error):
        out.write("FluidMacro", "z", np.zeros(3, dtype=np.complex128))
    with pytest.raises(_hdf5.Hdf5Error):
        _hdf5.read_all(os.path.join(str(tmp_path), "missing.h5"))


def test_npz_fallback_and_backend_switch(tmp_path, monkeypatch):
    monkeypatch.setenv("LBMPM_RESULT_BACKEND", "npz")
    out = ResultFile(str(tmp_path), "r", GROUPS)
    out.write("FluidMacro", "x", np.arange(4.0))
    assert out.path.endswith(".npz") and np.array_equal(load_results(out.path)["/FluidMacro/x"], np.arange(4.0))
    monkeypatch.setenv("LBMPM_RESULT_BACKEND", "sqlite")
    with pytest.raises(ValueError):
        ResultFile(str(tmp_path), "r2", GROUPS)


@pytest.mark.parametrize("backend", ["libhdf5", "npz"])
def test_read_planes_takes_a_slab_of_a_record(tmp_path, monkeypatch, backend):
    """a rank of a distributed run reads its own planes of a record ([CyclesSetup] IsCycle = 'yes', checkpoints): a hyperslab read"""
    if backend == "libhdf5" and not _hdf5.available():
        pytest.skip("no HDF5 C library on this machine")
    monkeypatch.setenv("LBMPM_RESULT_BACKEND", backend)
    out = ResultFile(str(tmp_path), "SimulationResultsRK3D", GROUPS)
    rng = np.random.default_rng(2)
    a = rng.standard_normal((11, 4, 6)); b = rng.standard_normal((11, 4, 6, 19)); c = np.arange(7, dtype=np.int64)
    out.write("FluidMacro", "FluidDensityRin3", a); out.write("FluidMacro", "PDF", b); out.write("FluidVelocity", "Info", c)
    assert find_result_file(str(tmp_path), "SimulationResultsRK3D") == out.path and find_result_file(str(tmp_path), "other") is None
    assert np.array_equal(read_planes(out.path, "/FluidMacro/FluidDensityRin3"), a)
    assert np.array_equal(read_planes(out.path, "/FluidMacro/FluidDensityRin3", 3, 5), a[3:8])
    assert np.array_equal(read_planes(out.path, "/FluidMacro/PDF", 9, 2), b[9:])
    assert np.array_equal(read_planes(out.path, "/FluidMacro/PDF", 0, 1), b[:1])
    got = read_planes(out.path, "/FluidVelocity/Info")
    assert got.dtype == np.int64 and np.array_equal(got, c)
    with pytest.raises(KeyError):
        read_planes(out.path, "/FluidMacro/missing")
    if backend == "libhdf5":
        with pytest.raises(_hdf5.Hdf5Error):
            read_planes(out.path, "/FluidMacro/PDF", 10, 2)
