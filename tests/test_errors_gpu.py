"""Error behaviour of the C ABI on a GPU box: every misuse returns a negative
status with a message in lbmpm_last_error(); nothing aborts, nothing falls back to a CPU path."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rk2d_rejects_bad_configurations_and_shapes():
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry
    dom = simple_geometry(20, 40)
    with pytest.raises((LbmpmError, ValueError, KeyError)):
        RK2DSolver(dom, dict(tauR=0.4))                       # tau <= 1/2
    with pytest.raises((LbmpmError, ValueError, KeyError)):
        RK2DSolver(dom, dict(wetting=7))                      # unknown WettingType
    s = RK2DSolver(dom, None)
    with pytest.raises(TypeError):
        s.set_macro(np.zeros((3, 3)), np.zeros((3, 3)))       # wrong shape
    with pytest.raises(LbmpmError) as e:
        s.get_tracer(0)                                       # no tracer configured
    assert "tracer" in str(e.value)
    s.set_macro(np.where(dom == 1, 1.0, 0.0), np.zeros(dom.shape))
    s.step(2)
    with pytest.raises(LbmpmError):
        s.configure_tracers()
        s.set_tracer(0, np.zeros(dom.shape))                  # tracer state must be set before the first step
    s.close()


def test_lattices_beyond_the_32_bit_plane_offsets_are_refused():
    """the 2-D kernels address a node inside a lattice plane with 32 bits (d2q9_device.h::pull_issue_asm): roundup(nx, 32) * ny must stay
    below 2^28 nodes, and create says so instead of wrapping an offset"""
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.sc2d import SC2DSolver
    dom = np.ones((16400, 16384), dtype=np.uint8)             # 2^28 + 2^18 nodes; nothing is allocated on the device
    with pytest.raises(LbmpmError) as e:
        RK2DSolver(dom, None)
    assert "2^28" in str(e.value)
    with pytest.raises(LbmpmError) as e:
        SC2DSolver(dom, dict(inter="EFS"))
    assert "2^28" in str(e.value)


def test_sc2d_rejects_bad_configurations():
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.sc2d import SC2DSolver
    from openlbmpm_amd.geometry import simple_geometry
    dom = simple_geometry(20, 48)
    with pytest.raises((LbmpmError, ValueError)):
        SC2DSolver(dom, dict(inter="ShanChen", relax="MRT"))  # MRT exists for the explicit forcing scheme only
    with pytest.raises((LbmpmError, ValueError)):
        SC2DSolver(dom, dict(inter="ShanChen", scheme=8))     # iso-8 force belongs to EFS
    with pytest.raises((LbmpmError, ValueError)):
        SC2DSolver(dom, dict(scheme=6))
    s = SC2DSolver(dom, dict(inter="EFS"))
    s.set_density(np.where(dom == 1, 1.0, 0.0), np.where(dom == 1, 0.02, 0.0))
    s.step(1)
    with pytest.raises(LbmpmError) as e:
        s.get("rho0")                                         # EFS densities need the diagnostics switch
    assert "diagnostics" in str(e.value)
    s.close()


def test_rk3d_protocol_misuse_is_reported():
    from openlbmpm_amd import _lib
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk3d import RK3DSlab
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    dom = porous_spheres(64, 16, 40, porosity=0.7, rmin=3.0, rmax=6.0, seed=3, nbuf=5)
    rR, rB = initial_densities_rk3d(dom, 5)
    with pytest.raises(LbmpmError):
        RK3DSlab(dom, 30, 20)                                 # slab outside the lattice
    with pytest.raises(LbmpmError):
        RK3DSlab(dom, 0, 40, params=dict(tauR=0.5))
    with pytest.raises(LbmpmError) as e:
        RK3DSlab(dom, 0, 40, params=dict(densityRL=0.0))      # the outlet rule divides by it
    assert "densityRL" in str(e.value)
    top = RK3DSlab(dom, 20, 20)
    top.set_density(rR[20:], rB[20:])
    with pytest.raises(LbmpmError) as e:
        top.step_single(1)                                    # a slab cannot step alone: it needs its neighbour's halo
    assert "single-slab" in str(e.value)
    top.collide_interior()
    with pytest.raises(LbmpmError):
        top.collide()                                         # the step was opened with collide_interior
    top.collide_boundary()
    with pytest.raises(LbmpmError) as e:
        top.get("rhoR")                                       # densities exist after phase_field(diagnostics=True) only
    assert "phase_field" in str(e.value)
    with pytest.raises(KeyError):
        top.buffer("no_such_buffer")
    ptr, n = C.c_void_p(), C.c_int64(0)
    assert _lib.lib().lbmpm_rk3d_buffer(top._h, 99, C.byref(ptr), C.byref(n)) < 0
    assert b"unknown buffer" in _lib.lib().lbmpm_last_error()
    top.close()


def test_empty_inputs():
    """zero steps are a no-op; lattices without a fluid node or below the minimum size are refused
    with a message instead of launching empty grids"""
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.sc2d import SC2DSolver
    from openlbmpm_amd.rk3d import RK3DSlab
    from openlbmpm_amd.geometry import simple_geometry
    dom = simple_geometry(20, 40)
    rR = np.where(dom == 1, 1.0, 0.0)
    s = RK2DSolver(dom, None)
    s.set_macro(rR, np.zeros(dom.shape))
    before = s.get("fR")
    s.step(0)
    assert s.steps_done == 0 and np.array_equal(s.get("fR"), before)
    with pytest.raises(LbmpmError):
        s.step(-3)
    s.close()
    for make in (lambda d: RK2DSolver(d, None), lambda d: SC2DSolver(d, dict(inter="EFS"))):
        with pytest.raises(LbmpmError) as e:
            make(np.zeros((40, 20), dtype=np.uint8))                 # no fluid node at all
        assert "fluid" in str(e.value)
        with pytest.raises(LbmpmError):
            make(np.ones((4, 3), dtype=np.uint8))                    # smaller than the boundary rows need
    with pytest.raises(LbmpmError) as e:
        RK3DSlab(np.zeros((12, 8, 8), dtype=np.uint8), 0, 12)
    assert "fluid" in str(e.value)
    with pytest.raises(LbmpmError):
        RK3DSlab(np.ones((4, 4, 4), dtype=np.uint8), 0, 4)               # fewer than 8 planes


def test_stream_test_reports_plausible_rates():
    """lbmpm_hbm_stream_test (the measured ceiling quoted by bench.py): sane ordering and magnitude on an
    MI355X (spec 8 TB/s), bad arguments refused"""
    from openlbmpm_amd import _lib
    L = _lib.lib()
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    assert L.lbmpm_hbm_stream_test(0, 1 << 30, 3, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert 2000.0 < a.value < 8000.0 and 2000.0 < b.value < 8000.0 and 2000.0 < c.value < 8000.0
    assert L.lbmpm_hbm_stream_test(0, 16, 3, C.byref(a), C.byref(b), C.byref(c)) < 0
    assert L.lbmpm_hbm_stream_test(0, 1 << 30, 0, C.byref(a), C.byref(b), C.byref(c)) < 0


def test_device_memory_accounting():
    """lbmpm_*_device_bytes: two copies of the populations dominate; the compact 3-D storage holds less than
    the dense one on a porous sample"""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.sc2d import SC2DSolver
    from openlbmpm_amd.rk3d import RK3DSlab
    from openlbmpm_amd.geometry import simple_geometry, porous_spheres
    dom = simple_geometry(64, 96)
    for s in (RK2DSolver(dom, None), SC2DSolver(dom, dict(inter="EFS"))):
        assert 2 * 18 * 8 * dom.size <= s.device_bytes < 8 * 18 * 8 * dom.size
        s.close()
    vox = porous_spheres(64, 32, 40, porosity=0.6, rmin=3.0, rmax=6.0, seed=1, nbuf=4)
    compact = RK3DSlab(vox, 0, 40)
    assert compact.dominant_kernel == "rk3dq_fused"        # 23 doubles per fluid cell and buffer (csrc/rk3dq.h)
    os.environ["LBMPM_RK3D_LAYOUT"] = "dense"
    try:
        dense = RK3DSlab(vox, 0, 40)
    finally:
        del os.environ["LBMPM_RK3D_LAYOUT"]
    assert 2 * 23 * 8 * compact.num_fluid_nodes <= compact.device_bytes < dense.device_bytes
    compact.close(); dense.close()


def test_create_destroy_does_not_leak_device_memory():
    """EFS with an iso-8 force keeps a psi array (16 B per cell) that destroy once forgot"""
    import torch
    from openlbmpm_amd.sc2d import SC2DSolver
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.rk3d import RK3DSlab
    from openlbmpm_amd.geometry import simple_geometry
    dom = simple_geometry(512, 512)
    dom3 = np.ones((12, 16, 64), dtype=np.uint8)

    def cycle():
        s = SC2DSolver(dom, dict(inter="EFS", scheme=8)); s.close()
        s = RK2DSolver(dom, None, diagnostics=True); s.close()
        s = RK3DSlab(dom3, 0, 12); s.close()
    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(10):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 4 << 20, "device memory shrank by %d bytes over 10 create/destroy cycles" % (free0 - free1)


def test_rk3d_fields_are_refused_when_stale():
    from openlbmpm_amd._lib import LbmpmError
    from openlbmpm_amd.rk3d import RK3DSlab
    dom = np.ones((12, 8, 64), dtype=np.uint8)
    s = RK3DSlab(dom, 0, 12)
    s.set_density(np.ones(dom.shape), np.zeros(dom.shape))
    with pytest.raises(LbmpmError):
        s.get("phi")                                          # nothing observed yet
    s.step_single(2)
    with pytest.raises(LbmpmError) as e:
        s.get("phi")                                          # the fused kernel keeps phi in LDS
    assert "stale" in str(e.value)
    s.phase_field(diagnostics=True)
    assert np.allclose(s.get("phi")[4:8], 1.0) and np.allclose(s.get("rhoR")[4:8], 1.0)     # away from the open planes
    s.step_single(1)
    with pytest.raises(LbmpmError):
        s.get("rhoR")
    s.close()
