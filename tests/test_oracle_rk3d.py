"""Self-consistency of the D3Q19 colour-gradient oracle (oracle/rk3d_oracle.c).  There is no
reference code for this model (PARITY UNPINNED), so the oracle itself is held to physics:
lattice symmetry, colour/mass bookkeeping, and a flat interface at rest staying at rest."""
import numpy as np

from oracle.rk3d import RK3DOracle


def _box(nz=24, ny=10, nx=12, walls=True):
    dom = np.ones((nz, ny, nx), dtype=np.uint8)
    if walls:
        dom[6:-6, :, 0] = 0; dom[6:-6, :, -1] = 0
    dom[10:14, 3:6, 4:7] = 0          # an obstacle
    zz = np.arange(nz)[:, None, None]
    fluid = dom == 1
    rR = np.where(fluid & (zz < nz - 8), 1.0, 0.0)
    rB = np.where(fluid & (zz >= nz - 8), 1.0, 0.0)
    return dom, rR, rB


def test_xy_transpose_symmetry():
    dom, rR, rB = _box()
    a = RK3DOracle(dom, rR, rB).run(12).macro()
    T = lambda v: np.ascontiguousarray(np.swapaxes(v, 1, 2))
    b = RK3DOracle(T(dom), T(rR), T(rB)).run(12).macro()
    for name in ("rhoR", "rhoB", "phi", "vz"):
        assert np.max(np.abs(a.field(name) - T(b.field(name)))) < 1e-13
    assert np.max(np.abs(a.field("vx") - T(b.field("vy")))) < 1e-13


def test_mass_only_changes_through_open_planes():
    dom, rR, rB = _box()
    p = dict(velocityZB=0.0, velocityZR=0.0, densityBL=1.0, densityRL=1.0e-8)
    o = RK3DOracle(dom, rR, rB, p)
    m0 = (rR + rB)[3:-3].sum()
    o.run(30).macro()
    m1 = (o.field("rhoR") + o.field("rhoB"))[3:-3].sum()
    assert np.isfinite(m1) and abs(m1 - m0) / m0 < 5e-3


def test_flat_interface_stays_flat():
    nz, ny, nx = 32, 6, 6
    dom = np.ones((nz, ny, nx), dtype=np.uint8)
    zz = np.arange(nz)[:, None, None] + np.zeros((nz, ny, nx))
    rR = np.where(zz < 16, 1.0, 0.0); rB = np.where(zz >= 16, 1.0, 0.0)
    o = RK3DOracle(dom, rR, rB, dict(velocityZB=0.0, densityRL=1.0, densityBL=1.0e-8)).run(40).macro()
    phi = o.field("phi")
    assert np.max(np.abs(phi - phi[:, :1, :1])) < 1e-12          # no x/y structure appears
    assert abs(o.field("vx")).max() < 1e-12 and abs(o.field("vy")).max() < 1e-12
    assert phi[4].mean() > 0.9 and phi[-5].mean() < -0.9
