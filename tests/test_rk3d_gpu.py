"""GPU tests of the D3Q19 colour-gradient solver (C ABI): against the independent CPU statement
oracle/rk3d_oracle.c (PARITY UNPINNED vs the reference, which has no 3-D code), and the
slab-decomposed run (k virtual ranks, halo buffers moved exactly as the RCCL path moves them)
against the single-domain run, bit for bit."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _case(nx=40, ny=21, nz=38, seed=4):
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    dom = porous_spheres(nx, ny, nz, porosity=0.7, rmin=3.0, rmax=7.0, seed=seed, nbuf=5)
    rR, rB = initial_densities_rk3d(dom, 5)
    return dom, rR, rB


def test_single_slab_vs_oracle():
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    dom, rR, rB = _case()
    par = dict(tauR=1.0, tauB=0.8)
    c = RK3DCluster(dom, 1, par)
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    for n in (1, 19):
        c.step(n); o.run(n)
        c.observe(); o.macro()
        for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz"):
            e = rel_err(c.get(f), o.field(f))
            assert e < TOL, "field %s rel err %.3e after %d steps" % (f, e, c.slabs[0].steps_done)
    c.close()


@pytest.mark.parametrize("k", [2, 3, 5])
def test_slabs_equal_single_domain_bitwise(k):
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx=33, ny=18, nz=41, seed=9)
    out = []
    for kk in (1, k):
        c = RK3DCluster(dom, kk)
        c.set_density(rR, rB)
        c.step(15)
        c.observe()
        out.append({f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")})
        c.close()
    for f in out[0]:
        assert np.array_equal(out[0][f], out[1][f]), f


def test_single_slab_convenience_equals_phases():
    from openlbmpm_amd.rk3d import RK3DCluster, RK3DSlab
    dom, rR, rB = _case(nx=24, ny=12, nz=20, seed=2)
    s = RK3DSlab(dom, 0, dom.shape[0])
    s.set_density(rR, rB)
    s.step_single(7)
    s.phase_field(diagnostics=True)
    c = RK3DCluster(dom, 1)
    c.set_density(rR, rB)
    c.step(7); c.observe()
    assert np.array_equal(s.get("phi"), c.get("phi")) and np.array_equal(s.get("vz"), c.get("vz"))
    s.close(); c.close()
