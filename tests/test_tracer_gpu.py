"""GPU parity of the tracer transport fused into the colour-gradient kernel (BASELINE config 4)
against the coupled oracle (flow oracle pinned by the reference driver; tracer kernels pinned one
by one by the reference kernels; coupling order pinned by captures of the real, repaired driver:
tests/test_tr_coupled.py)."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _case(porous):
    from openlbmpm_amd.geometry import simple_geometry, porous_disks, image_domain
    if porous:
        dom = image_domain(porous_disks(140, 70, porosity=0.7, rmin=3.0, rmax=8.0, seed=5), 8, 0.5)
    else:
        dom = simple_geometry(24, 56)
    ny, nx = dom.shape
    ii = np.mgrid[0:ny, 0:nx][0]
    fluid = dom == 1
    top = ii >= ny - 12
    rR = np.where(fluid & top, 1.0, 0.0); rB = np.where(fluid & ~top, 1.0, 0.0)
    c0 = np.where(fluid & ~top, 0.3 + 0.2 * np.sin(ii / 3.0), 0.0)
    c1 = np.where(fluid & ~top, 0.1, 0.0)
    return dom, rR, rB, np.stack([c0, c1])


@pytest.mark.parametrize("porous", [False, True], ids=["capillary", "porous"])
def test_two_tracers_vs_coupled_oracle(porous):
    from openlbmpm_amd.rk2d import RK2DSolver
    from oracle.tr import CoupledOracle
    dom, rR, rB, conc = _case(porous)
    flow = dict(theta=70.0, tauR=1.0, tauB=0.8)
    tr = dict(diffX=(1. / 6., 0.12), diffY=(1. / 6., 0.2), dXY=0.01, dYX=0.02, beta=(1.0, 0.6), crit=0.5,
              inlet_conc=(1.0, 0.25), free_outlet=True, dirichlet_inlet=True)
    s = RK2DSolver(dom, flow, diagnostics=True)
    s.set_macro(rR, rB)
    s.configure_tracers(**tr)
    for k in range(2):
        s.set_tracer(k, conc[k])
    o = CoupledOracle(dom, flow, rR, rB, conc, tr)
    for n in (1, 2, 47):
        s.step(n); o.run(n)
        for k in range(2):
            e = rel_err(s.get_tracer(k, compact=True), o.C[k])
            assert e < 1e-9, "tracer %d rel err %.3e after %d steps" % (k, e, s.steps_done)
        for f in ("rhoR", "vx", "phi"):
            assert rel_err(s.get_compact(f), getattr(o.flow, f)) < 1e-9, f
    s.close()


def test_three_tracers_with_reaction_vs_coupled_oracle():
    """[SystemType] Reaction = 'yes': A + B -> C between tracers 0, 1, 2 (calReactionTracersGPU)"""
    from openlbmpm_amd.rk2d import RK2DSolver
    from oracle.tr import CoupledOracle
    dom, rR, rB, conc = _case(True)
    conc = np.stack([conc[0], conc[1] * 3.0, np.zeros_like(conc[0])])
    flow = dict(theta=70.0, tauR=1.0, tauB=0.8)
    tr = dict(diffX=(1. / 6., 0.12, 0.15), diffY=(1. / 6., 0.2, 0.15), dXY=0.01, dYX=0.02, beta=(1.0, 0.6, 0.8), crit=0.5,
              inlet_conc=(1.0, 0.25, 0.0), free_outlet=True, dirichlet_inlet=True, reaction_rate=0.05,
              diffJ=(1. / 3., 0.3, 0.4))
    s = RK2DSolver(dom, flow)
    s.set_macro(rR, rB)
    s.configure_tracers(**tr)
    for k in range(3):
        s.set_tracer(k, conc[k])
    o = CoupledOracle(dom, flow, rR, rB, conc, tr)
    for n in (1, 2, 40):
        s.step(n); o.run(n)
        for k in range(3):
            e = rel_err(s.get_tracer(k, compact=True), o.C[k])
            assert e < 1e-9, "tracer %d rel err %.3e after %d steps" % (k, e, s.steps_done)
    assert o.C[2].max() > 1e-4          # the product tracer started from zero
    with pytest.raises(Exception):
        s2 = RK2DSolver(dom, flow)
        s2.configure_tracers(reaction_rate=0.1)      # one tracer cannot react
    s.close()


def test_tracer_mass_conserved_without_open_boundaries():
    from openlbmpm_amd.rk2d import RK2DSolver
    dom, rR, rB, conc = _case(True)
    s = RK2DSolver(dom, None)
    s.set_macro(rR, rB)
    s.configure_tracers(free_outlet=False, dirichlet_inlet=False)
    s.set_tracer(0, conc[0])
    m0 = conc[0].sum()
    s.step(200)
    c = s.get_tracer(0)
    assert np.isfinite(c).all() and abs(c.sum() - m0) / m0 < 1e-11
    s.close()


@pytest.mark.parametrize("ny,nx", [(57, 72), (58, 72), (60, 72), (61, 72), (44, 72), (141, 2100)], ids=lambda v: str(v))
def test_tracer_step_on_lattices_whose_height_is_no_multiple_of_the_tile(ny, nx):
    """The tracer step runs as up to three launches: the variant that re-sums the densities after the boundary rows (the transport
    driver's order, Transport2DRK.py:1199-1287) on every tile row whose region -- 8 own rows + 3 rows of halo -- holds one of the rows
    0, 1, ny-2, ny-1, the plain variant in between.  With ny % 8 in 1..4 row ny-2 lies among the own rows or in the halo of the
    last-but-one tile row (round 3 gave it the plain variant: the stored densities of the inlet row then depended on the tile that
    computed them).  Stored densities (rec_*: the streamed, boundary-corrected state), phase field, velocity and concentration
    against the coupled oracle after every step of a short run, pressure inlet (where the two orders differ) and pressure outlet."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry
    from oracle.tr import CoupledOracle
    # (2100 wide: 33 tiles per row, where the tiles are walked in staggered bands -- xcd_tile, d2q9_device.h)
    dom = simple_geometry(nx, ny)
    nyy, nx = dom.shape
    ii, jj = np.mgrid[0:nyy, 0:nx]
    fluid = dom == 1
    top = ii >= nyy - 12
    rR = np.where(fluid & top, 1.0, 0.0) * (1.0 + 0.01 * np.sin(jj / 5.0)); rB = np.where(fluid & ~top, 1.0, 0.0) * (1.0 + 0.01 * np.cos(jj / 7.0))
    conc = np.where(fluid & ~top, 0.3 + 0.2 * np.sin(ii / 3.0), 0.0)[None]
    flow = dict(theta=70.0, tauR=1.0, tauB=0.8, inlet="Dirichlet", rhoRH=1.02)
    tr = dict(diffX=(1. / 6.,), diffY=(1. / 6.,), beta=(1.0,), crit=0.5, inlet_conc=(1.0,), free_outlet=True, dirichlet_inlet=True)
    s = RK2DSolver(dom, flow, diagnostics=True)
    s.set_macro(rR, rB)
    s.configure_tracers(**tr)
    s.set_tracer(0, conc[0])
    o = CoupledOracle(dom, flow, rR, rB, conc, tr)
    for n in range(12):
        s.step(1); o.run(1)
        assert rel_err(s.get_tracer(0, compact=True), o.C[0]) < 1e-9, (ny, n)
        for f in ("rhoR", "rhoB", "vx", "vy", "phi"):
            assert rel_err(s.get_compact(f), getattr(o.flow, f)) < 1e-9, (ny, n, f)
    s.close()
