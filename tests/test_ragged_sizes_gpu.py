"""Ragged and extreme lattice sizes through the C ABI against the oracles: widths that are not
multiples of the 64-wide tiles (and of the 32-element pitch), one tile plus one column, the
narrowest domains the solvers accept, heights that leave a partial tile row, 3-D slabs thinner
than a march chunk -- on seeded porous inputs, a few steps each."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

SIZES_2D = [(12, 44), (33, 47), (63, 45), (64, 48), (65, 49), (127, 46), (129, 53), (200, 41)]
# at least 32 tiles wide: the tiles of an XCD's share are walked in staggered bands of four tile rows (xcd_tile, d2q9_device.h) -- whole
# bands, a last band of fewer rows, shares that begin and end inside a tile row, a partial tile column
SIZES_WIDE = [(2048, 72), (2050, 44), (2241, 93)]


def _image(nx, ny, seed, nbuf):
    from openlbmpm_amd.geometry import porous_disks, image_domain
    # discs of radius >= 3: no one-node diagonal chains, where the reference's wetting rule is an exact tie
    # (d1 == d2, A:1665-1673) that rounding decides
    img = porous_disks(nx, ny - 2 * nbuf, porosity=0.75, rmin=3.0, rmax=6.0, seed=seed)
    return image_domain(img, nbuf, 0.5)


@pytest.mark.parametrize("nx,ny", SIZES_2D, ids=["%dx%d" % s for s in SIZES_2D])
def test_rk2d_ragged(nx, ny):
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import initial_densities_rk
    from oracle.rk import RKOracle
    dom = _image(nx, ny, nx + ny, 6)
    rR, rB = initial_densities_rk(dom, True, 6)
    # discs and a field that is uniform in x make mirror-symmetric interfaces, and on a symmetry axis both
    # wetting rules of the reference end in an exact tie (d1 == d2, A:1665-1673, A:2482-2490) that only rounding
    # decides; a ripple in the initial densities removes the ties
    yy, xx = np.mgrid[0:dom.shape[0], 0:dom.shape[1]]
    ripple = 1.0 + 1.0e-3 * np.sin(0.37 * xx + 0.11 * yy)
    rR, rB = rR * ripple, rB * ripple
    par = dict(theta=75.0, tauR=0.9, tauB=1.1, relax="MRT" if nx % 2 else "SRT", wetting=1 + nx % 2)
    s = RK2DSolver(dom, par, diagnostics=True)
    s.set_macro(rR, rB)
    o = RKOracle(dom, par, rR, rB)
    s.step(12); o.run(12)
    for f in ("fR", "fB", "rhoR", "rhoB", "vx", "vy", "phi", "Gx", "Gy", "Fx", "Fy"):
        assert rel_err(s.get_compact(f), getattr(o, f)) < 1e-9, (f, dom.shape)
    # the curvature is a quotient by |G|: compared where the colour gradient is not vanishing (ahead of the
    # interface |G| ~ 1e-9 and K only amplifies rounding; the force it enters is K G)
    live = np.hypot(o.Gx, o.Gy) > 1e-6
    assert rel_err(s.get_compact("K")[live], o.K[live]) < 1e-9, ("K", dom.shape)
    s.close()


@pytest.mark.parametrize("nx,ny,grains", [(2048, 136, 0), (2241, 173, 0), (2241, 173, 7)], ids=["2048x136", "2241x173", "2241x173-grains"])
def test_rk2d_wide(nx, ny, grains):
    """Lattices at least 32 tiles wide and tall enough that every XCD's share holds whole tile rows: the staggered band walk of xcd_tile
    (d2q9_device.h) with full bands, a short last band, shares that start and end inside a tile row and a partial tile column.  A
    capillary between side walls, optionally with a few large grains (a porous image of this size holds thousands of wall nodes, and
    one near-tie of the reference's wetting rule -- see test_rk2d_ragged -- costs more than the 1e-9 asked here)."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    from oracle.rk import RKOracle
    dom = simple_geometry(nx, ny)
    yy, xx = np.mgrid[0:ny, 0:nx]
    for g in range(grains):
        cx, cy, r = 150 + g * 290 + 13 * (g % 3), 40 + (g * 37) % (ny - 80), 9.5 + 1.3 * (g % 4)
        dom[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = 0
    rR, rB = initial_densities_rk(dom, True, 6, mode="intrusion")
    ripple = 1.0 + 1.0e-3 * np.sin(0.37 * xx + 0.11 * yy)
    rR, rB = rR * ripple, rB * ripple
    par = dict(theta=75.0, tauR=0.9, tauB=1.1, relax="MRT", wetting=1 + nx % 2)
    s = RK2DSolver(dom, par, diagnostics=True)
    s.set_macro(rR, rB)
    o = RKOracle(dom, par, rR, rB)
    s.step(10); o.run(10)
    for f in ("fR", "fB", "rhoR", "rhoB", "vx", "vy", "phi", "Gx", "Gy", "Fx", "Fy"):
        assert rel_err(s.get_compact(f), getattr(o, f)) < 1e-9, (f, dom.shape)
    s.close()


@pytest.mark.parametrize("nx,ny", SIZES_2D + SIZES_WIDE, ids=["%dx%d" % s for s in SIZES_2D + SIZES_WIDE])
def test_sc2d_ragged(nx, ny):
    from openlbmpm_amd.sc2d import SC2DSolver
    from oracle.sc import SCOracle, initial_densities
    dom = _image(nx, ny + 30, nx * 3 + ny, 20)
    scheme = (4, 8, 10)[nx % 3]
    par = dict(inter="EFS", relax="MRT" if nx % 2 else "SRT", outlet="Convective" if ny % 2 else "Dirichlet", scheme=scheme)
    dens = dict(rho0=1.0, rho1=1.0, bg0=0.15, bg1=0.15)
    o = SCOracle(dom, dict(par, **dens), image=True)
    rho = initial_densities(dom, True, dict(par, **dens))
    s = SC2DSolver(dom, par, diagnostics=True)
    s.set_density(rho[0], rho[1])
    s.step(10); o.run(10)
    for k in range(2):
        assert rel_err(s.get_compact("f%d" % k), o.f[k]) < 1e-9, (k, dom.shape, scheme)
        assert rel_err(s.get_compact("rho%d" % k), o.rho[k]) < 1e-9
    s.close()


SIZES_3D = [(64, 5, 9), (128, 8, 10), (40, 5, 9), (192, 12, 8), (65, 9, 13), (64, 17, 40), (4, 4, 8)]


@pytest.mark.parametrize("layout", ["q23", "dense"])
@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("nx,ny,nz", SIZES_3D, ids=["%dx%dx%d" % s for s in SIZES_3D])
def test_rk3d_ragged(nx, ny, nz, relax, layout, monkeypatch):
    """every nx runs the compact 23-value storage (row segments of nx / ceil(nx / 64) cells), LBMPM_RK3D_LAYOUT=dense the dense
    layout; thin slabs end inside the first march chunk; 4 x 4 x 8 is the smallest lattice lbmpm_rk3d_create accepts"""
    if layout == "dense":
        monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    from openlbmpm_amd.rk3d import RK3DCluster
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    from oracle.rk3d import RK3DOracle
    nbuf = 2
    if nx >= 40:
        dom = porous_spheres(nx, ny, nz, porosity=0.75, rmin=1.5, rmax=3.5, seed=nx + ny + nz, nbuf=nbuf)
    else:
        dom = np.ones((nz, ny, nx), dtype=np.uint8)
    rR, rB = initial_densities_rk3d(dom, nbuf)
    par = dict(tauR=0.9, tauB=1.2, relax=relax)
    for k in (1, 2):
        if nz < 4 * k:
            continue
        c = RK3DCluster(dom, k, par)
        assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
        c.set_density(rR, rB)
        o = RK3DOracle(dom, rR, rB, par)
        c.step(7); o.run(7)
        c.observe(); o.macro()
        for f in ("rhoR", "rhoB", "phi", "vz") + (("vx", "vy") if nx >= 40 else ()):      # (uniform in x, y: vx = vy = 0 exactly in the oracle)
            assert rel_err(c.get(f), o.field(f)) < 1e-10, (f, dom.shape, k)
        c.close()
