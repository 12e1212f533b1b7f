"""Self-consistency of the D3Q19 colour-gradient oracle (oracle/rk3d_oracle.c).  There is no
reference code for this model; its pin is the reduction to the reference's 2-D loop in
tests/test_rk3d_reduction.py.  Here the oracle is additionally held to physics:
lattice symmetry, colour/mass bookkeeping, and a flat interface at rest staying at rest."""
import numpy as np
import pytest

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


def test_mrt_basis_is_orthogonal_and_reduces_to_bgk():
    """D3Q19 moment basis of the oracle's MRT option: rows mutually orthogonal with the norms of
    d'Humieres et al. 2002; with all rates equal the operator is omega * identity; the conserved
    moments (rho, j) of any input are annihilated by the default rates."""
    import ctypes as C
    from oracle import lib
    L = lib()
    P = C.POINTER(C.c_double)
    M = np.zeros((19, 19))
    L.rk3d_mrt_basis_public(M.ctypes.data_as(P))
    G = M @ M.T
    assert np.array_equal(np.diag(G), [19, 2394, 252, 10, 40, 10, 40, 10, 40, 36, 72, 12, 24, 4, 4, 4, 8, 8, 8])
    assert np.count_nonzero(G - np.diag(np.diag(G))) == 0
    rng = np.random.default_rng(3)
    d = rng.standard_normal(19)
    out = d.copy()
    S = np.full(19, 0.8)
    L.rk3d_mrt_relax_public(S.ctypes.data_as(P), out.ctypes.data_as(P))
    assert np.allclose(out, 0.8 * d, rtol=0, atol=1e-14)
    S = np.array([0., 1.19, 1.4, 0., 1.2, 0., 1.2, 0., 1.2, 0.9, 1.4, 0.9, 1.4, 0.9, 0.9, 0.9, 1.98, 1.98, 1.98])
    out = d.copy()
    L.rk3d_mrt_relax_public(S.ctypes.data_as(P), out.ctypes.data_as(P))
    assert np.abs(M[[0, 3, 5, 7]] @ out).max() < 1e-13


def test_mrt_run_conserves_mass_and_keeps_the_flat_interface():
    from oracle.rk3d import RK3DOracle
    nz, ny, nx = 24, 6, 6
    dom = np.ones((nz, ny, nx), dtype=np.uint8)
    zz = np.arange(nz)[:, None, None] * np.ones((1, ny, nx))
    rR = np.where(zz < 12, 1.0, 1e-8); rB = np.where(zz < 12, 1e-8, 1.0)
    o = RK3DOracle(dom, rR, rB, dict(relax="MRT", velocityZB=0.0)).run(60).macro()
    phi = o.field("phi")
    assert np.abs(phi - phi[:, :1, :1]).max() < 1e-12                 # stays z-dependent only
    assert np.abs(o.field("vx")).max() < 1e-14 and np.abs(o.field("vy")).max() < 1e-14


def test_mrt_rates_static_droplet_stays_put_where_s_m_198_does_not():
    """Why the MRT option does not use d'Humieres' s_m = 1.98: next to the Zou-He planes a mode
    grows until a static droplet is destroyed; with s_m = s_q = 1.2 the spurious currents settle
    below those of BGK."""
    import ctypes as C
    from oracle import lib
    from oracle.rk3d import RK3DOracle
    L = lib()
    n = 24
    dom = np.ones((n, n, n), dtype=np.uint8)
    z, y, x = np.mgrid[0:n, 0:n, 0:n]
    ins = (x - 11.5) ** 2 + (y - 11.5) ** 2 + (z - 11.5) ** 2 < 36
    rR = np.where(ins, 1.0, 1e-8); rB = np.where(ins, 1e-8, 1.0)

    def umax(rates):
        a = np.array(rates, dtype=np.float64)
        L.rk3d_set_mrt_rates_public(a.ctypes.data_as(C.POINTER(C.c_double)))
        try:
            o = RK3DOracle(dom, rR, rB, dict(relax="MRT", velocityZB=0.0)).run(1600).macro()
        finally:
            d = np.array([1.19, 1.4, 1.2, 1.4, 1.2])
            L.rk3d_set_mrt_rates_public(d.ctypes.data_as(C.POINTER(C.c_double)))
        v = np.sqrt(o.field("vx") ** 2 + o.field("vy") ** 2 + o.field("vz") ** 2)
        return float(np.max(v)) if np.isfinite(v).all() else np.inf

    assert umax([1.19, 1.4, 1.2, 1.4, 1.2]) < 1e-4
    assert umax([1.19, 1.4, 1.2, 1.4, 1.98]) > 5e-4


def test_mrt_without_the_matrix_products_is_the_same_operator():
    """the timed CPU baseline relaxes with four projections + the odd/even split instead of two 19 x 19 products"""
    import ctypes as C
    from oracle import lib
    L = lib()
    rng = np.random.default_rng(5)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    cx = np.array([0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0.])
    cy = np.array([0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1.])
    cz = np.array([0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1.])
    A = np.stack([np.ones(19), cx, cy, cz])
    for inv_tau in (0.6, 1.0, 1.9):
        d = rng.normal(size=19)
        d -= A.T @ np.linalg.solve(A @ A.T, A @ d)          # no mass, no momentum (f - feq)
        a, b = np.zeros(19), np.zeros(19)
        L.rk3d_mrt_relax_both_public(C.c_double(inv_tau), P(np.ascontiguousarray(d)), P(a), P(b))
        assert np.max(np.abs(a - b)) < 1e-14 * max(1.0, np.max(np.abs(a)))


@pytest.mark.parametrize("basis", ["rk3d_mrt_basis_public", "rk3dcsf_mrt_basis_public"])       # the perturbation model's oracle, the 3-D CSF model's
def test_mrt_equilibrium_moments_are_the_published_ones(basis):
    """known answer from the paper the basis is taken from (d'Humieres, Ginzburg, Krafczyk, Lallemand, Luo 2002, appendix A, D3Q19): the
    moments of the second-order equilibrium the loop relaxes towards are  e = -11 rho + 19 j.j / rho,  eps = w_eps rho + w_epsj j.j / rho
    with w_eps = 3, w_epsj = -11/2,  q = -2/3 j,  3 p_xx = (2 jx^2 - jy^2 - jz^2) / rho,  p_ww = (jy^2 - jz^2) / rho,  p_xy = jx jy / rho ...,
    pi = w_xx p with w_xx = -1/2,  m = 0 -- the parameter set for which the paper's model is the BGK equilibrium (order of the rows:
    rho, e, eps, jx, qx, jy, qy, jz, qz, 3pxx, 3pixx, pww, piww, pxy, pyz, pxz, mx, my, mz)"""
    import ctypes as C
    from oracle import lib
    L = lib()
    M = np.zeros((19, 19))
    getattr(L, basis)(M.ctypes.data_as(C.POINTER(C.c_double)))
    cx = np.array([0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0.])
    cy = np.array([0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1.])
    cz = np.array([0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1.])
    assert np.array_equal(M[3], cx) and np.array_equal(M[5], cy) and np.array_equal(M[7], cz)       # the direction order of the library
    w = np.where(cx ** 2 + cy ** 2 + cz ** 2 == 0, 1 / 3, np.where(cx ** 2 + cy ** 2 + cz ** 2 == 1, 1 / 18, 1 / 36))
    rho, ux, uy, uz = 1.37, 0.031, -0.052, 0.017
    eu = cx * ux + cy * uy + cz * uz
    feq = rho * w * (1 + 3 * eu + 4.5 * eu ** 2 - 1.5 * (ux * ux + uy * uy + uz * uz))
    jx, jy, jz = rho * ux, rho * uy, rho * uz
    jj = jx * jx + jy * jy + jz * jz
    pxx3, pww = (2 * jx * jx - jy * jy - jz * jz) / rho, (jy * jy - jz * jz) / rho
    want = [rho, -11 * rho + 19 * jj / rho, 3 * rho - 5.5 * jj / rho, jx, -2 / 3 * jx, jy, -2 / 3 * jy, jz, -2 / 3 * jz,
            pxx3, -0.5 * pxx3, pww, -0.5 * pww, jx * jy / rho, jy * jz / rho, jx * jz / rho, 0.0, 0.0, 0.0]
    assert np.allclose(M @ feq, want, rtol=0, atol=2e-15)
