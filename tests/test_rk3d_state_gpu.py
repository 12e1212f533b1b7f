"""State in and out of the D3Q19 solver (include/lbmpm.h "State in and out"; SURVEY 8 a17 / 8b: "field ids cover fR, fB ... in the
reference's host order", section 5: "exact restart (f arrays)").  The reference's 2-D semantics carried to 3-D: start from densities
+ velocity (RKD2Q9.py:577-601), restart from recorded populations (RKD2Q9.py:491-559), record the populations (RKD2Q9.py:938-957).
  * set_macro without a velocity is set_density bit for bit; with one it equals the oracle started from the same equilibria,
  * get_pdf, streamed, is the oracle's population array,
  * get_pdf -> set_pdf continues a run to rounding, also from the 23-value into a 38-value storage,
  * get_state -> set_state in a new context -- one slab or three -- continues it BIT FOR BIT."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

FIELDS = ("rhoR", "rhoB", "phi", "vx", "vy", "vz")
CX = np.array([0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0])
CY = np.array([0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1])
CZ = np.array([0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1])
OPP = np.array([0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17])
W = np.array([1. / 3.] + [1. / 18.] * 6 + [1. / 36.] * 12)

LAYOUTS = {"q23": {}, "dense": {"LBMPM_RK3D_LAYOUT": "dense"}, "compact38": {"LBMPM_RK3D_STORAGE": "38"}}
KERNEL = {"q23": "rk3dq_fused", "dense": "rk3d_fused", "compact38": "rk3dc_fused"}


def _case(nx, ny=17, nz=34, walls=False, seed=None):
    from openlbmpm_amd.geometry import porous_spheres, initial_densities_rk3d
    dom = porous_spheres(nx, ny, nz, porosity=0.72, rmin=2.0, rmax=5.0, seed=nx + ny if seed is None else seed, nbuf=4, walls=walls)
    rR, rB = initial_densities_rk3d(dom, 4)
    return dom, rR, rB


def _env(monkeypatch, layout):
    for k in ("LBMPM_RK3D_LAYOUT", "LBMPM_RK3D_STORAGE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in LAYOUTS[layout].items():
        monkeypatch.setenv(k, v)


def _observe(c):
    c.observe()
    return {f: c.get(f) for f in FIELDS}


def equilibrium(rho, vx, vy, vz):
    """RKD2Q9.py:577-601 with the D3Q19 tables: rho w (1 + (3 eu + 4.5 eu^2 - 1.5 u^2))"""
    eu = CX * vx[..., None] + CY * vy[..., None] + CZ * vz[..., None]
    usq = (vx * vx + vy * vy + vz * vz)[..., None]
    return (rho[..., None] * W) * (1. + (3. * eu + 4.5 * eu * eu - 1.5 * usq))


def stream(f, dom):
    """pull streaming with half-way bounce-back, periodic in x and y, closed in z: what a time step makes of stored populations"""
    fluid = dom == 1
    out = np.zeros_like(f)
    for i in range(19):
        src = np.roll(f[..., i], (CZ[i], CY[i], CX[i]), axis=(0, 1, 2))
        ok = np.roll(fluid, (CZ[i], CY[i], CX[i]), axis=(0, 1, 2))
        if CZ[i] == 1:
            ok[0] = False
        if CZ[i] == -1:
            ok[-1] = False
        out[..., i] = np.where(ok, src, f[..., OPP[i]])
    return np.where(fluid[..., None], out, 0.0)


@pytest.mark.parametrize("layout,nx", [("q23", 64), ("q23", 40), ("dense", 40), ("compact38", 64)])
def test_set_macro_without_a_velocity_is_set_density_bit_for_bit(layout, nx, monkeypatch):
    from openlbmpm_amd.rk3d import RK3DCluster
    _env(monkeypatch, layout)
    dom, rR, rB = _case(nx)
    out = []
    for how in ("density", "macro"):
        c = RK3DCluster(dom, 1, dict(relax="MRT"))
        assert c.slabs[0].dominant_kernel == KERNEL[layout]
        (c.set_density if how == "density" else c.set_macro)(rR, rB)
        st0 = c.get_state()[0]
        c.step(6)
        out.append((st0, c.get_state()[0], _observe(c)))
        c.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for f in FIELDS:
        assert np.array_equal(out[0][2][f], out[1][2][f]), f


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("layout,nx", [("q23", 64), ("q23", 100), ("dense", 40), ("compact38", 64)])
def test_set_macro_with_a_velocity_against_the_oracle(layout, nx, relax, monkeypatch):
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    _env(monkeypatch, layout)
    dom, _, _ = _case(nx)
    rng = np.random.default_rng(5)
    fluid = dom == 1
    w = 0.15 + 0.7 * rng.random(dom.shape)
    rR, rB = np.where(fluid, w, 0.0), np.where(fluid, 1.1 - w, 0.0)
    v = [np.where(fluid, 0.02 * (rng.random(dom.shape) - 0.5), 0.0) for _ in range(3)]
    par = dict(relax=relax, tauR=0.9, tauB=1.1)
    c = RK3DCluster(dom, 2, par)
    c.set_macro(rR, rB, *v)
    o = RK3DOracle(dom, rR, rB, par).set_populations(equilibrium(rR, *v), equilibrium(rB, *v))
    for n in (0, 1, 8):
        c.step(n); o.run(n)
        got = _observe(c); o.macro()
        umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
        for f in FIELDS:
            assert rel_err(got[f], o.field(f), scale=umax if f[0] == "v" else None) < 1e-10, (f, n)
    c.close()


@pytest.mark.parametrize("layout,nx", [("q23", 64), ("q23", 37), ("dense", 40), ("compact38", 64)])
def test_get_pdf_streamed_is_the_oracles_population_array(layout, nx, monkeypatch):
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    _env(monkeypatch, layout)
    dom, rR, rB = _case(nx)
    par = dict(relax="MRT", tauR=0.9, tauB=1.1)
    c = RK3DCluster(dom, 1, par)
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    fR, fB = c.get_pdf()            # before the first step: the rest state itself
    assert not c.slabs[0].post_collision
    assert rel_err(fR, o.field("fR")) < 1e-15 and rel_err(fB, o.field("fB")) < 1e-15
    c.step(12); o.run(12)
    assert c.slabs[0].post_collision and c.slabs[0].state_info()["steps"] == 12
    fR, fB = c.get_pdf()
    scale = float(np.max(np.abs(o.field("fR") + o.field("fB"))))
    # the ghost planes z = 0, nz-1 are not part of the comparison, nor is what streams out of them into the planes next to them (the
    # Zou-He closures of the next step overwrite exactly those directions; the 23-value storage never collides or stores a ghost plane)
    keep = np.ones(dom.shape + (19,), dtype=bool)
    keep[0] = keep[-1] = False
    keep[1][..., CZ == 1] = False
    keep[-2][..., CZ == -1] = False
    for got, name in ((fR, "fR"), (fB, "fB")):
        assert rel_err(np.where(keep, stream(got, dom), 0.0), np.where(keep, o.field(name), 0.0), scale=scale) < 1e-10, name
    c.close()


@pytest.mark.parametrize("src,dst,nx", [("q23", "q23", 100), ("q23", "dense", 100), ("dense", "q23", 40), ("q23", "compact38", 64)])
def test_a_run_continues_from_its_recorded_populations(src, dst, nx, monkeypatch):
    """get_pdf after 10 steps -> set_pdf in a new context (also of another storage) -> 10 more steps == 20 steps, to rounding; and the
    colour swap of the reference's cycle restart (RKD2Q9.py:540-556) keeps the populations a state of the model"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom, rR, rB = _case(nx)
    par = dict(relax="MRT", tauR=0.9, tauB=0.8)
    _env(monkeypatch, src)
    a = RK3DCluster(dom, 1, par)
    a.set_density(rR, rB)
    a.step(10)
    fR, fB = a.get_pdf()
    a.step(10)
    want = _observe(a)
    a.close()
    _env(monkeypatch, dst)
    b = RK3DCluster(dom, 2, par)
    assert b.slabs[0].dominant_kernel == KERNEL[dst]
    b.set_pdf(fR, fB)
    gR, gB = b.get_pdf()
    scale = float(np.max(fR + fB))
    assert rel_err(gR, fR, scale=scale) < 2e-15 and rel_err(gB, fB, scale=scale) < 2e-15
    b.step(10)
    got = _observe(b)
    umax = max(float(np.max(np.abs(want[f]))) for f in ("vx", "vy", "vz"))
    for f in FIELDS:
        assert rel_err(got[f], want[f], scale=umax if f[0] == "v" else None) < 1e-11, f
    # colours swapped in the top planes (the buffer rows of the reference's drainage-imbibition cycle)
    sR, sB = fR.copy(), fB.copy()
    sR[-6:], sB[-6:] = fB[-6:], fR[-6:]
    b.set_pdf(sR, sB)
    gR, gB = b.get_pdf()
    assert rel_err(gR, sR, scale=scale) < 2e-15 and rel_err(gB, sB, scale=scale) < 2e-15
    b.close()


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("layout,nx", [("q23", 64), ("q23", 100), ("q23", 32), ("dense", 40), ("compact38", 64)])
def test_a_restart_from_the_stored_state_is_bit_exact(layout, nx, relax, monkeypatch):
    """run 40 == run 20 -> get_state -> NEW context (one slab and three) -> set_state -> run 20, bit for bit: every field and the
    stored state itself; the front crosses row segments that change between flagged and mixed meanwhile"""
    from openlbmpm_amd.rk3d import RK3DCluster
    _env(monkeypatch, layout)
    dom, rR, rB = _case(nx, walls=(nx == 100))
    par = dict(relax=relax, tauR=0.9, tauB=0.8, velocityZB=-2.0e-3)
    a = RK3DCluster(dom, 1, par)
    a.set_density(rR, rB)
    st0, info0 = a.get_state()
    assert info0 == dict(doubles_per_cell=23 if layout == "q23" else 38, steps=0, post_collision=False)
    a.step(20)
    st, info = a.get_state()
    assert info["steps"] == 20 and info["post_collision"] and st.shape == dom.shape + (info["doubles_per_cell"],)
    assert np.all(st[dom == 0] == 0.0)
    a.step(20)
    want, want_state = _observe(a), a.get_state()[0]
    a.close()
    for k in (1, 3):
        b = RK3DCluster(dom, k, par)
        b.set_state(st, info["steps"], info["post_collision"])
        assert np.array_equal(b.get_state()[0], st)
        b.step(20)
        assert b.slabs[0].steps_done == 40
        got = _observe(b)
        for f in FIELDS:
            assert np.array_equal(got[f], want[f]), (k, f)
        assert np.array_equal(b.get_state()[0], want_state), k
        # and from the state before the first step
        b.set_state(st0, 0, False)
        b.step(40)
        got = _observe(b)
        for f in FIELDS:
            assert np.array_equal(got[f], want[f]), (k, f, "from step 0")
        b.close()


def test_a_state_of_the_other_storage_is_refused(monkeypatch):
    from openlbmpm_amd._lib import LbmpmError, ERR_INVALID
    from openlbmpm_amd.rk3d import RK3DSlab
    dom, rR, rB = _case(64)
    s = RK3DSlab(dom, 0, dom.shape[0])
    with pytest.raises(LbmpmError) as e:
        s.set_state(np.zeros(dom.shape + (38,)))
    assert e.value.status == ERR_INVALID and "23" in str(e.value)
    with pytest.raises(TypeError):
        s.set_pdf(np.zeros(dom.shape + (9,)), np.zeros(dom.shape + (9,)))
    s.close()
