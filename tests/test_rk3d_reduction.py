"""The pin of the D3Q19 colour-gradient model: reduction to the reference's D2Q9 perturbation loop.

The reference has no 3-D code (main.py:22 imports a missing module), but it has the 2-D loop the 3-D model
extends (runRKColorGradient2DPerturbation, RKD2Q9.py:978-1223) and its kernels.  A D3Q19 lattice that is
uniform along y (periodic, any ny) projects onto D2Q9 in (x, z): summing the populations over e_y gives the
D2Q9 weights 4/9, 1/9, 1/36, the B_i of Liu et al. -1/3, 1/18, 1/36 give the reference's constantBNew
-2/9, 1/9, 1/36 (RKD2Q9.py:131-133), the gradient 3 sum w e phi, BGK, the perturbation, the Zou-He velocity
and pressure closures (A:657, A:1008 <-> Hecht & Harting) and half-way bounce-back all project term by term.
The ONE term that does not is the |e_i| inside cos(theta_i) of the recolouring (A:1241-1267): the diagonal
neighbours (+-1, +-1, 0) of D3Q19 fold onto the D2Q9 axes with weight 1/(36 sqrt 2) instead of 1/36.  So

* with the recolouring weights overridden by their projection-exact values (recolor_axis = 1/9 - 2/(36 sqrt 2);
  the only use of that switch), the 3-D code must reproduce the captures of the REAL 2-D driver
  (tests/golden/rkpert_srt_*.npz) on rhoR, rhoB, phi, u at every snapshot -- inlet/outlet planes, ghost planes
  and solid-phi walls included;
* with the model's own weights it must equal the pinned D2Q9 oracle (oracle/rk_pert_oracle.c, pinned by
  tests/test_oracle_rk_pert.py) run with the correspondingly projected recolouring weights.

CPU tests hold oracle/rk3d_oracle.c to this; the -m gpu tests hold the HIP kernels (rk3dc_fused, rk3d_fused,
split schedule) to it through the C ABI."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_params
from oracle.rk import RKPertOracle
from oracle.rk3d import RK3DOracle

SQ2 = np.sqrt(2.0)
RC_EXACT = dict(recolor_axis=1. / 9. - 2. / (36. * SQ2), recolor_diag=1. / (36. * SQ2))
RW_PROJECTED = np.array([0.] + [1. / 18. + 2. / (36. * SQ2)] * 4 + [1. / 36.] * 4)   # what the model's own weights fold onto
FIELDS = (("rhoR", "rhoR"), ("rhoB", "rhoB"), ("phi", "phi"), ("vx", "vx"), ("vz", "vy"))
TOL = 1e-10


def scenario(name):
    d = np.load(os.path.join(GOLDEN, "rkpert_%s.npz" % name))
    p = load_params(d)
    dom2 = d["isDomain"]
    sp = float(d["solidPhi"])
    par3 = dict(AkR=float(d["AkR"]), AkB=float(d["AkB"]), beta=p["beta"], tauR=p["tauR"], tauB=p["tauB"],
                SolidRhoR=0.5 * (1. + sp), SolidRhoB=0.5 * (1. - sp), velocityZR=p["vyR"], velocityZB=p["vyB"],
                densityRL=p["rhoRL"], densityBL=p["rhoBL"], relax="SRT")
    par2 = dict(beta=p["beta"], AkR=float(d["AkR"]), AkB=float(d["AkB"]), solidPhi=sp, tauR=p["tauR"], tauB=p["tauB"], relax="SRT",
                vyR=p["vyR"], vyB=p["vyB"], rhoBL=p["rhoBL"], rhoRL=p["rhoRL"])
    return d, dom2, par2, par3


def unstream(rho2, o2):
    """populations whose first streaming gives the rest state w_i rho (what the golden captures start from, see
    tests/golden/gen/make_golden_rk_pert.py::unstream): compact [N][9]"""
    w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    rho = rho2.reshape(-1)[o2.fluidNodes]
    nb = o2.nbr.reshape(-1, 8)
    g = np.empty((rho.size, 9))
    g[:, 0] = w[0] * rho
    for i in range(1, 9):
        q = nb[:, i - 1]
        g[:, i] = w[i] * np.where(q >= 0, rho[np.maximum(q, 0)], rho)
    return g


def extrude(a2, ny):
    """[z][x] -> [z][y][x], uniform along y"""
    return np.ascontiguousarray(np.repeat(np.asarray(a2)[:, None, :], ny, axis=1))


def dense2(d, compact):
    ny, nx = d["isDomain"].shape
    out = np.zeros(ny * nx)
    out[d["fluidNodes"]] = compact
    return out.reshape(ny, nx)


def check_against(fields3, ref2, dom2, ny, what):
    """fields3: name -> [z][y][x]; ref2: name -> dense [z][x]"""
    worst = 0.0
    for f3, f2 in FIELDS:
        a, g = fields3[f3], ref2[f2]
        assert np.max(np.abs(a - a[:, :1, :])) <= 1e-13 * max(1.0, np.max(np.abs(a))), (what, f3, "not uniform along y")
        m = dom2 == 1
        scale = max(np.max(np.abs(g[m])), 1e-300)
        err = np.max(np.abs(a[:, 0, :][m] - g[m])) / scale
        worst = max(worst, err)
        assert err < TOL, (what, f3, err)
    if "vy" in fields3:
        assert np.max(np.abs(fields3["vy"])) < 1e-13
    return worst


@pytest.mark.parametrize("name", ["srt_capillary", "srt_porous", "srt_porous64"])
def test_oracle3d_reproduces_the_reference_2d_driver(name):
    d, dom2, par2, par3 = scenario(name)
    ny = 3
    dom3 = extrude(dom2, ny)
    rR = extrude(dense2(d, d["init_rhoR"]), ny); rB = extrude(dense2(d, d["init_rhoB"]), ny)
    o = RK3DOracle(dom3, rR, rB, dict(par3, **RC_EXACT))
    done = 0
    for k in d["snaps"]:
        o.run(int(k) - 1 - done); done = int(k) - 1
        o.macro()
        got = {f: o.field(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}
        ref = {f: dense2(d, d["s%d_%s" % (k, f)]) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
        check_against(got, ref, dom2, ny, (name, int(k)))
    assert done >= 49


@pytest.mark.parametrize("name", ["srt_capillary", "srt_porous"])
def test_oracle3d_default_weights_equal_the_pinned_2d_oracle(name):
    d, dom2, par2, par3 = scenario(name)
    ny, steps = 4, 70
    dom3 = extrude(dom2, ny)
    rR2 = dense2(d, d["init_rhoR"]); rB2 = dense2(d, d["init_rhoB"])
    o3 = RK3DOracle(dom3, extrude(rR2, ny), extrude(rB2, ny), par3).run(steps).macro()
    o2 = RKPertOracle(dom2, par2, rhoR0=rR2, rhoB0=rB2, recolor_weights=RW_PROJECTED)
    o2 = RKPertOracle(dom2, par2, fR0=unstream(rR2, o2), fB0=unstream(rB2, o2), recolor_weights=RW_PROJECTED).run(steps + 1)
    got = {f: o3.field(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}
    ref = {f: o2.dense(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
    check_against(got, ref, dom2, ny, name)
    # and the projection is not vacuous: the reference's own weights give a different answer
    o2w = RKPertOracle(dom2, par2, fR0=unstream(rR2, o2), fB0=unstream(rB2, o2)).run(steps + 1)
    assert np.max(np.abs(o2w.dense("phi") - o2.dense("phi"))) > 1e-6


# ------------------------------------------------------------------------------------------- the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("name,ny,env", [
    ("srt_capillary", 4, {}),                                  # nx = 16, 28: compact storage of 23 values on short row segments (rk3dq_fused<.., RAGGED>)
    ("srt_porous", 5, {}),
    ("srt_capillary", 4, {"LBMPM_RK3D_LAYOUT": "dense"}),      # dense storage, fused z-marching kernel (rk3d_fused)
    ("srt_porous", 5, {"LBMPM_RK3D_LAYOUT": "dense"}),
    ("srt_porous", 8, {"LBMPM_RK3D_VARIANT": "1"}),            # split schedule (rk3d_phase_field + rk3d_collide)
    ("srt_porous64", 8, {}),                                   # nx = 64: compact storage of 23 values, rk3dq_fused (the bench kernel)
    ("srt_porous64", 11, {"LBMPM_RK3D_CHUNK": "7"}),           # ... cut tiles, several chunks per column
    ("srt_porous64", 8, {"LBMPM_RK3D_STORAGE": "38"}),         # compact storage of both colour lattices, rk3dc_fused
], ids=lambda v: v if isinstance(v, str) else (str(v) if isinstance(v, int) else ",".join("%s=%s" % (k[11:], x) for k, x in v.items()) or "default"))
def test_hip_reproduces_the_reference_2d_driver(name, ny, env, monkeypatch, knobs):
    """y-uniform D3Q19 lattice through the C ABI == captures of the real D2Q9 perturbation driver, every snapshot"""
    from openlbmpm_amd.rk3d import RK3DCluster
    knobs(env)
    d, dom2, par2, par3 = scenario(name)
    dom3 = extrude(dom2, ny)
    c = RK3DCluster(dom3, 1, dict(par3, **RC_EXACT))
    if name == "srt_porous64":
        assert c.slabs[0].dominant_kernel == ("rk3dc_fused" if env.get("LBMPM_RK3D_STORAGE") == "38" else "rk3dq_fused")
    else:
        assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if not env else ("rk3d_collide" if "LBMPM_RK3D_VARIANT" in env else "rk3d_fused"))
    c.set_density(extrude(dense2(d, d["init_rhoR"]), ny), extrude(dense2(d, d["init_rhoB"]), ny))
    done, worst = 0, 0.0
    for k in d["snaps"]:
        c.step(int(k) - 1 - done); done = int(k) - 1
        c.observe()
        got = {f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}
        ref = {f: dense2(d, d["s%d_%s" % (k, f)]) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
        worst = max(worst, check_against(got, ref, dom2, ny, (name, int(k))))
    c.close()
    assert done >= 49
    print("%s: worst field-relative difference to the reference's 2-D driver %.2e" % (name, worst))


@pytest.mark.gpu
@pytest.mark.parametrize("relax_slabs", [1, 3])
def test_hip_default_weights_equal_the_pinned_2d_oracle(relax_slabs):
    """the shipped model (its own recolouring weights), k slabs: equals the pinned D2Q9 oracle with the projected weights"""
    from openlbmpm_amd.rk3d import RK3DCluster
    d, dom2, par2, par3 = scenario("srt_porous64")
    ny, steps = 6, 60
    rR2 = dense2(d, d["init_rhoR"]); rB2 = dense2(d, d["init_rhoB"])
    c = RK3DCluster(extrude(dom2, ny), relax_slabs, par3)
    c.set_density(extrude(rR2, ny), extrude(rB2, ny))
    c.step(steps); c.observe()
    o2 = RKPertOracle(dom2, par2, rhoR0=rR2, rhoB0=rB2)
    o2 = RKPertOracle(dom2, par2, fR0=unstream(rR2, o2), fB0=unstream(rB2, o2), recolor_weights=RW_PROJECTED).run(steps + 1)
    got = {f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}
    ref = {f: o2.dense(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
    check_against(got, ref, dom2, ny, "default weights")
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["q23", "dense"])
def test_pressure_inlet_reduces_to_the_2d_loops_pressure_inlet(layout, monkeypatch):
    """[BoundaryCondition] BoundaryTypeInlet = 'Dirichlet' in 3-D (round 6: lbmpm_rk3d_config.inlet_type; the z-plane form of
    calConstPressureInletGPU, AcceleratedRKGPU2D.py:925-962, ghost plane :968-1002).  No capture of the real 2-D driver runs that inlet
    in the perturbation loop; the chain of pins is: the reference's kernel -> its same-named entry point (rkpert / kats fixtures, 1e-13)
    -> rk2dp_fused with the pressure inlet (tests/test_rk2d_pert_gpu.py, 1e-9) -> this test: a y-uniform D3Q19 lattice through
    rk3dq_fused / rk3d_fused AND through oracle/rk3d_oracle.c, with the projection-exact recolouring weights, against rk2dp_fused."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.rk3d import RK3DCluster
    d, dom2, par2, par3 = scenario("srt_capillary")
    ny, steps = 4, 60
    pin = dict(inlet="Dirichlet", densityRH=1.0e-8, densityBH=1.004)
    rR2 = dense2(d, d["init_rhoR"]); rB2 = dense2(d, d["init_rhoB"])
    o2 = RKPertOracle(dom2, par2, rhoR0=rR2, rhoB0=rB2)                      # (only for its node tables: unstream)
    nxy = dom2.shape
    def dense_pdf(compact):
        out = np.zeros(nxy[0] * nxy[1] * 9).reshape(-1, 9)
        out[d["fluidNodes"]] = compact
        return out.reshape(nxy + (9,))
    s2 = RK2DSolver(dom2, dict(beta=par2["beta"], tauR=par2["tauR"], tauB=par2["tauB"], relax="SRT", inlet="Dirichlet", outlet="Dirichlet",
                               rhoRH=pin["densityRH"], rhoBH=pin["densityBH"], rhoRL=par2["rhoRL"], rhoBL=par2["rhoBL"]),
                    diagnostics=True, perturbation=dict(AkR=par2["AkR"], AkB=par2["AkB"], solidPhi=par2["solidPhi"]))
    s2.set_pdf(dense_pdf(unstream(rR2, o2)), dense_pdf(unstream(rB2, o2)))
    s2.step(steps + 1)
    ref = {f: s2.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
    s2.close()
    assert np.isfinite(ref["phi"]).all() and float(np.max(np.abs(ref["vy"]))) > 1e-5       # the pressure difference drives a flow
    if layout == "dense":
        monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    par = dict(par3, **RC_EXACT, **pin)
    c = RK3DCluster(extrude(dom2, ny), 2, par)
    assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
    c.set_density(extrude(rR2, ny), extrude(rB2, ny))
    c.step(steps); c.observe()
    check_against({f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}, ref, dom2, ny, ("pressure inlet", layout))
    c.close()
    o3 = RK3DOracle(extrude(dom2, ny), extrude(rR2, ny), extrude(rB2, ny), par).run(steps).macro()
    check_against({f: o3.field(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}, ref, dom2, ny, "pressure inlet, oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["q23", "dense"])
def test_convective_outlet_reduces_to_the_2d_loops_convective_outlet(layout, monkeypatch):
    """[BoundaryCondition] BoundaryTypeOutlet = 'Convective' in 3-D (round 6: lbmpm_rk3d_config.outlet_type): the z-plane form of
    convectiveOutletGPU / ...Ghost2GPU / ...Ghost3GPU (AcceleratedRKGPU2D.py:700-784).  Chain of pins as for the pressure inlet: the
    reference's kernels -> their same-named entry points -> rk2dp_fused with the convective outlet (tests/test_rk2d_pert_gpu.py) ->
    this test, for rk3dq_fused + rk3dq_conv_*, rk3d_fused and oracle/rk3d_oracle.c, with the projection-exact recolouring weights."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.rk3d import RK3DCluster
    d, dom2, par2, par3 = scenario("srt_capillary")
    ny, steps = 4, 60
    rR2 = dense2(d, d["init_rhoR"]); rB2 = dense2(d, d["init_rhoB"])
    o2 = RKPertOracle(dom2, par2, rhoR0=rR2, rhoB0=rB2)                      # (only for its node tables: unstream)
    nxy = dom2.shape
    def dense_pdf(compact):
        out = np.zeros(nxy[0] * nxy[1] * 9).reshape(-1, 9)
        out[d["fluidNodes"]] = compact
        return out.reshape(nxy + (9,))
    s2 = RK2DSolver(dom2, dict(beta=par2["beta"], tauR=par2["tauR"], tauB=par2["tauB"], relax="SRT", inlet="Neumann", outlet="Convective",
                               vyR=par2["vyR"], vyB=par2["vyB"]),
                    diagnostics=True, perturbation=dict(AkR=par2["AkR"], AkB=par2["AkB"], solidPhi=par2["solidPhi"]))
    s2.set_pdf(dense_pdf(unstream(rR2, o2)), dense_pdf(unstream(rB2, o2)))
    s2.step(steps + 1)
    ref = {f: s2.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy")}
    s2.close()
    assert np.isfinite(ref["phi"]).all()
    if layout == "dense":
        monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
    par = dict(par3, **RC_EXACT, outlet="Convective")
    c = RK3DCluster(extrude(dom2, ny), 2, par)
    assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
    c.set_density(extrude(rR2, ny), extrude(rB2, ny))
    c.step(steps); c.observe()
    check_against({f: c.get(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}, ref, dom2, ny, ("convective outlet", layout))
    c.close()
    o3 = RK3DOracle(extrude(dom2, ny), extrude(rR2, ny), extrude(rB2, ny), par).run(steps).macro()
    check_against({f: o3.field(f) for f in ("rhoR", "rhoB", "phi", "vx", "vy", "vz")}, ref, dom2, ny, "convective outlet, oracle")
