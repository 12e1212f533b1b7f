"""GPU parity of the fused colour-gradient solver (liblbmpm_hip.so through the C ABI)
against (a) the golden vectors captured from the real reference driver and (b) the CPU
oracle on larger seeded inputs.

Tolerance: the north star asks for 1e-6 relative on rho, u and phase field after N steps;
these tests hold every compared field (f_R, f_B, rho, u, phi, G, F, K) to 1e-9
field-relative on the golden scenarios (<= 200 steps)."""
import os

import numpy as np
import pytest

from helpers import golden_files, load_params, rel_err

pytestmark = pytest.mark.gpu

FILES = golden_files("rk_")
FIELDS = ("fR", "fB", "rhoR", "rhoB", "vx", "vy", "phi", "Gx", "Gy", "Fx", "Fy", "K")
TOL = 1e-9


def _dense(d, compact):
    dom = d["isDomain"]
    out = np.zeros((dom.size,) + compact.shape[1:])
    out[d["fluidNodes"]] = compact
    return out.reshape(dom.shape + compact.shape[1:])


@pytest.mark.parametrize("variant", [0, 1], ids=["fused", "split"])
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_golden_scenarios(path, variant):
    from openlbmpm_amd.rk2d import RK2DSolver
    d = np.load(path)
    par = load_params(d)
    keys = ("sigma", "theta", "wetting", "beta", "delta", "tauR", "tauB", "tautype", "relax", "inlet",
            "outlet", "vyR", "vyB", "rhoBH", "rhoRH", "rhoBL", "rhoRL")
    s = RK2DSolver(d["isDomain"], {k: par[k] for k in keys}, diagnostics=True, variant=variant)
    s.set_pdf(_dense(d, d["init_fR"]), _dense(d, d["init_fB"]))
    done = 0
    for k in d["snaps"]:
        s.step(int(k) - done)
        done = int(k)
        for f in FIELDS:
            e = rel_err(s.get_compact(f), d["s%d_%s" % (k, f)])
            assert e < TOL, "%s step %d field %s rel err %.3e" % (os.path.basename(path), k, f, e)
    # HDF5 record 0 (written during step 1 after the boundary kernels, RKD2Q9.py:1382-1393)
    s.set_pdf(_dense(d, d["init_fR"]), _dense(d, d["init_fB"]))
    h5 = {k.split("/")[-1]: d[k] for k in d.files if k.startswith("h5|")}
    for name, key in (("rec_rhoR", "FluidDensityRin0"), ("rec_rhoB", "FluidDensityBin0"),
                      ("rec_fR", "FluidPDFRat0"), ("rec_fB", "FluidPDFBat0"),
                      ("rec_vx", "FluidVelocityXAt0"), ("rec_vy", "FluidVelocityYAt0")):
        assert rel_err(s.get(name), h5[key]) < TOL, name
    s.close()


def _porous_case(nx, ny, seed):
    from openlbmpm_amd.geometry import porous_disks, image_domain, initial_densities_rk
    img = porous_disks(nx, ny, porosity=0.7, rmin=3.0, rmax=9.0, seed=seed)
    dom = image_domain(img, 6, 0.5)
    rR, rB = initial_densities_rk(dom, True, 6)
    return dom, rR, rB


@pytest.mark.parametrize("variant", [0, 1], ids=["fused", "split"])
@pytest.mark.parametrize("shape", [(150, 97), (70, 203)], ids=["150x97", "70x203"])
def test_porous_vs_oracle(shape, variant):
    """Seeded porous domains that straddle tile boundaries (sizes not multiples of the
    64x16 tile), wetting everywhere: HIP vs the CPU oracle, 60 steps, 1e-9."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from oracle.rk import RKOracle
    dom, rR, rB = _porous_case(shape[0], shape[1], seed=11)
    par = dict(theta=70.0, tauR=1.0, tauB=0.8)
    s = RK2DSolver(dom, par, diagnostics=True, variant=variant)
    s.set_macro(rR, rB)
    o = RKOracle(dom, par, rR, rB)
    for n in (1, 59):
        s.step(n); o.run(n)
        for f in FIELDS:
            e = rel_err(s.get_compact(f), getattr(o, f))
            assert e < TOL, "field %s rel err %.3e after %d steps" % (f, e, s.steps_done)
    s.close()


@pytest.mark.parametrize("tracer", [False, True], ids=["flow", "flow+tracer"])
def test_tile_shapes_and_schedules_agree_bit_for_bit(tracer, knobs):
    """rk2d_fused issues the own node's pulls as asm loads ahead of the fluid mask and waits with hand-counted s_waitcnt; the count of
    mask loads, halo nodes per lane and boundary-row variants differs between the tile shapes (LBMPM_RK2D_SHAPE = 0: 64 x 8, 2: 64 x 16
    with 1 024 threads, 3: 64 x 4 whose halo outnumbers its threads, 1: two nodes per lane -- the plain C++ pull phase).  On a porous
    lattice that is no multiple of any tile, with boundary rows and partial tiles on two edges, every shape must give the state of
    the default one bit for bit, and (without tracer) the split three-kernel schedule, which pulls in plain C++, too."""
    from openlbmpm_amd.rk2d import RK2DSolver
    dom, rR, rB = _porous_case(203, 141, seed=5)
    par = dict(theta=65.0, tauR=1.0, tauB=0.9)

    def run(shape, variant=0):
        knobs({"LBMPM_RK2D_SHAPE": str(shape)})       # (tile shapes other than the default: the development build)
        s = RK2DSolver(dom, par, variant=variant)
        s.set_macro(rR, rB)
        if tracer:
            s.configure_tracers(diffX=(1. / 6.,), diffY=(1. / 6.,), beta=(1.0,), inlet_conc=(1.0,))
            s.set_tracer(0, np.where(rB > 0, 0.5, 0.0))
        s.step(40)
        out = [s.get("fR"), s.get("fB")] + ([s.get_tracer(0)] if tracer else [])
        s.close()
        return out

    ref = run(0)
    assert all(np.isfinite(a).all() for a in ref)
    for shape in ((3,) if tracer else (1, 2, 3)):          # (the tracer step exists for the 64 x 8 and 64 x 4 shapes)
        for a, b in zip(ref, run(shape)):
            assert np.array_equal(a, b), "shape %d" % shape
    if not tracer:
        for a, b in zip(ref, run(0, variant=1)):
            assert np.array_equal(a, b), "split schedule"


def test_full_size_properties():
    """Size-independent properties at the benchmark size (1024 x 1024, BASELINE configs[1]):
    (i) the fused and the split kernel schedules agree bit-for-bit after 50 steps;
    (ii) total mass changes only through the open rows: |dm|/m stays below the inflow bound
         |v_in| * nx * steps / m (plus the same again for the pressure outlet);
    (iii) with the inlet velocity set to zero and... colour is conserved separately to the
         same bound; nothing is NaN."""
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.geometry import simple_geometry, initial_densities_rk
    nx = ny = 1024
    steps = 50
    dom = simple_geometry(nx, ny)
    rR, rB = initial_densities_rk(dom, False, 10, mode="intrusion")
    out = []
    for variant in (0, 1):
        s = RK2DSolver(dom, None, variant=variant)
        s.set_macro(rR, rB)
        s.step(steps)
        out.append((s.get("rhoR"), s.get("rhoB"), s.get("fR")))
        s.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    rho = out[0][0] + out[0][1]
    assert np.isfinite(rho).all()
    m0 = (rR + rB).sum(); m1 = rho.sum()
    assert m1 > 0.99 * m0
    bound = 4.0 * 1.0e-4 * nx * steps / m0
    assert abs(m1 - m0) / m0 < bound, (m0, m1, bound)
    # colour: red only enters through the inlet
    assert abs(out[0][0].sum() - rR.sum()) / rR.sum() < 4.0 * 1.0e-4 * nx * steps / rR.sum()
