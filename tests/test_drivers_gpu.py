"""GPU tests of the driver counterparts (ini in -> result file out with the reference's dataset
names) against what the real reference drivers recorded (golden 'h5|' captures)."""
import os

import numpy as np
import pytest

from helpers import load_params, golden_files, rel_err
from ini_fixtures import write_rk, write_sc

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]



def test_rk_driver_records_match_reference(tmp_path):
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.results import load_results
    d = np.load([f for f in golden_files("rk_") if f.endswith("rk_csf_mrt_capillary.npz")][0])
    write_rk(str(tmp_path), nx=20, ny=48, steps=60, interval=25)
    sim = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out"))
    path = sim.runRKColorGradient2D()
    res = load_results(path)
    assert sim.records == 3          # steps 0, 25, 50
    for name in ("FluidMacro/FluidDensityRin0", "FluidMacro/FluidDensityBin0", "FluidPDF/FluidPDFRat0",
                 "FluidPDF/FluidPDFBat0", "FluidVelocity/FluidVelocityXAt0", "FluidVelocity/FluidVelocityYAt0"):
        g = d["h5|SimulationResultsRK.h5:/" + name]
        assert rel_err(res["/" + name], g) < 1e-9, name
    assert "/FluidMacro/FluidDensityRin2" in res


def test_sc_driver_records_match_reference(tmp_path):
    from openlbmpm_amd.ShanChenD2Q9 import ShanChenD2Q9
    from openlbmpm_amd.results import load_results
    d = np.load([f for f in golden_files("sc_") if f.endswith("sc_sc_srt_convective.npz")][0])
    write_sc(str(tmp_path), inter="ShanChen", nx=20, ny=48, steps=80, outlet="Convective")
    sim = ShanChenD2Q9(str(tmp_path), output_dir=str(tmp_path / "out"))
    path = sim.runTypeSCmodel()
    res = load_results(path)
    assert sim.records == 2          # passes 1 and 81 of 81 (ShanChenD2Q9.py:1561)
    for rec in (0, 1):
        for name in ("FluidMacro/FluidDensityType0in%d", "FluidMacro/FluidDensityType1in%d",
                     "FluidVelocity/FluidVelocityXAt%d", "FluidVelocity/FluidVelocityYAt%d"):
            g = d["h5|SimulationResults.h5:/" + name % rec]
            assert rel_err(res["/" + name % rec], g) < 1e-9, name % rec


def test_transport_driver_records_match_coupled_oracle(tmp_path):
    """Transport2DRK counterpart: flow records as the RK driver writes them, concentration records =
    state after the tracer update of the recording step (Transport2DRK.py:1418-1432)"""
    from ini_fixtures import write_transport
    from openlbmpm_amd.Transport2DRK import Transport2DRK
    from openlbmpm_amd.results import load_results
    from oracle.tr import CoupledOracle
    write_rk(str(tmp_path), nx=20, ny=48, steps=60, interval=25)
    write_transport(str(tmp_path))
    sim = Transport2DRK(str(tmp_path), output_dir=str(tmp_path / "out"))
    flow_path, conc_path = sim.runTransport2DMPMCRKNew()
    assert sim.records == 3
    res = load_results(conc_path)
    # the same run in the oracle
    ref = Transport2DRK(str(tmp_path), output_dir=str(tmp_path / "out2"))
    ref.initializeDomainBorder(); ref.initializeDomainCondition(); ref.initializeTransportDomain()
    p, t = ref.par, ref.tr
    keys = ("sigma", "theta", "wetting", "beta", "delta", "tauR", "tauB", "tautype", "relax", "inlet", "outlet",
            "vyR", "vyB", "rhoBH", "rhoRH", "rhoBL", "rhoRL")
    o = CoupledOracle(ref.isDomain, {k: p[k] for k in keys}, ref.fluidsRhoR, ref.fluidsRhoB, ref.tracerConc,
                      dict(diffX=tuple(t["diffX"]), diffY=tuple(t["diffY"]), dXY=t["dXY"], dYX=t["dYX"], beta=(t["beta"],) * 2,
                           crit=0.5, inlet_conc=tuple(ref.inletConcentration), free_outlet=True, dirichlet_inlet=True))
    sel = ref.isDomain.reshape(-1) == 1
    done = 0
    for rec in range(3):
        o.run(rec * 25 + 1 - done); done = rec * 25 + 1
        for k in range(2):
            got = res["/TransportMacro/TracerConcType%din%d" % (k, rec)].reshape(-1)[sel]
            assert rel_err(got, o.C[k]) < 1e-9, (k, rec)
    assert ref.tracerConc[1].max() == 0.0 and abs(ref.tracerConc[0].max() - 1.0) < 1e-15   # Transport2DRK.py:417-424
    flow = load_results(flow_path)
    assert "/FluidMacro/FluidDensityRin2" in flow and "/FluidVelocity/FluidVelocityYAt2" in flow


def test_transport_restart_and_boundary_spellings(tmp_path):
    """(i) IsCycle = 'yes': the flow restarts from SimulationResultsRK, every tracer from the record of the LAST tracer
    (Transport2DRK.py:431-439, the `[:, :]` assignment); (ii) the loop acts on 'Freeflow' / 'Dirichlet' only (:1363, :1378):
    with the reader's spelling 'FreeFlow' no outlet rule runs; (iii) the inlet value of tracer 0 is 1.0 whatever the file
    says (:1161)"""
    import shutil
    import warnings
    from ini_fixtures import TRANSPORT_INI
    from openlbmpm_amd.Transport2DRK import Transport2DRK
    from openlbmpm_amd.results import load_results
    first = tmp_path / "first"; first.mkdir()
    write_rk(str(first), nx=20, ny=64, steps=30, interval=25)
    (first / "transportsetup.ini").write_text(TRANSPORT_INI)
    a = Transport2DRK(str(first), output_dir=str(tmp_path / "out1"))
    flow_path, conc_path = a.runTransport2DMPMCRKNew()
    init = tmp_path / "LBMInitial"; init.mkdir()
    shutil.copy(flow_path, str(init)); shutil.copy(conc_path, str(init))
    second = tmp_path / "second"; second.mkdir()
    write_rk(str(second), nx=20, ny=64, steps=5, interval=25, cycle="yes", last=1)
    (second / "transportsetup.ini").write_text(TRANSPORT_INI)
    b = Transport2DRK(str(second), output_dir=str(tmp_path / "out2"), initial_dir=str(init))
    b.initializeDomainBorder(); b.initializeDomainCondition(); b.initializeTransportDomain()
    last = load_results(conc_path)["/TransportMacro/TracerConcType1in1"]
    fluid = b.isDomain == 1
    assert np.array_equal(b.tracerConc[0][fluid], last[fluid]) and np.array_equal(b.tracerConc[1][fluid], last[fluid])
    b2 = Transport2DRK(str(second), output_dir=str(tmp_path / "out2"), initial_dir=str(init))
    b2.runTransport2DMPMCRKNew()
    assert np.isfinite(b2.tracerConc).all()
    empty = tmp_path / "none"; empty.mkdir()
    with pytest.raises(Exception, match="TransportResults"):
        c = Transport2DRK(str(second), output_dir=str(tmp_path / "out3"), initial_dir=str(empty))
        c.isDomain = b.isDomain
        c.initializeTransportDomain()
    # ---- spellings
    third = tmp_path / "third"; third.mkdir()
    write_rk(str(third), nx=20, ny=48, steps=40, interval=100)
    (third / "transportsetup.ini").write_text(TRANSPORT_INI)
    from openlbmpm_amd.rk2d import RK2DSolver
    seen = []
    plain = RK2DSolver.configure_tracers

    def spy(self, **kw):
        seen.append((kw["free_outlet"], kw["dirichlet_inlet"], tuple(kw["inlet_conc"])))
        return plain(self, **kw)
    RK2DSolver.configure_tracers = spy
    ref = Transport2DRK(str(third), output_dir=str(tmp_path / "o4")); ref.runTransport2DMPMCRKNew()
    (third / "transportsetup.ini").write_text(TRANSPORT_INI.replace("'Freeflow'", "'FreeFlow'"))
    with pytest.warns(UserWarning, match="Freeflow"):
        alt = Transport2DRK(str(third), output_dir=str(tmp_path / "o5"))
    alt.runTransport2DMPMCRKNew()
    (third / "transportsetup.ini").write_text(TRANSPORT_INI.replace("ConcentrationInlet = 1.0, 0.25", "ConcentrationInlet = 0.5, 0.25"))
    with pytest.warns(UserWarning, match="ConcentrationInlet"):
        same = Transport2DRK(str(third), output_dir=str(tmp_path / "o6"))
    same.runTransport2DMPMCRKNew()
    assert np.array_equal(same.tracerConc[0], ref.tracerConc[0])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        own = Transport2DRK(str(third), output_dir=str(tmp_path / "o7"), inlet_concentration_from_ini=True)
    own.runTransport2DMPMCRKNew()
    RK2DSolver.configure_tracers = plain
    assert seen == [(True, True, (1.0, 0.25)), (False, True, (1.0, 0.25)), (True, True, (1.0, 0.25)), (True, True, (0.5, 0.25))]


def test_restart_from_previous_results(tmp_path):
    """[CyclesSetup] IsCycle = 'yes' (RKD2Q9.py:491-508): densities and velocities of record LastStep,
    top 20 rows refilled with blue, populations at equilibrium"""
    import shutil
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.results import load_results
    first = tmp_path / "first"; first.mkdir()
    write_rk(str(first), nx=20, ny=64, steps=60, interval=25)
    a = RKColorGradientLBM(str(first), output_dir=str(tmp_path / "out1"))
    path = a.runRKColorGradient2D()
    init = tmp_path / "LBMInitial"; init.mkdir()
    shutil.copy(path, str(init))
    second = tmp_path / "second"; second.mkdir()
    write_rk(str(second), nx=20, ny=64, steps=30, interval=25, cycle="yes", last=2)
    b = RKColorGradientLBM(str(second), output_dir=str(tmp_path / "out2"), initial_dir=str(init))
    b.initializeDomainBorder(); b.initializeDomainCondition()
    old = load_results(path)
    assert np.array_equal(b.fluidsRhoR[:-20], old["/FluidMacro/FluidDensityRin2"][:-20])
    assert np.array_equal(b.physicalVY, old["/FluidVelocity/FluidVelocityYAt2"])
    assert b.fluidsRhoR[-20:].max() == 0.0 and b.fluidsRhoB[-20:, 1:-1].min() > 0.0
    b2 = RKColorGradientLBM(str(second), output_dir=str(tmp_path / "out2"), initial_dir=str(init))
    res = load_results(b2.runRKColorGradient2D())
    r0 = res["/FluidMacro/FluidDensityRin0"] + res["/FluidMacro/FluidDensityBin0"]
    assert np.isfinite(r0).all() and abs(r0.sum() - (b.fluidsRhoR + b.fluidsRhoB).sum()) / r0.sum() < 1e-3
    with pytest.raises(Exception):
        RKColorGradientLBM(str(second), output_dir=str(tmp_path / "o3"), initial_dir=str(tmp_path / "nowhere")).runRKColorGradient2D()


def test_rk_image_cycle_takes_cycleInitialRK_over_with_the_colours_swapped_in_the_buffer_rows(tmp_path):
    """[CyclesSetup] IsCycle = 'yes' WITH a pore image (RKD2Q9.py:532-556): densities and populations of ~/LBMInitial/cycleInitialRK
    taken over, red and blue changing places in the top numBufferingLayers rows (:540-556), the velocity whole -- asserted row by row
    against a hand-built file; then a real previous state run on from there."""
    import re
    from openlbmpm_amd import config
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.geometry import porous_disks
    from openlbmpm_amd.results import ResultFile
    img = porous_disks(44, 60, porosity=0.7, rmin=2.5, rmax=5.0, seed=3)
    write_rk(str(tmp_path), steps=30, interval=30)
    ini = tmp_path / "RKtwophasesetup2D.ini"
    ini.write_text(re.sub(r"(?m)^(\s*Existance\s*=).*$", r"\1 'yes'", ini.read_text()))
    a = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "a"), image=img)
    a.runRKColorGradient2D()
    dom, nb = a.isDomain, a.par["nbuf"]
    fluid = dom == 1
    init = tmp_path / "LBMInitial"

    def write(rR, rB, fR, fB, vx, vy):
        out = ResultFile(str(init), "cycleInitialRK", (("FluidMacro", "MacroData"), ("FluidPDF", "MicroData"), ("FluidVelocity", "MacroVelocity")))
        out.write("FluidMacro", "FluidDensityR", rR); out.write("FluidMacro", "FluidDensityB", rB)
        out.write("FluidPDF", "FluidPDFR", fR); out.write("FluidPDF", "FluidPDFB", fB)
        out.write("FluidVelocity", "FluidVelocityX", vx); out.write("FluidVelocity", "FluidVelocityY", vy)

    rng = np.random.default_rng(11)
    hand = [np.where(fluid, rng.random(dom.shape), 0.0) for _ in range(2)] + [np.where(fluid[..., None], rng.random(dom.shape + (9,)), 0.0) for _ in range(2)] + \
           [np.where(fluid, rng.standard_normal(dom.shape), 0.0) for _ in range(2)]
    write(*hand)
    ini.write_text(re.sub(r"(?m)^(\s*IsCycle\s*=).*$", r"\1 'yes'", ini.read_text()))
    b = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "b"), image=img, initial_dir=str(init))
    b.initializeDomainBorder(); b.initializeDomainCondition()
    rR, rB, fR, fB, vx, vy = hand
    ny = dom.shape[0]
    for i in range(ny):
        top = i >= ny - nb
        assert np.array_equal(b.fluidsRhoR[i], rB[i] if top else rR[i]) and np.array_equal(b.fluidsRhoB[i], rR[i] if top else rB[i]), i
        assert np.array_equal(b.fluidPDFR[i], fB[i] if top else fR[i]) and np.array_equal(b.fluidPDFB[i], fR[i] if top else fB[i]), i
        assert np.array_equal(b.physicalVX[i], vx[i]) and np.array_equal(b.physicalVY[i], vy[i]), i
    # a dataset of another shape, or none, is refused with the dataset's name
    write(hand[0], hand[1], hand[2][:, :-1], hand[3], hand[4], hand[5])
    with pytest.raises(config.ConfigError, match="FluidPDFR"):
        RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "c"), image=img, initial_dir=str(init)).runRKColorGradient2D()
    # the state run a left behind, run on with the colours swapped in the buffer rows
    s = a.solver
    write(s.get("rec_rhoR"), s.get("rec_rhoB"), s.get("fR"), s.get("fB"), s.get("rec_vx"), s.get("rec_vy"))
    c = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "c"), image=img, initial_dir=str(init))
    c.timeSteps = 20
    from openlbmpm_amd.results import load_results
    res = load_results(c.runRKColorGradient2D())
    r0, b0 = res["/FluidMacro/FluidDensityRin0"], res["/FluidMacro/FluidDensityBin0"]
    assert np.isfinite(r0).all() and np.isfinite(b0).all()
    # blue filled the top buffer rows of run a; after the swap they are red
    assert float(b0[-nb + 2:-2].sum()) < 0.05 * float(r0[-nb + 2:-2].sum())
    assert abs(float(r0[:-nb].sum()) - float(s.get("rec_rhoR")[:-nb].sum())) < 1e-2 * float(r0.sum())


def test_sc_driver_with_iso8_scheme_matches_reference(tmp_path):
    """[ForceScheme] ExplicitScheme = 8 through the driver: the run ends where the reference's ends"""
    from openlbmpm_amd.ShanChenD2Q9 import ShanChenD2Q9
    d = np.load([f for f in golden_files("sc_") if f.endswith("sc_efs_srt_iso8.npz")][0])
    write_sc(str(tmp_path), inter="EFS", nx=20, ny=48, steps=60, scheme=8)
    sim = ShanChenD2Q9(str(tmp_path), output_dir=str(tmp_path / "out"))
    sim.runTypeSCmodel()
    sel = d["isDomain"].reshape(-1) == 1
    for k in (0, 1):
        got = sim.solver.get("rho%d" % k).reshape(-1)[sel]
        assert rel_err(got, d["s60_rho"][k]) < 1e-9


@pytest.mark.parametrize("scenario,kw", [("efs_srt_freeflow", dict(outlet="Freeflow")), ("efs_srt_chang", dict(method="Chang"))])
def test_sc_driver_runs_the_loop_alternates(tmp_path, scenario, kw):
    """BoundaryTypeOutlet = 'Freeflow' (ShanChenD2Q9.py:1865-1884) and BoundaryMethod = 'Chang' (:1999-2006) through the
    driver: the run ends where the real driver's ends"""
    from openlbmpm_amd.ShanChenD2Q9 import ShanChenD2Q9
    d = np.load([f for f in golden_files("sc_") if f.endswith("sc_%s.npz" % scenario)][0])
    par = load_params(d)
    write_sc(str(tmp_path), inter="EFS", nx=20, ny=48, steps=60, **kw)
    text = (tmp_path / "efs2D.ini").read_text().replace("FluidsTau = 1.,1.", "FluidsTau = %r,%r" % (par["tau0"], par["tau1"]))
    (tmp_path / "efs2D.ini").write_text(text)
    sim = ShanChenD2Q9(str(tmp_path), output_dir=str(tmp_path / "out"))
    sim.runTypeSCmodel()
    sel = d["isDomain"].reshape(-1) == 1
    for k in (0, 1):
        got = sim.solver.get("rho%d" % k).reshape(-1)[sel]
        assert rel_err(got, d["s60_rho"][k]) < 1e-9


def test_sc_next_cycle_starts_from_previous_records(tmp_path):
    """[DICycles] Option = 'yes' (ShanChenD2Q9.py:788-815): old fluid from the record LastStep below the top 30 rows,
    new fluid in the top 30 rows"""
    import shutil
    from openlbmpm_amd.ShanChenD2Q9 import ShanChenD2Q9
    from openlbmpm_amd.results import load_results
    from openlbmpm_amd.geometry import porous_disks
    img = np.where(porous_disks(40, 60, porosity=0.75, rmin=2.0, rmax=4.0, seed=2) == 1, 255.0, 0.0)
    first = tmp_path / "first"; first.mkdir()
    write_sc(str(first), inter="EFS", steps=40, image="yes")
    a = ShanChenD2Q9(str(first), output_dir=str(tmp_path / "o1"), image=img)
    path = a.runTypeSCmodel()
    init = tmp_path / "LBMInitial"; init.mkdir()
    shutil.copy(path, str(init))
    second = tmp_path / "second"; second.mkdir()
    write_sc(str(second), inter="EFS", steps=10, image="yes", cycle="yes", last=0)
    b = ShanChenD2Q9(str(second), output_dir=str(tmp_path / "o2"), image=img, initial_dir=str(init))
    b.initializeDomainBorder(); b.initializeDomainCondition()
    old = load_results(path)["/FluidMacro/FluidDensityType0in0"]
    fluid = b.isDomain == 1
    assert np.array_equal(b.fluidsDensity[0][:-30][fluid[:-30]], old[:-30][fluid[:-30]])
    assert np.all(b.fluidsDensity[1][-30:][fluid[-30:]] == b.par["rho1"]) and np.all(b.fluidsDensity[0][-30:][fluid[-30:]] == b.par["bg0"])
    b2 = ShanChenD2Q9(str(second), output_dir=str(tmp_path / "o2"), image=img, initial_dir=str(init))
    res = load_results(b2.runTypeSCmodel())
    assert np.isfinite(res["/FluidMacro/FluidDensityType0in0"]).all()
    from openlbmpm_amd import config
    write_sc(str(second), inter="EFS", steps=10, image="no", cycle="yes")
    with pytest.raises(config.ConfigError):
        ShanChenD2Q9(str(second))


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_rk3d_driver_records_match_oracle(tmp_path, relax):
    """RKColorGradient3D(ini).runRKColorGradient3D(): duct from the ini sizes, records every
    `record_every` steps plus the final state; compared with the 3-D oracle on the same set-up"""
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.geometry import initial_densities_rk3d
    from openlbmpm_amd.results import load_results
    from oracle.rk3d import RK3DOracle
    write_rk3d(str(tmp_path), nx=20, ny=14, nz=40, steps=30, relax=relax)
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=12)
    res = load_results(sim.runRKColorGradient3D())
    assert sim.records == 4                     # steps 0, 12, 24 and the final state (30)
    dom = duct(20, 14, 40)
    rR, rB = initial_densities_rk3d(dom, 10)
    o = RK3DOracle(dom, rR, rB, dict(tauB=0.9, relax=relax))
    for k, n in enumerate((0, 12, 12, 6)):
        o.run(n).macro()
        for name, f in (("FluidMacro/FluidDensityRin%d", "rhoR"), ("FluidMacro/FluidDensityBin%d", "rhoB"),
                        ("FluidVelocity/FluidVelocityZAt%d", "vz"), ("FluidVelocity/FluidVelocityXAt%d", "vx")):
            assert rel_err(res["/" + name % k], o.field(f)) < 1e-10, (name, k)


def test_cli_runs(tmp_path):
    from openlbmpm_amd.__main__ import main
    write_sc(str(tmp_path), inter="EFS", steps=40, relax="MRT")
    assert main(["sc", str(tmp_path), "--out", str(tmp_path / "o")]) == 0
    from ini_fixtures import write_rk3d
    write_rk3d(str(tmp_path), nx=16, ny=12, nz=24, steps=10)
    assert main(["rk3d", str(tmp_path), "--out", str(tmp_path / "o3")]) == 0
    from ini_fixtures import write_rk3d_csf
    write_rk3d_csf(str(tmp_path), nx=16, ny=12, nz=24, steps=10)            # the same entry with [SurfaceTension] SurfaceTensionType = 'CSF'
    assert main(["rk3d", str(tmp_path), "--out", str(tmp_path / "o3c")]) == 0


@pytest.mark.parametrize("calibrate,gather,ranks", [(False, True, 2), (True, True, 2), (False, False, 2), (False, True, 8)],
                         ids=["equal-fluid-cuts", "measured-re-cut", "a-file-per-rank", "eight-ranks"])
def test_rk3d_driver_under_two_processes_writes_the_same_records(tmp_path, calibrate, gather, ranks):
    """RKColorGradient3D under torchrun (two ranks sharing this GPU, gloo transport): rank 0 writes ONE file with the whole lattice's
    arrays (the reference's record is one dense array per field, RKD2Q9.py:938-957), equal to the single-process driver's bit for bit --
    also after the one measured re-cut of the slabs a long run makes at its start (calibrate_partition; wherever the cuts land, the slab
    step equals the single domain); gather_records = False: every rank writes the planes it owns, stacked they are the same arrays.
    eight-ranks: the node's rank count rehearsed on this one GPU (slabs of five planes), the stitched record read back."""
    import os
    import subprocess
    import sys
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
    from openlbmpm_amd.results import load_results
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    write_rk3d(str(tmp_path), nx=20, ny=14, nz=40, steps=20, relax="MRT")
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
dist.init_process_group("gloo")
torch.cuda.set_device(0)
sim = RKColorGradient3D(%r, output_dir=%r, record_every=8, device=0)
sim.calibrate_partition = %r
sim.gather_records = %r
sim.runRKColorGradient3D()
dist.destroy_process_group()
''' % (root, str(tmp_path), str(tmp_path / "out2"), calibrate, gather))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=dict(os.environ), timeout=600)
    single = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out1"), record_every=8)
    ref = load_results(single.runRKColorGradient3D())
    assert len(ref) == 5 * single.records
    if gather:
        files = os.listdir(tmp_path / "out2")
        assert len(files) == 1 and files[0].startswith("SimulationResultsRK3D."), files
        got = load_results(str(tmp_path / "out2" / files[0]))
        assert set(got) == set(ref)
        for key in ref:
            assert np.array_equal(got[key], ref[key]), key
        return
    parts = []
    for r in range(2):
        files = [f for f in os.listdir(tmp_path / "out2") if f.startswith("SimulationResultsRK3D_rank%d." % r)]
        assert len(files) == 1
        parts.append(load_results(str(tmp_path / "out2" / files[0])))
    assert set(parts[0]) == set(ref)
    for key in ref:
        assert np.array_equal(np.concatenate([parts[0][key], parts[1][key]], axis=0), ref[key]), key


def test_rk3d_shipped_ini_sizes_run_the_fast_kernel_through_the_cli(tmp_path):
    """the reference's own 3-D input is 32 x 32 x 96 (IniFiles/RKtwophasesetup3D.ini:5-7): `python -m openlbmpm_amd rk3d` runs it on
    rk3dq_fused (the compact 23-value storage serves every nx), records equal the oracle's"""
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.__main__ import main
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.geometry import initial_densities_rk3d
    from openlbmpm_amd.results import load_results
    from oracle.rk3d import RK3DOracle
    write_rk3d(str(tmp_path), steps=40)                        # the ini's sizes and parameters, fewer steps
    assert main(["rk3d", str(tmp_path), "--out", str(tmp_path / "cli")]) == 0
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=40)
    res = load_results(sim.runRKColorGradient3D())
    assert sim.solver.dominant_kernel == "rk3dq_fused" and sim.isDomain.shape == (96, 32, 32)
    dom = duct(32, 32, 96)
    rR, rB = initial_densities_rk3d(dom, 10)
    o = RK3DOracle(dom, rR, rB, dict(tauB=0.9)).run(40).macro()
    for name, f in (("FluidMacro/FluidDensityRin1", "rhoR"), ("FluidMacro/FluidDensityBin1", "rhoB"), ("FluidVelocity/FluidVelocityZAt1", "vz")):
        assert rel_err(res["/" + name], o.field(f)) < 1e-10, name


def test_rk3d_driver_with_a_pressure_inlet_and_a_convective_outlet(tmp_path):
    """[BoundaryCondition] BoundaryTypeInlet = 'Dirichlet' + densityRH / densityBH (the keys of RKtwophasesetup2D.ini) and
    BoundaryTypeOutlet = 'Convective' through the 3-D driver: records equal the oracle's"""
    import re
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.geometry import initial_densities_rk3d
    from openlbmpm_amd.results import load_results
    from oracle.rk3d import RK3DOracle
    write_rk3d(str(tmp_path), nx=24, ny=14, nz=40, steps=30, relax="MRT")
    ini = tmp_path / "RKtwophasesetup3D.ini"
    text = re.sub(r"(?m)^(\s*BoundaryTypeInlet\s*=).*$", r"\1 'Dirichlet'\ndensityRH = 1e-8\ndensityBH = 1.004", ini.read_text())
    ini.write_text(re.sub(r"(?m)^(\s*BoundaryTypeOutlet\s*=).*$", r"\1 'Convective'", text))         # ... and the convective outlet
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=30)
    res = load_results(sim.runRKColorGradient3D())
    dom = duct(24, 14, 40)
    rR, rB = initial_densities_rk3d(dom, 10)
    o = RK3DOracle(dom, rR, rB, dict(tauB=0.9, relax="MRT", inlet="Dirichlet", densityRH=1e-8, densityBH=1.004, outlet="Convective")).run(30).macro()
    for name, f in (("FluidMacro/FluidDensityRin1", "rhoR"), ("FluidMacro/FluidDensityBin1", "rhoB"), ("FluidVelocity/FluidVelocityZAt1", "vz")):
        assert rel_err(res["/" + name], o.field(f)) < 1e-10, name
    assert float(np.abs(res["/FluidVelocity/FluidVelocityZAt1"]).max()) > 1e-5


def _set_cycle(ini_dir, last_step):
    import re
    f = os.path.join(ini_dir, "RKtwophasesetup3D.ini")
    text = open(f).read()
    text = re.sub(r"(?m)^(\s*IsCycle\s*=).*$", r"\1 'yes'", text)
    text = re.sub(r"(?m)^(\s*LastStep\s*=).*$", r"\1 %d" % last_step, text)
    open(f, "w").write(text)


def test_rk3d_cycle_restart_from_the_last_record(tmp_path):
    """[CyclesSetup] IsCycle = 'yes' without an image (RKD2Q9.py:492-508 in 3-D): densities and velocity of record LastStep of
    ~/LBMInitial/SimulationResultsRK3D, the top 20 planes refilled with blue, populations = their equilibria -- checked against
    the oracle started from exactly those equilibria"""
    import shutil
    from ini_fixtures import write_rk3d
    from test_rk3d_state_gpu import equilibrium
    from openlbmpm_amd import config
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.results import load_results
    from oracle.rk3d import RK3DOracle
    write_rk3d(str(tmp_path), nx=20, ny=14, nz=60, steps=24)
    first = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=12)
    prev = load_results(first.runRKColorGradient3D())
    init = tmp_path / "LBMInitial"; init.mkdir()
    _set_cycle(str(tmp_path), 2)
    with pytest.raises(config.ConfigError, match="SimulationResultsRK3D"):
        RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "o2"), initial_dir=str(init)).runRKColorGradient3D()
    shutil.copy(first.result_path, str(init))
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "o2"), initial_dir=str(init), record_every=10)
    sim.timeSteps = 10
    res = load_results(sim.runRKColorGradient3D())
    dom = duct(20, 14, 60)
    rR, rB = prev["/FluidMacro/FluidDensityRin2"].copy(), prev["/FluidMacro/FluidDensityBin2"].copy()
    rR[-20:] = 0.0; rB[-20:] = np.where(dom[-20:] == 1, 1.0, 0.0)
    v = [prev["/FluidVelocity/FluidVelocity%sAt2" % ax] for ax in "XYZ"]
    # record 0 = the fields the run starts from (away from the Zou-He planes, which impose their own densities)
    assert rel_err(res["/FluidMacro/FluidDensityRin0"][2:-2], rR[2:-2]) < 1e-14 and rel_err(res["/FluidMacro/FluidDensityBin0"][2:-2], rB[2:-2]) < 1e-14
    o = RK3DOracle(dom, rR, rB, dict(tauB=0.9)).set_populations(equilibrium(rR, *v), equilibrium(rB, *v)).run(10).macro()
    for name, f in (("FluidMacro/FluidDensityRin1", "rhoR"), ("FluidMacro/FluidDensityBin1", "rhoB"), ("FluidVelocity/FluidVelocityZAt1", "vz")):
        assert rel_err(res["/" + name], o.field(f)) < 1e-10, name


def test_rk3d_image_cycle_takes_the_populations_over_with_the_colours_swapped(tmp_path):
    """the image branch (RKD2Q9.py:532-556 in 3-D): cycleInitialRK3D's populations, colours swapped in the top buffer planes; below
    them the run continues the previous one (to rounding: the populations pass through the 38-value form)"""
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
    from openlbmpm_amd.geometry import porous_spheres
    from openlbmpm_amd.results import load_results
    write_rk3d(str(tmp_path), steps=30)
    vox = porous_spheres(40, 18, 30, porosity=0.7, rmin=2.0, rmax=5.0, seed=8, nbuf=0, walls=False)
    init = tmp_path / "LBMInitial"
    a = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "a"), domain=None, record_every=30, num_buffering_layers=6)
    from openlbmpm_amd.geometry import voxel_domain
    a._domain = voxel_domain(vox, 6)
    a.runRKColorGradient3D()
    path = a.write_cycle_initial(str(init))
    stored = load_results(path)
    fR, fB = a.solver.get_pdf()
    assert np.array_equal(stored["/FluidPDF/FluidPDFR"], fR) and stored["/FluidPDF/FluidPDFR"].shape == a._domain.shape + (19,)
    a.solver.step_single(6)
    a.solver.phase_field(diagnostics=True)
    want = {f: a.solver.get(f) for f in ("rhoR", "rhoB", "vz")}
    _set_cycle(str(tmp_path), 0)
    b = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "b"), domain=a._domain, record_every=6, num_buffering_layers=6, initial_dir=str(init))
    b.timeSteps = 6
    b.initializeDomainBorder(); b.initializeDomainCondition()
    # the swap, plane by plane (RKD2Q9.py:540-556)
    assert np.array_equal(b.fluidPDFR[:-6], fR[:-6]) and np.array_equal(b.fluidPDFB[:-6], fB[:-6])
    assert np.array_equal(b.fluidPDFR[-6:], fB[-6:]) and np.array_equal(b.fluidPDFB[-6:], fR[-6:])
    assert np.array_equal(b.fluidsRhoR[-6:], stored["/FluidMacro/FluidDensityB"][-6:])
    res = load_results(b.runRKColorGradient3D())
    # below the buffer (and the planes the swapped colours reach in 6 steps: two per step) the run continues the previous one
    z = slice(0, a._domain.shape[0] - 6 - 14)
    for name, f in (("FluidMacro/FluidDensityRin1", "rhoR"), ("FluidMacro/FluidDensityBin1", "rhoB"), ("FluidVelocity/FluidVelocityZAt1", "vz")):
        assert rel_err(res["/" + name][z], want[f][z], scale=float(np.max(np.abs(want[f])))) < 1e-11, name
    # ... and in the buffer the colours have changed places
    assert res["/FluidMacro/FluidDensityBin0"][-4:-2].sum() < 1e-2 * res["/FluidMacro/FluidDensityRin0"][-4:-2].sum()


def test_rk3d_checkpoint_restart_is_bit_exact(tmp_path):
    """checkpoint() / restart_from=: records of the continued run == records of the uninterrupted run, bit for bit"""
    from ini_fixtures import write_rk3d
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
    from openlbmpm_amd.results import load_results
    write_rk3d(str(tmp_path), nx=36, ny=14, nz=40, steps=40, relax="MRT")
    whole = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "whole"), record_every=10)
    ref = load_results(whole.runRKColorGradient3D())
    a = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "a"), record_every=10, checkpoint_every=20)
    a.runRKColorGradient3D()
    assert os.path.isfile(a.checkpoint_path)
    b = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "b"), record_every=10, restart_from=a.checkpoint_path)
    got = load_results(b.runRKColorGradient3D())
    assert b.records == whole.records == 5
    for k in (2, 3, 4):
        for name in ("FluidMacro/FluidDensityRin%d", "FluidMacro/FluidDensityBin%d", "FluidVelocity/FluidVelocityXAt%d", "FluidVelocity/FluidVelocityZAt%d"):
            assert np.array_equal(got["/" + name % k], ref["/" + name % k]), (name, k)


def test_records_are_checked_for_nan_and_logged(tmp_path, caplog):
    """at the output cadence every driver checks its recorded fields (the reference lets NaN run into its result
    files) and logs one line with the step, the update rate and the masses"""
    import logging
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.results import SimulationDiverged
    write_rk(str(tmp_path), nx=20, ny=48, steps=60, interval=20)
    sim = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out"))
    with caplog.at_level(logging.INFO, logger="openlbmpm_amd"):
        sim.runRKColorGradient2D()
    lines = [r.getMessage() for r in caplog.records if r.name == "openlbmpm_amd"]
    assert len(lines) == 3 and "record 2 step 40" in lines[2] and "MLUPS" in lines[2] and "massR" in lines[2]
    # a run that blows up (tau -> 1/2 with a strong inlet) is stopped at the first record holding NaN / Inf
    import re
    bad = tmp_path / "bad"; bad.mkdir()
    write_rk(str(bad), nx=20, ny=48, steps=4000, interval=500)
    text = (bad / "RKtwophasesetup2D.ini").read_text()
    text = re.sub(r"(?m)^(\s*velocityYR\s*=).*$", r"\1 -0.6", text)
    (bad / "RKtwophasesetup2D.ini").write_text(text)
    blow = RKColorGradientLBM(str(bad), output_dir=str(tmp_path / "out2"))
    with pytest.raises(SimulationDiverged, match="non-finite"):
        blow.runRKColorGradient2D()
    quiet = RKColorGradientLBM(str(bad), output_dir=str(tmp_path / "out3"))
    quiet.nan_guard = "off"
    quiet.runRKColorGradient2D()                     # the reference's behaviour: NaN in the file, no complaint
    assert not np.isfinite(quiet.fluidsRhoR).all()


@pytest.mark.parametrize("name", ["srt_capillary", "mrt_capillary"])
def test_perturbation_driver_in_the_literal_order_of_the_reference_loop(tmp_path, name):
    """`perturbation_order = "literal"` (opt-in): the loop WITHOUT repair R3 -- calTotalFluidPDF right after streaming, RKD2Q9.py:1065 --
    kernel by kernel on the kernel-level layer, against tests/golden/rkpert_*_literal.npz (the real driver run with R1, R2, R4
    only).  What R3 changes is measured in tests/test_oracle_rk_pert.py: |phi_literal - phi_repaired| = 0.63 after 80 steps."""
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    d = np.load([f for f in golden_files("rkpert_") if f.endswith("rkpert_%s_literal.npz" % name)][0])
    assert not any(str(r).startswith("R3") for r in d["repairs"])
    par = load_params(d)
    write_rk(str(tmp_path), nx=par["nx"], ny=par["ny"], steps=par["steps"], interval=25, relax=par["relax"])
    ini = tmp_path / "RKtwophasesetup2D.ini"
    ini.write_text(ini.read_text().replace("SurfaceTensionType = 'CSF'", "SurfaceTensionType = 'Perturbation'"))
    sim = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out"))
    sim.perturbation_order = "literal"
    sim.nan_guard = "off"
    sim.par.update({k: par[k] for k in ("beta", "delta", "tauR", "tauB", "vyR", "vyB", "rhoBL", "rhoRL", "nbuf")},
                   AkR=float(d["AkR"]), AkB=float(d["AkB"]), solidPhi=0.5)
    dom = d["isDomain"]
    dense = lambda c: (lambda a: (a.__setitem__(d["fluidNodes"], c), a.reshape(dom.shape + c.shape[1:]))[1])(np.zeros((dom.size,) + c.shape[1:]))
    snaps = [int(k) for k in d["snaps"]]
    seen = {}

    def at(step):
        if step in snaps:
            T = sim._pert_table
            seen[step] = {k: T[v].copy_to_host() for k, v in dict(fR="fluidPDFR", fB="fluidPDFB", fTot="fluidPDFTotal", rhoR="fluidRhoR",
                                                                   rhoB="fluidRhoB", phi="phiValue", vx="physicalVX", vy="physicalVY").items()}
    sim.runRKColorGradient2DPerturbation(progress=at, initial_pdf=(dense(d["init_fR"]), dense(d["init_fB"])))
    assert sorted(seen) == snaps
    for k in snaps:
        for f, got in seen[k].items():
            assert rel_err(got, d["s%d_%s" % (k, f)]) < 1e-12, (k, f)
    fused = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out2"))
    fused.perturbation_order, fused.perturbation_schedule = "literal", "fused"
    with pytest.raises(ValueError, match="repaired order"):
        fused.runRKColorGradient2DPerturbation()


@pytest.mark.parametrize("name", ["srt_capillary", "mrt_capillary"])
def test_perturbation_driver_reproduces_the_repaired_reference_driver(tmp_path, name):
    """[SurfaceTension] SurfaceTensionType = 'Perturbation' through the driver (kernel by kernel on the kernel-level layer):
    the state at the end of the captured time steps of the real runRKColorGradient2DPerturbation (repairs R1-R4), and the
    records it writes"""
    from openlbmpm_amd.RKD2Q9 import RKColorGradientLBM
    from openlbmpm_amd.results import load_results
    d = np.load([f for f in golden_files("rkpert_") if f.endswith("rkpert_%s.npz" % name)][0])
    par = load_params(d)
    write_rk(str(tmp_path), nx=par["nx"], ny=par["ny"], steps=par["steps"], interval=25, relax=par["relax"])
    ini = tmp_path / "RKtwophasesetup2D.ini"
    ini.write_text(ini.read_text().replace("SurfaceTensionType = 'CSF'", "SurfaceTensionType = 'Perturbation'"))
    sim = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out"))
    sim.perturbation_schedule = "kernels"
    sim.par.update({k: par[k] for k in ("beta", "delta", "tauR", "tauB", "vyR", "vyB", "rhoBL", "rhoRL", "nbuf")},
                   AkR=float(d["AkR"]), AkB=float(d["AkB"]), solidPhi=0.5)
    dom = d["isDomain"]
    dense = lambda c: (lambda a: (a.__setitem__(d["fluidNodes"], c), a.reshape(dom.shape + c.shape[1:]))[1])(np.zeros((dom.size,) + c.shape[1:]))
    snaps = [int(k) for k in d["snaps"]]
    seen = {}

    def at(step):
        if step in snaps:
            T = sim._pert_table
            seen[step] = {k: T[v].copy_to_host() for k, v in dict(fR="fluidPDFR", fB="fluidPDFB", fTot="fluidPDFTotal", rhoR="fluidRhoR",
                                                                   rhoB="fluidRhoB", phi="phiValue", vx="physicalVX", vy="physicalVY").items()}
    path = sim.runRKColorGradient2DPerturbation(progress=at, initial_pdf=(dense(d["init_fR"]), dense(d["init_fB"])))
    assert np.array_equal(sim.fluidNodes, d["fluidNodes"]) and sorted(seen) == snaps
    for k in snaps:
        for f, got in seen[k].items():
            assert rel_err(got, d["s%d_%s" % (k, f)]) < 1e-12, (k, f)
    res = load_results(path)
    assert sim.records == (par["steps"] - 1) // 25 + 1 and "/FluidMacro/FluidDensityRin%d" % (sim.records - 1) in res
    # the same run on the fused solver (one launch per step; the default where it applies): the same records
    fused = RKColorGradientLBM(str(tmp_path), output_dir=str(tmp_path / "out_fused"))
    fused.par.update(sim.par)
    ticks = []
    path2 = fused.runRKColorGradient2DPerturbation(progress=ticks.append, initial_pdf=(dense(d["init_fR"]), dense(d["init_fB"])))
    assert fused.solver.model == "Perturbation" and ticks[-1] == par["steps"] and fused.records == sim.records
    res2 = load_results(path2)
    assert sorted(res2) == sorted(res)
    for key in res:
        scale = max(float(np.max(np.abs(res[k2]))) for k2 in res if k2.split("/")[1] == key.split("/")[1] and k2.rstrip("0123456789")[-3:] == key.rstrip("0123456789")[-3:])
        assert float(np.max(np.abs(res2[key] - res[key]))) <= 1e-9 * max(scale, 1e-300), key


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
def test_rk3d_driver_runs_the_csf_loop(tmp_path, relax):
    """[SurfaceTension] SurfaceTensionType = 'CSF' in RKtwophasesetup3D.ini: RKColorGradient3D runs the 2-D CSF loop carried to D3Q19; its
    records hold what the reference records (the lattice after the next step's boundary planes, RKD2Q9.py:1382-1393) == the oracle's;
    a run continued from a checkpoint equals the uninterrupted one bit for bit"""
    from ini_fixtures import write_rk3d_csf
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.geometry import initial_densities_rk3d
    from openlbmpm_amd.results import load_results
    from oracle.rk3dcsf import RK3DCSFOracle
    write_rk3d_csf(str(tmp_path), nx=14, ny=12, nz=40, steps=24, relax=relax, sigma=0.05, theta=60.0)
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=12, checkpoint_every=12)
    res = load_results(sim.runRKColorGradient3D())
    assert sim.solver.dominant_kernel == "csf3d_collide" and sim.records == 3
    dom = duct(14, 12, 40)
    rR, rB = initial_densities_rk3d(dom, 10, 1.0, 1.0)
    par = dict(sigma=0.05, theta=60.0, wetting=2, beta=1.0, delta=0.98, tauR=1.0, tauB=0.9, tautype=2, relax=relax, velocityZR=0.0, velocityZB=-1.0e-4,
               densityBL=1.0, densityRL=1.0e-8)
    fl = dom == 1
    for k in range(3):
        o = RK3DCSFOracle(dom, rR, rB, par).run(12 * k).step_a()
        for name, key in (("rhoR", "/FluidMacro/FluidDensityRin%d" % k), ("rhoB", "/FluidMacro/FluidDensityBin%d" % k),
                          ("vx", "/FluidVelocity/FluidVelocityXAt%d" % k), ("vz", "/FluidVelocity/FluidVelocityZAt%d" % k)):
            e = rel_err(res[key][fl], o.field(name)[fl], scale=1e-4 if name[0] == "v" else None)
            assert e < 1e-9, (k, name, e)
    # restart from the checkpoint written after 12 steps
    again = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out2"), record_every=12, restart_from=sim.checkpoint_path)
    res2 = load_results(again.runRKColorGradient3D())
    for key in ("/FluidMacro/FluidDensityRin2", "/FluidVelocity/FluidVelocityZAt2"):
        assert np.array_equal(res2[key], res[key]), key


def test_rk3d_csf_cycle_restart_from_the_last_record(tmp_path):
    """the same [CyclesSetup] IsCycle = 'yes' branch with SurfaceTensionType = 'CSF': the record LastStep's densities and velocity, the top 20
    planes refilled with blue, populations = their equilibria (RKD2Q9.py:492-508 in 3-D) -- against the CSF oracle started from those fields"""
    import shutil
    from ini_fixtures import write_rk3d_csf
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D, duct
    from openlbmpm_amd.results import load_results
    from oracle.rk3dcsf import RK3DCSFOracle
    write_rk3d_csf(str(tmp_path), nx=18, ny=14, nz=60, steps=24, relax="MRT", sigma=0.05, theta=60.0)
    first = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out"), record_every=12)
    prev = load_results(first.runRKColorGradient3D())
    init = tmp_path / "LBMInitial"; init.mkdir()
    _set_cycle(str(tmp_path), 2)
    shutil.copy(first.result_path, str(init))
    sim = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "o2"), initial_dir=str(init), record_every=10)
    sim.timeSteps = 10
    res = load_results(sim.runRKColorGradient3D())
    dom = duct(18, 14, 60)
    rR, rB = prev["/FluidMacro/FluidDensityRin2"].copy(), prev["/FluidMacro/FluidDensityBin2"].copy()
    rR[-20:] = 0.0; rB[-20:] = np.where(dom[-20:] == 1, 1.0, 0.0)
    v = [prev["/FluidVelocity/FluidVelocity%sAt2" % ax] * (dom == 1) for ax in "XYZ"]
    par = dict(sigma=0.05, theta=60.0, wetting=2, beta=1.0, delta=0.98, tauR=1.0, tauB=0.9, tautype=2, relax="MRT", velocityZR=0.0, velocityZB=-1.0e-4,
               densityBL=1.0, densityRL=1.0e-8)
    o = RK3DCSFOracle(dom, rR, rB, par, velocity=v).run(10).step_a()
    fl = dom == 1
    for name, f in (("FluidMacro/FluidDensityRin1", "rhoR"), ("FluidMacro/FluidDensityBin1", "rhoB"), ("FluidVelocity/FluidVelocityZAt1", "vz")):
        assert rel_err(res["/" + name][fl], o.field(f)[fl], scale=1e-4 if f == "vz" else None) < 1e-9, name


def test_rk3d_csf_image_cycle_takes_the_populations_over_with_the_colours_swapped(tmp_path):
    """the image branch of IsCycle = 'yes' (RKD2Q9.py:532-556 in 3-D) with the CSF model: cycleInitialRK3D written from a finished run, taken over
    with the colours swapped in the top buffer planes; the run starts from exactly those populations (record 0 below the open planes)"""
    from ini_fixtures import write_rk3d_csf
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
    from openlbmpm_amd.geometry import porous_spheres, voxel_domain
    from openlbmpm_amd.results import load_results
    write_rk3d_csf(str(tmp_path), steps=30, relax="MRT", sigma=0.05, theta=60.0)
    vox = porous_spheres(40, 18, 30, porosity=0.7, rmin=2.0, rmax=5.0, seed=8, nbuf=0, walls=False)
    init = tmp_path / "LBMInitial"
    a = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "a"), domain=voxel_domain(vox, 6), record_every=30, num_buffering_layers=6)
    a.runRKColorGradient3D()
    assert a.solver.dominant_kernel == "csf3d_collide"
    path = a.write_cycle_initial(str(init))
    stored = load_results(path)
    fR, fB = a.solver.get_pdf()
    assert np.array_equal(stored["/FluidPDF/FluidPDFR"], fR) and fR.shape == a._domain.shape + (19,)
    _set_cycle(str(tmp_path), 0)
    b = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "b"), domain=a._domain, record_every=6, num_buffering_layers=6, initial_dir=str(init))
    b.timeSteps = 6
    b.initializeDomainBorder(); b.initializeDomainCondition()
    assert np.array_equal(b.fluidPDFR[:-6], fR[:-6]) and np.array_equal(b.fluidPDFB[-6:], fR[-6:]) and np.array_equal(b.fluidPDFR[-6:], fB[-6:])
    res = load_results(b.runRKColorGradient3D())
    z = slice(2, a._domain.shape[0] - 8)         # record 0 = the densities of the populations taken over (away from the open planes and the swapped buffer)
    want = stored["/FluidPDF/FluidPDFR"].sum(axis=-1)
    assert rel_err(res["/FluidMacro/FluidDensityRin0"][z], want[z]) < 1e-13
    assert np.all(np.isfinite(res["/FluidVelocity/FluidVelocityZAt1"]))
    assert res["/FluidMacro/FluidDensityBin0"][-4:-2].sum() < 1e-2 * res["/FluidMacro/FluidDensityRin0"][-4:-2].sum()


@pytest.mark.parametrize("ranks", [2, 3])
def test_rk3d_csf_driver_on_several_ranks_writes_the_single_process_record(tmp_path, ranks):
    """RKColorGradient3D with SurfaceTensionType = 'CSF' under torchrun (the ranks share this GPU, gloo): one slab per rank, three face messages
    per step (rk3dcsf.RK3DCSFDistributed); rank 0 writes ONE file equal to the single-process driver's bit for bit; the checkpoint the ranks
    wrote together continues a single-process run to the same last record"""
    import subprocess
    import sys
    from ini_fixtures import write_rk3d_csf
    from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
    from openlbmpm_amd.results import load_results
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    write_rk3d_csf(str(tmp_path), nx=14, ny=12, nz=40, steps=24, relax="MRT", sigma=0.05, theta=60.0)
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from openlbmpm_amd.RKColorGradientD3Q19 import RKColorGradient3D
dist.init_process_group("gloo")
torch.cuda.set_device(0)
sim = RKColorGradient3D(%r, output_dir=%r, record_every=12, checkpoint_every=12, device=0)
sim.runRKColorGradient3D()
assert sim.solver.solver.world == %d and sim.nzl < 40
dist.destroy_process_group()
''' % (root, str(tmp_path), str(tmp_path / "out2"), ranks))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=dict(os.environ), timeout=600)
    single = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out1"), record_every=12)
    ref = load_results(single.runRKColorGradient3D())
    files = sorted(os.listdir(tmp_path / "out2"))
    rec = [f for f in files if f.startswith("SimulationResultsRK3D.")]
    ck = [f for f in files if f.startswith("CheckpointRK3D.")]
    assert len(rec) == 1 and len(ck) == 1, files
    got = load_results(str(tmp_path / "out2" / rec[0]))
    assert set(got) == set(ref)
    for key in ref:
        assert np.array_equal(got[key], ref[key]), key
    again = RKColorGradient3D(str(tmp_path), output_dir=str(tmp_path / "out3"), record_every=12, restart_from=str(tmp_path / "out2" / ck[0]))
    res = load_results(again.runRKColorGradient3D())
    for key in ("/FluidMacro/FluidDensityRin2", "/FluidVelocity/FluidVelocityZAt2"):
        assert np.array_equal(res[key], ref[key]), key
