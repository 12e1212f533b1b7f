"""GPU tests of the kernel-level drop-in path (include/lbmpm_kernels.h through the numba.cuda-shaped shim in
openlbmpm_amd/dropin) against the golden vectors the real reference produced.

A time loop here is DATA: a list of (kernel name, {kernel parameter: table entry}) over one table of device arrays and
scalars; the launcher picks each kernel's arguments by the reference kernel's own parameter names
(openlbmpm_amd/_kernel_specs.py), so no driver text is re-typed.  The sequences are those of the working reference loops:
    colour gradient CSF   RKD2Q9.py:1295-1490          explicit forcing   ShanChenD2Q9.py:1714-2087
    original Shan-Chen    ShanChenD2Q9.py:1492-1629
One test keeps a handful of literal `kernel[grid, block](...)` statements: that call form is the drop-in interface."""
import math
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, load_params, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXV = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); EYV = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
W9 = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)


@pytest.fixture(scope="module")
def rt():
    sys.path.insert(0, os.path.join(ROOT, "openlbmpm_amd", "dropin"))
    for m in ("numba", "numba.cuda"):
        sys.modules.pop(m, None)
    import _runtime
    yield _runtime
    sys.path.remove(os.path.join(ROOT, "openlbmpm_amd", "dropin"))
    for m in ("numba", "numba.cuda"):
        sys.modules.pop(m, None)


class Table(dict):
    def __init__(self, rt, **kw):
        dict.__init__(self)
        self.rt = rt
        self.put(**kw)

    def put(self, **kw):
        for k, v in kw.items():
            self[k] = self.rt.to_device(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        return self

    def host(self, name):
        return self[name].copy_to_host()


def run(rt, module, table, sequence):
    from openlbmpm_amd._kernel_specs import KERNELS
    for kern, rename in sequence:
        rt.launch_by_name(module, kern, {a: table[rename.get(a, a)] for a in KERNELS[(module, kern)][2]})


def check(table, golden, prefix, fields, tol, what):
    for key, entry in fields.items():
        e = rel_err(table.host(entry), golden[prefix + key])
        assert e < tol, "%s %s%s rel err %.3e" % (what, prefix, key, e)


def test_numba_call_form_and_signature_checks(rt):
    """`kernel[grid, block](*args)` with to_device / copy_to_host is the interface a reference driver uses"""
    from numba import cuda
    import AcceleratedRKGPU2D as RKGPU2D
    assert cuda.is_available()
    grid1D, threadPerBlock1D = (4, 1), (32, 1)
    rhoR = np.linspace(0.2, 1.0, 100); rhoB = np.linspace(0.9, 0.1, 100)
    deviceFluidRhoR = cuda.to_device(rhoR); deviceFluidRhoB = cuda.to_device(rhoB)
    deviceColorValue = cuda.device_array_like(rhoR)
    RKGPU2D.calPhaseFieldPhi[grid1D, threadPerBlock1D](100, 128, deviceFluidRhoR, deviceFluidRhoB, deviceColorValue)
    assert np.array_equal(deviceColorValue.copy_to_host(), (rhoR - rhoB) / (rhoR + rhoB))
    with pytest.raises(TypeError):       # wrong argument count, like Numba's explicit signatures
        RKGPU2D.calPhaseFieldPhi[grid1D, threadPerBlock1D](100, 128, deviceFluidRhoR, deviceFluidRhoB)
    with pytest.raises(TypeError):       # host array where a device array is expected
        RKGPU2D.calPhaseFieldPhi[grid1D, threadPerBlock1D](100, 128, rhoR, deviceFluidRhoB, deviceColorValue)


# ------------------------------------------------------------------------------------------------ colour gradient, CSF
def csf_sequence(par, has_wetting):
    R = dict(fluidPDF="fluidPDFR", fluidPDFNew="fluidPDFRNew"); Bq = dict(fluidPDF="fluidPDFB", fluidPDFNew="fluidPDFBNew")
    grad = dict(colorValueFluid="ColorValue", colorValueSolid="colorValueSolid")
    coll = dict(colorValue="ColorValue", fluidTotalPDF="fluidPDFTotal")
    seq = []
    if par["inlet"] == "Neumann":
        seq += [("constantTotalVelocityInlet", {}), ("ghostPointsConstantVelocityRK", {})]
    else:
        seq += [("calConstPressureInletGPU", {}), ("ghostPointsConstPressureInletRK", {})]
    if par["outlet"] == "Convective":
        seq += [("convectiveOutletGPU", {}), ("convectiveOutletGhost2GPU", {}), ("convectiveOutletGhost3GPU", {})]
    else:
        seq += [("calConstPressureLowerGPUTotal", {}), ("ghostPointsConstPressureLowerRK", {})]
    seq += [("calTotalFluidPDF", {}), ("calPhysicalVelocityRKGPU2DNew1", {}), ("calPhaseFieldPhi", dict(phiValue="ColorValue")),
            ("calColorValueOnSolid", dict(grad, xDim="xDim")), ("calRKInitialGradient", grad)]
    if has_wetting:
        seq += [("updateColorGradientOnWetting" if par["wetting"] == 1 else "updateColorGradientOnWettingNew", {})]
    seq += [("calForceTermInColorGradient2D" if par["wetting"] == 1 else "calForceTermInColorGradientNew2D", {})]
    if par["relax"] == "SRT":
        seq += [("calRKCollision1TotalGPU2DSRTM", {}), ("calPerturbationFromForce2D", coll)]
    else:
        seq += [("calRKCollision1TotalGPU2DMRTM", {}), ("calPerturbationFromForce2DMRT", coll)]
    seq += [("calRecoloringProcessM", {}), ("calStreaming1GPU", R), ("calStreaming1GPU", Bq), ("calStreaming2GPU", R), ("calStreaming2GPU", Bq),
            ("calTotalFluidPDF", {}), ("calMacroDensityRKGPU2D", {})]
    return seq


@pytest.mark.parametrize("scenario", ["csf_mrt_capillary", "csf_mrt_convective", "csf_mrt_pinlet", "csf_mrt_wetting1", "csf_srt_capillary"])
def test_colour_gradient_loop(rt, scenario):
    """velocity / pressure inlet, pressure / convective outlet, wetting type 1 / 2, SRT / MRT: one capture of the real driver each"""
    from oracle.rk import RKOracle, mrt_matrices
    d = np.load(os.path.join(GOLDEN, "rk_%s.npz" % scenario))
    par = load_params(d)
    o = RKOracle(d["isDomain"], par)                           # host set-up only (compaction tables, normals, initial fields)
    assert np.array_equal(o.fluidNodes, d["fluidNodes"])
    N = o.N
    M, Minv, S = mrt_matrices()
    th = par["theta"] / 180. * np.pi
    z = lambda *s: np.zeros(s)
    t = Table(rt, totalNodes=N, totalNum=N, totalSolidWetting=o.W, totalWettingNodes=o.W, numColorSolid=o.W, totalFluidWettingNodes=o.Wf,
              nx=par["nx"], ny=par["ny"], xDim=128, fluidNodes=o.fluidNodes, domainNewIndex=o.newIndex, wettingNodes=o.wettingSolidNodes,
              neighboringNodes=np.zeros(8 * N, dtype=np.int64), neighboringWettingNodes=np.zeros(8 * max(o.W, 1), dtype=np.int64),
              fluidRhoR=o.rhoR, fluidRhoB=o.rhoB, fluidPDFR=o.fR, fluidPDFB=o.fB, fluidPDFRNew=z(N, 9), fluidPDFBNew=z(N, 9),
              fluidPDFTotal=o.fR + o.fB, physicalVX=z(N), physicalVY=z(N), ColorValue=z(N), colorValueSolid=z(max(o.W, 1)),
              forceX=z(N), forceY=z(N), gradientX=z(N), gradientY=z(N), KValue=o.rhoB, fluidNodesWetting=o.fluidWet,
              unitVectorNsx=o.nsx, unitVectorNsy=o.nsy, weightsCoeff=W9, unitEX=EXV, unitEY=EYV, transformationM=M, inverseTM=Minv, collisionS=S,
              cosTheta=float(np.cos(th)), sinTheta=float(np.sin(th)), specificVY=par["vyB"] + par["vyR"], constPL=par["rhoBL"] + par["rhoRL"],
              constPHB=par["rhoBH"], constPHR=par["rhoRH"], surfaceTension=par["sigma"], optionF=par["tautype"], tauR=par["tauR"],
              tauB=par["tauB"], deltaValue=par["delta"], betaValue=par["beta"])
    run(rt, "rk", t, [("fillNeighboringNodes", {})])                       # RKD2Q9.py:709-716
    assert np.array_equal(t.host("neighboringNodes"), d["neighboringNodes"])
    if o.W:
        run(rt, "rk", t, [("fillNeighboringWettingNodes", {})])
        assert np.array_equal(t.host("neighboringWettingNodes"), d["neighboringWettingSolidNodes"])
    t["neighboringWettingSolid"] = t["neighboringWettingNodes"]
    seq = csf_sequence(par, o.Wf > 0)
    fields = dict(fR="fluidPDFR", fB="fluidPDFB", rhoR="fluidRhoR", rhoB="fluidRhoB", vx="physicalVX", vy="physicalVY", phi="ColorValue",
                  Gx="gradientX", Gy="gradientY", Fx="forceX", Fy="forceY", K="KValue")
    snaps = [int(k) for k in d["snaps"] if int(k) <= 50]
    for step in range(1, max(snaps) + 1):
        run(rt, "rk", t, seq)
        if step in snaps:
            check(t, d, "s%d_" % step, fields, 1e-11, "%s step %d" % (scenario, step))


# ------------------------------------------------------------------------------------------------ Shan-Chen family
def sc_table(rt, g, par, with_iso):
    from oracle.sc import simple_geometry, initial_densities
    nx, ny = par["nx"], par["ny"]
    dom = simple_geometry(nx, ny)
    fluidNodes = np.flatnonzero(dom.reshape(-1) == 1).astype(np.int64)
    assert np.array_equal(fluidNodes, g["fluidNodes"])
    N = fluidNodes.size
    newIndex = -np.ones(nx * ny, dtype=np.int64); newIndex[fluidNodes] = np.arange(N)
    rho0 = np.ascontiguousarray(initial_densities(dom, False, par).reshape(2, -1)[:, fluidNodes])
    f0 = np.ascontiguousarray(W9[None, None, :] * rho0[:, :, None])
    scheme = int(par.get("scheme", 4))
    weights = {4: [1. / 3.] * 4 + [1. / 12.] * 4,
               8: [4. / 21.] * 4 + [4. / 45.] * 4 + [1. / 60.] * 4 + [1. / 5040.] * 4 + [2. / 315.] * 8,
               10: [262. / 1785.] * 4 + [93. / 1190.] * 4 + [7. / 340.] * 4 + [9. / 9520.] * 4 + [6. / 595.] * 8 +
                   [2. / 5355.] * 4 + [1. / 7140.] * 8}[scheme]          # ShanChenD2Q9.py:1675-1689
    tau = np.array([par["tau0"], par["tau1"]])
    z = lambda *s: np.zeros(s)
    t = Table(rt, totalNodes=N, totalNum=N, numFluids=2, nx=nx, ny=ny, xDim=256, fluidNodes=fluidNodes, domainNewIndex=newIndex,
              neighboringNodes=np.zeros(8 * N, dtype=np.int64), fluidPDF=f0, fluidPDFOld=f0, fluidPDFNew=f0, fluidRho=rho0,
              fluidPotential=z(2, N), forceX=z(2, N), forceY=z(2, N), velocityPX=z(N), velocityPY=z(N), eqVX=z(N), eqVY=z(N),
              fEq=f0, fForce=z(2, N, 9), tau=tau, interactionCoeff=np.array([[0., par["G"]], [par["G"], 0.]]),
              interactionSolid=np.array([par["Gs0"], par["Gs1"]]), interCoeff=np.array([[0., par["G"]], [par["G"], 0.]]),
              interSolid=np.array([par["Gs0"], par["Gs1"]]), EX=EXV, EY=EYV, weightCoeff=W9, weightsCoeff=W9, weightInter=np.array(weights),
              specificVY=np.array([par["vy0"], par["vy1"]]), densityL=1.002, conserveS=np.ones(2), primeVX=z(N), primeVY=z(N))
    t["physicalVY"] = t["velocityPY"]; t["equilibriumVX"] = t["eqVX"]; t["equilibriumVY"] = t["eqVY"]
    run(rt, "sc", t, [("fillNeighboringNodes", {})])
    assert np.array_equal(t.host("neighboringNodes"), g["neighboringNodes"])
    if with_iso and scheme in (8, 10):
        t.put(isoNodes=np.zeros((24 if scheme == 8 else 36) * N, dtype=np.int64))
        run(rt, "sc", t, [("fillNeighboringNodesISO8" if scheme == 8 else "fillNeighboringNodesISO10", dict(neighboringNodes="isoNodes"))])
    return t, N, f0, tau, scheme


@pytest.mark.parametrize("scenario", ["efs_srt_dirichlet", "efs_mrt_dirichlet", "efs_srt_convective", "efs_srt_iso8", "efs_srt_iso10",
                                      "efs_srt_freeflow", "efs_srt_chang"])
def test_explicit_forcing_loop(rt, scenario):
    from oracle.sc import collision_matrices
    g = np.load(os.path.join(GOLDEN, "sc_%s.npz" % scenario))
    par = load_params(g)
    t, N, f0, tau, scheme = sc_table(rt, g, par, True)
    mrt = par["relax"] == "MRT"
    if mrt:
        t.put(collisionMatrix=collision_matrices(tau), fForceM=np.zeros_like(f0), fluidPDFM=f0)
    iso = dict(neighboringNodes="isoNodes")
    force = {4: ("calExplicit4thOrderScheme", {}), 8: ("calExplicit8thOrderScheme", iso), 10: ("calExplicit10thOrderScheme", iso)}[scheme]
    chain = [("calFluidPotentialGPUEql", {}), force,
             ("transformEquilibriumVelocity", {}) if mrt else ("calEquilibriumVEFGPU", {}),
             ("calEquilibriumFuncEFGPU", {}), ("calForceDistrGPU", {})]
    chang = ("calVelocityBoundaryHigherChangGPU", dict(fluidPDFOld="fluidPDFOld", fluidPDFNew="fluidPDF"))              # S:1803-1808, :1999-2006
    inlet = {4: [chang if par.get("method") == "Chang" else ("constantVelocityZouHeBoundaryHigher", {}), ("ghostPointsConstantVelocityInlet", {})],   # S:1794-1825, :1989-2020
             8: [("constantVelocityZouHeBoundaryHigher8", {}), ("ghostPointsConstantVelocity8", {}), ("ghostPointsConstantVelocity82", {})],
             10: []}[scheme]
    outlet_p = {4: [("constantPressureZouHeBoundaryLower", {}), ("ghostPointsConstantPressureOutlet", {})],         # S:1826-1849, :1931-1953
                8: [("constantPressureZouHeBoundaryLower8", {}), ("ghostPointsConstantPressureOutlet8", {}), ("ghostPointsConstantPressureOutlet82", {})],
                10: []}[scheme]
    onto = dict(fluidPDFNew="fluidPDF")          # (the outlet kernels call their target array fluidPDFNew; the loops pass the main one)
    outlet = ([("convectiveOutletEachGPU", onto), ("convectiveOutletEach2GPU", onto), ("convectiveOutletEach3GPU", onto)]
              if par["outlet"] == "Convective" else outlet_p if par["outlet"] == "Dirichlet" else [])
    # 'Freeflow' (S:1865-1884): rows 2, 1, 0 take f-bar, F_i, f_eq and rho of the row above, between the moment transforms and the collision
    freeflow = [(k, onto) for k in ("convectiveOutletGPUEFS", "convectiveOutletGhost2GPUEFS", "convectiveOutletGhost3GPUEFS")] if par["outlet"] == "Freeflow" else []
    collide = ([("transfromForceTerm", {}), ("transformPDFandEquil", {})] + freeflow + [("calAfterCollisionMRT", {})] if mrt
               else freeflow + [("calCollisionEXGPU", {})])
    macro = [("calFluidRhoGPU", {}), ("calPhysicalVelocity", {})]
    # before the loop, ShanChenD2Q9.py:1714-1849
    run(rt, "sc", t, chain + [("transformPDFGPU", {})] + inlet + (outlet_p if par["outlet"] == "Dirichlet" else []))
    loop = [("savePDFLastStep", {})] + collide + [("calStreaming1GPU", {}), ("calStreaming2GPU", {})] + macro + outlet + inlet + macro + chain   # S:1852-2087
    fields = dict(f="fluidPDF", rho="fluidRho", Fx="forceX", Fy="forceY", vx="velocityPX", vy="velocityPY", ueqx="eqVX", ueqy="eqVY", feq="fEq",
                  fforce="fForce")
    snaps = [int(k) for k in g["snaps"] if int(k) <= 10]
    for i in range(max(snaps) + 1):
        run(rt, "sc", t, loop)
        if i in snaps:
            check(t, g, "s%d_" % i, fields, 1e-11, "%s pass %d" % (scenario, i))


@pytest.mark.parametrize("scenario", ["sc_srt_convective", "sc_srt_chang"])
def test_original_shan_chen_loop(rt, scenario):
    """runOptimizedLBM (Neumann / Zou-He inlet, convective outlet): the fused interaction + collision kernel, the three
    outlet-row kernels and the (result-less) whole-fluid velocity"""
    g = np.load(os.path.join(GOLDEN, "sc_%s.npz" % scenario))
    par = load_params(g)
    t, N, f0, tau, _ = sc_table(rt, g, par, False)
    assert rel_err(f0, g["init_f"]) < 1e-15
    t.put(weightInter=np.array([1. / 9.] * 4 + [1. / 36.] * 4))                 # ShanChenD2Q9.py:1478
    inlet = (("calVelocityBoundaryHigherChangGPU", dict(fluidPDFOld="fluidPDFOld", fluidPDFNew="fluidPDF")) if par.get("method") == "Chang"      # S:1529
             else ("constantVelocityZouHeBoundaryHigher", {}))
    head = [inlet, ("ghostPointsConstantVelocityInlet", {}), ("savePDFLastStep", {}), ("calMacroWholeVelocity", {})]
    tail = [("calFluidRhoGPU", {}), ("calFluidPotentialGPUEql", {}), ("interactionCollisionProcess", {}), ("calStreaming1GPU", {}),
            ("calStreaming2GPU", {}), ("convectiveOutletGPU", dict(fluidPDFNew="fluidPDF")), ("convectiveOutletGhost2GPU", dict(fluidPDFNew="fluidPDF")),
            ("convectiveOutletGhost3GPU", dict(fluidPDFNew="fluidPDF")),
            ("calFluidRhoGPU", {}), ("calPhysicalVelocity", {})]
    fields = dict(f="fluidPDF", rho="fluidRho", Fx="forceX", Fy="forceY", vx="velocityPX", vy="velocityPY")
    snaps = [int(k) for k in g["snaps"]]
    for step in range(1, max(snaps) + 1):
        run(rt, "sc", t, head)
        if step == 3:             # the kernel's own formula (O:345-356); nothing downstream reads these arrays
            f, r = t.host("fluidPDF"), t.host("fluidRho")
            mx = sum((f[k, :, 1] - f[k, :, 3] + f[k, :, 5] - f[k, :, 6] - f[k, :, 7] + f[k, :, 8]) / tau[k] for k in range(2))
            my = sum((f[k, :, 2] - f[k, :, 4] + f[k, :, 5] + f[k, :, 6] - f[k, :, 7] - f[k, :, 8]) / tau[k] for k in range(2))
            rt_ = sum(r[k] / tau[k] for k in range(2))
            assert rel_err(t.host("primeVX"), mx / rt_) < 1e-12 and rel_err(t.host("primeVY"), my / rt_) < 1e-12
        run(rt, "sc", t, tail)
        if step in snaps:
            check(t, g, "s%d_" % step, fields, 1e-11, "%s step %d" % (scenario, step))


# ------------------------------------------------------------------------------------------------ tracer kernels
def test_tracer_kernels_against_reference_vectors(rt):
    d = np.load(os.path.join(GOLDEN, "tr_kernels.npz"))
    N = int(d["fluidNodes"].size); ny, nx = d["isDomain"].shape
    newidx = -np.ones((ny, nx), dtype=np.int64); newidx.reshape(-1)[d["fluidNodes"]] = np.arange(N)
    t = Table(rt, totalNodes=N, totalNum=N, nx=nx, ny=ny, xDim=128, numTracers=2, numSchemes=5, fluidNodes=d["fluidNodes"], domainNewIndex=newidx,
              neighboringNodes=np.zeros(4 * N, dtype=np.int64), tracerConc=np.zeros((2, N)), tracerPDF=d["conc_in_g"],
              unitVX=np.array([0., 1., -1, 0., 0.]), unitVY=np.array([0., 0., 0., 1., -1.]), velocityVX=d["col_vx"], velocityVY=d["col_vy"],
              transportM=d["M"], inverseRelaxationMS=d["A"], weightsCoeff=d["w"], critiriaValue=0.5, valueTransportDomain=np.zeros(N),
              fluidRhoR=d["ind_rhoR"], betaTracer=d["itf_beta"], unitEX=EXV, unitEY=EYV, gradientX=d["itf_Gx"], gradientY=d["itf_Gy"],
              tracerPDFNew=np.zeros((2, N, 5)), concBoundary=d["ina_cb"])
    t["neighboringTRNodes"] = t["neighboringNodes"]
    steps = [("fillNeighboringNodesTransport", ("neighboringNodes", "nbr", 0)), ("calConcentrationGPU", ("tracerConc", "conc_out", 1e-13)),
             ("calCollisionTransportLinearEqlMRTGPU", ("tracerPDF", "col_out_g", 1e-13)), ("calValueTransportDomain", ("valueTransportDomain", "ind_out", 0)),
             ("calTransportWithInterfaceD2Q5", ("tracerPDF", "itf_out_g", 1e-13)), ("calFreeConcBoundary3", ("tracerPDF", "free_out_g", 0)),
             ("calStreamingTransportGPU", None), ("calStreamingTransport2GPU", ("tracerPDF", "str_out_g", 0)),
             ("calInamuroConstConcBoundary", ("tracerPDF", "ina_out_g", 1e-13))]
    for kern, expect in steps:
        run(rt, "tr", t, [(kern, {})])
        if expect:
            got, want = t.host(expect[0]), d[expect[1]]
            assert (np.array_equal(got, want) if expect[2] == 0 else rel_err(got, want) < expect[2]), kern
    t.put(numTracers=3, reactionRate=d["rea_rate"], diffJcoeffs=d["rea_J"], tracerConc=d["rea_conc"], tracerPDF=d["rea_in_g"])
    run(rt, "tr", t, [("calReactionTracersGPU", {})])                      # A + B -> C between three tracers
    assert rel_err(t.host("tracerPDF"), d["rea_out_g"]) < 1e-13
