"""GPU tests of the kernel-level entry points added for the perturbation loop (AcceleratedRKGPU2D.py:125, :657, :1008,
:1125, :1169, :1272) and of the RKGPU2DBoundary.py drop-in module, against vectors produced by the real reference
kernels (tests/golden/gen/make_golden_rk_pert.py, make_golden_rkb.py).  Launches go through the numba-shaped shim with
the arguments picked BY THE REFERENCE KERNELS' PARAMETER NAMES (openlbmpm_amd/_kernel_specs.py) from one name -> array
table: a loop is a list of kernel names, not a re-typed driver."""
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, load_params, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-13
EXV = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); EYV = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
W9 = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)


@pytest.fixture(scope="module")
def rt():
    sys.path.insert(0, os.path.join(ROOT, "openlbmpm_amd", "dropin"))
    import _runtime
    yield _runtime
    sys.path.remove(os.path.join(ROOT, "openlbmpm_amd", "dropin"))


class Table(dict):
    """name -> device array / scalar, filled from host values; host(name) reads an array back"""

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


def test_rkgpu2dboundary_module_exports_the_sixteen_kernels(rt):
    sys.modules.pop("RKGPU2DBoundary", None)
    import RKGPU2DBoundary as B
    names = [n for n in dir(B) if isinstance(getattr(B, n), rt.Kernel)]
    assert len(names) == 16 and "convectiveAverageBoundaryGPU3" in names
    with pytest.raises(TypeError):
        B.ghostPointsConstantVelocityRK[(1, 1), (1, 1)](*([1] * 12))          # this module's variant takes 10 arguments


def test_each_kernel_of_rkgpu2dboundary(rt):
    d = np.load(os.path.join(GOLDEN, "rkb_kernels.npz"))
    ny, nx = d["isDomain"].shape
    N = int(d["fluidNodes"].size)
    scal = dict(totalNodes=N, nx=nx, ny=ny, xDim=128, specificVYR=float(d["vyR"]), specificVYB=float(d["vyB"]), constPHB=float(d["pHB"]),
                constPHR=float(d["pHR"]), constPLB=float(d["pLB"]), constPLR=float(d["pLR"]), constPL=float(d["pL"]), specificVY=float(d["vIn"]))

    def fresh():
        return Table(rt, fluidNodes=d["fluidNodes"], neighboringNodes=d["nbr"], fluidPDFR=d["fR"], fluidPDFB=d["fB"], fluidRhoR=d["rhoR"],
                     fluidRhoB=d["rhoB"], fluidPDFTotal=d["fT"], physicalVY=d["vy"], normalVelocity=d["vy"], fluidPDFROld=d["fROld"],
                     fluidPDFBOld=d["fBOld"], **scal)
    alias = dict(fR="fluidPDFR", fB="fluidPDFB", rhoR="fluidRhoR", rhoB="fluidRhoB", fT="fluidPDFTotal", vy="physicalVY", vn="normalVelocity")
    # chains as the generator ran them (each starts from the fresh state)
    chains = [("rkb", ["constantVelocityZHBoundaryHigherRK", "ghostPointsConstantVelocityRK"]),
              ("rkb", ["convectiveOutletGPU", "convectiveOutletGhost2GPU", "convectiveOutletGhost3GPU"]),
              ("rkb", ["convectiveAverageBoundaryGPU", "convectiveAverageBoundaryGPU2", "convectiveAverageBoundaryGPU3"]),
              ("rkb", ["calConstPressureInletGPU", "ghostPointsConstPressureInletRK"]),
              ("rkb", ["calConstPressureLowerGPU", "ghostPointsConstPressureLowerRK"]),
              ("rk", ["calConstPressureLowerGPU", "ghostPointsConstPressureLowerRK"]),          # the compact-index namesakes (A:1008, A:1045)
              ("rkb", ["calConstPressureHighGPU"]), ("rkb", ["constantVelocityZHBoundaryHigherNewRK"]),
              ("rk", ["constantVelocityZHBoundaryHigherNewRK"]), ("rkb", ["calConstPressureLowerGPUTotal"]), ("rkb", ["constantTotalVelocityInlet"])]
    checked = 0
    for mod, names in chains:
        t = fresh()
        for name in names:
            rt.launch_by_name(mod, name, t)
            key = ("A_" if mod == "rk" else "") + name
            for f in [k.split("__")[1] for k in d.files if k.startswith(key + "__")]:
                e = rel_err(t.host(alias[f]), d["%s__%s" % (key, f)])
                assert e < TOL, (mod, name, f, e)
                checked += 1
    assert checked >= 60
    # the two modules' pressure-outlet kernels really differ on this domain (solid column: compact row != grid row)
    assert not np.array_equal(d["A_calConstPressureLowerGPU__fR"], d["calConstPressureLowerGPU__fR"])


def test_each_kernel_of_the_perturbation_loop(rt):
    d = np.load(os.path.join(GOLDEN, "rkpert_kernels.npz"))
    ny, nx = d["isDomain"].shape
    N = int(d["fluidNodes"].size)
    zN9 = np.zeros((N, 9))
    base = dict(totalNodes=N, nx=nx, ny=ny, xDim=128, delta=0.98, tauR=float(d["tauR"]), tauB=float(d["tauB"]), unitEX=EXV, unitEY=EYV,
                constantCR=np.zeros(9), constantCB=np.zeros(9), weightsCoeff=W9, schemeGradient=np.ones(9), fluidNodes=d["fluidNodes"],
                neighboringNodes=d["nbr"], fluidRhoR=d["rhoR"], fluidRhoB=d["rhoB"], phiValue=d["phi"], physicalVX=d["vx"], physicalVY=d["vy"],
                collisionR1=zN9, collisionB1=zN9, CGX=np.zeros(nx * ny), CGY=np.zeros(nx * ny), transformationM=d["M"], inverseTM=d["Minv"],
                collisionS=d["S"], constantB=d["constantB"])
    t = Table(rt, fluidPDFR=d["fR"], fluidPDFB=d["fB"], **base).put(physicalVX=np.zeros(N), physicalVY=np.zeros(N))
    rt.launch_by_name("rk", "calPhysicalVelocityRKGPU2D", t)                                           # A:125
    assert np.array_equal(t.host("physicalVX"), d["vx"]) and np.array_equal(t.host("physicalVY"), d["vy"])
    rt.launch_by_name("rk", "calRKCollision1GPU2DSRTNew", t)                                           # A:1125
    assert rel_err(t.host("fluidPDFR"), d["col1_fR"]) < TOL and rel_err(t.host("fluidPDFB"), d["col1_fB"]) < TOL
    t.put(fluidPDFTotal=d["c23_in_fT"], betaCoeff=float(d["beta"]), AkR=float(d["AkR"]), AkB=float(d["AkB"]), solidPhi=float(d["solidPhi"]))
    rt.launch_by_name("rk", "calRKCollision23GPUNew", t)                                               # A:1169
    for f, key in (("fluidPDFTotal", "c23_fT"), ("fluidPDFR", "c23_fR"), ("fluidPDFB", "c23_fB")):
        assert rel_err(t.host(f), d[key]) < TOL, key
    t.put(fluidRhoR=np.full(N, 0.7), fluidRhoB=np.full(N, 0.3), fluidPDFTotal=d["fT"], solidPhi=0.4)      # zero-gradient branches
    rt.launch_by_name("rk", "calRKCollision23GPUNew", t)
    assert np.array_equal(t.host("fluidPDFTotal"), d["c23u_fT"]) and rel_err(t.host("fluidPDFR"), d["c23u_fR"]) < TOL
    for key, bf in (("mrt1_fT", (0., 0.)), ("mrt1_fT_force", tuple(d["bodyF"]))):                       # A:1272
        t.put(fluidRhoR=d["rhoR"], fluidRhoB=d["rhoB"], fluidPDFTotal=d["fT"], bodyFX=float(bf[0]), bodyFY=float(bf[1]))
        rt.launch_by_name("rk", "calRKCollision1GPU2DMRTNew", t)
        assert rel_err(t.host("fluidPDFTotal"), d[key]) < TOL, key
    t = Table(rt, fluidPDFR=d["fR"], fluidPDFB=d["fB"], **base).put(specificVYR=float(d["vyR"]), specificVYB=float(d["vyB"]),
                                                                      forceX=np.zeros(1), forceY=np.zeros(1))
    rt.launch_by_name("rk", "constantVelocityZHBoundaryHigherRK", t)                                   # A:657
    for f, key in (("fluidPDFR", "zh_fR"), ("fluidPDFB", "zh_fB"), ("fluidRhoR", "zh_rhoR"), ("fluidRhoB", "zh_rhoB")):
        assert rel_err(t.host(f), d[key]) < TOL, key
    rt.launch_by_name("rk", "ghostPointsConstantVelocityRK", t)                                        # A:607
    for f, key in (("fluidPDFR", "zhg_fR"), ("fluidPDFB", "zhg_fB"), ("fluidRhoR", "zhg_rhoR"), ("fluidRhoB", "zhg_rhoB")):
        assert rel_err(t.host(f), d[key]) < TOL, key
    t = Table(rt, fluidPDFR=d["fR"], fluidPDFB=d["fB"], **base).put(constPLB=float(d["pLB"]), constPLR=float(d["pLR"]))
    rt.launch_by_name("rk", "calConstPressureLowerGPU", t)                                             # A:1008
    for f, key in (("fluidPDFR", "pl_fR"), ("fluidPDFB", "pl_fB"), ("fluidRhoR", "pl_rhoR"), ("fluidRhoB", "pl_rhoB")):
        assert rel_err(t.host(f), d[key]) < TOL, key
    rt.launch_by_name("rk", "ghostPointsConstPressureLowerRK", t)                                      # A:1045
    for f, key in (("fluidPDFR", "plg_fR"), ("fluidPDFB", "plg_fB"), ("fluidRhoR", "plg_rhoR"), ("fluidRhoB", "plg_rhoB")):
        assert rel_err(t.host(f), d[key]) < TOL, key


# the perturbation loop as a list of launches (order of RKD2Q9.py:1046-1223 with the repairs of the golden generator: the sum of the
# colours is taken after collision 1 for SRT, just before it for MRT); (kernel, {parameter name of that kernel: name in the table})
def pert_loop(relax):
    R, Bq = dict(fluidPDF="fluidPDFR", fluidPDFNew="fluidPDFRNew"), dict(fluidPDF="fluidPDFB", fluidPDFNew="fluidPDFBNew")
    seq = [("calStreaming1GPU", R), ("calStreaming1GPU", Bq), ("calStreaming2GPU", R), ("calStreaming2GPU", Bq),
           ("calConstPressureLowerGPU", {}), ("ghostPointsConstPressureLowerRK", {}),
           ("constantVelocityZHBoundaryHigherRK", {}), ("ghostPointsConstantVelocityRK", {}),
           ("calMacroDensityRKGPU2D", {}), ("calPhysicalVelocityRKGPU2D", {}), ("calPhaseFieldPhi", {})]
    if relax == "MRT":
        seq += [("calTotalFluidPDF", {}), ("calRKCollision1GPU2DMRTNew", {})]
    else:
        seq += [("calRKCollision1GPU2DSRTNew", {}), ("calTotalFluidPDF", {})]
    return seq + [("calRKCollision23GPUNew", {})]


@pytest.mark.parametrize("name", ["srt_capillary", "srt_porous", "mrt_capillary"])
def test_perturbation_loop_through_the_dropin_kernels(rt, name):
    from openlbmpm_amd._kernel_specs import KERNELS
    d = np.load(os.path.join(GOLDEN, "rkpert_%s.npz" % name))
    p = load_params(d)
    ny, nx = d["isDomain"].shape
    N = int(d["fluidNodes"].size)
    M = d["M"] if "M" in d.files else np.zeros((9, 9))
    t = Table(rt, totalNodes=N, totalNum=N, nx=nx, ny=ny, xDim=128, delta=p["delta"], tauR=p["tauR"], tauB=p["tauB"], bodyFX=0.0, bodyFY=0.0,
              betaCoeff=p["beta"], AkR=float(d["AkR"]), AkB=float(d["AkB"]), solidPhi=float(d["solidPhi"]), specificVYR=p["vyR"], specificVYB=p["vyB"],
              constPLB=p["rhoBL"], constPLR=p["rhoRL"], unitEX=EXV, unitEY=EYV, constantCR=np.zeros(9), constantCB=np.zeros(9), weightsCoeff=W9,
              schemeGradient=np.ones(9), constantB=d["constantB"], fluidNodes=d["fluidNodes"], neighboringNodes=d["neighboringNodes"],
              fluidPDFR=d["init_fR"], fluidPDFB=d["init_fB"], fluidPDFRNew=np.zeros((N, 9)), fluidPDFBNew=np.zeros((N, 9)),
              fluidPDFTotal=np.zeros((N, 9)), fluidRhoR=d["init_fR"].sum(axis=1), fluidRhoB=d["init_fB"].sum(axis=1), phiValue=np.zeros(N),
              physicalVX=np.zeros(N), physicalVY=np.zeros(N), collisionR1=np.zeros((N, 9)), collisionB1=np.zeros((N, 9)), CGX=np.zeros(nx * ny),
              CGY=np.zeros(nx * ny), forceX=np.zeros(1), forceY=np.zeros(1), transformationM=M, inverseTM=d["Minv"] if "Minv" in d.files else M,
              collisionS=d["S"] if "S" in d.files else np.zeros(9))
    seq = pert_loop(p["relax"])
    fields = dict(fR="fluidPDFR", fB="fluidPDFB", fTot="fluidPDFTotal", rhoR="fluidRhoR", rhoB="fluidRhoB", phi="phiValue", vx="physicalVX", vy="physicalVY")
    done = 0
    for k in d["snaps"]:
        for _ in range(int(k) - done):
            for kern, rename in seq:
                names = KERNELS[("rk", kern)][2]
                rt.launch_by_name("rk", kern, {a: t[rename.get(a, a)] for a in names})
        done = int(k)
        for f, dev in fields.items():
            e = rel_err(t.host(dev), d["s%d_%s" % (k, f)])
            assert e < 1e-11, (name, int(k), f, e)
