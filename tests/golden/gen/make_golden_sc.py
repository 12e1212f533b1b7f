#!/usr/bin/env python3
"""Generate Shan-Chen golden vectors by running the REAL reference drivers
(ShanChenD2Q9.runOptimizedEFLBM -- explicit forcing, ShanChenD2Q9.py:1631-2087 -- and
ShanChenD2Q9.runOptimizedLBM -- original Shan-Chen, :1433-1629) under the numba stand-in.

Container-only (needs /root/reference).  Usage:
    python tests/golden/gen/make_golden_sc.py [scenario ...]
Writes tests/golden/sc_<scenario>.npz (geometry, tables, initial state, end-of-iteration
snapshots of every device array that carries state).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

OUT = os.environ.get("LBMPM_GOLDEN_OUT") or os.path.dirname(HERE)      # (redirected by tests/test_golden_provenance.py)

BASIC_INI = """[Scheme]
Type = 'SRT'
[Geometry]
length = 1.0
width = 1.0
nx = {nx}
ny = {ny}
[Time]
TimeLength = 1.0
TimeStep = 1.0
[InitialCondition]
VelocityXLB = 0.0
VelocityYLB = 0.0
[BodyForce]
gValue = 0.0
[FlowDomain]
xDomain = 0,{nx}
yDomain = 0,{ny}
"""

TWOPHASE_INI = """[PictureSetup]
Exist = '{image}'
[SeparationBorder]
xGrid = {nx}
yGrid = {ny}
[FluidsTypes]
NumberOfFluids = 2
[InterType]
InteractionType = '{inter}'
[Parallelism]
Parallel = 'yes'
xDimension = 256
ThreadsNum = 32
[RelaxationType]
Type = '{relax}'
[DuplicateDomain]
Option = 'no'
[DICycles]
Option = 'no'
LastStep = 1105
"""

MODEL_INI = """[FluidProperties]
InitialDensities = {rho0},{rho1}
BackgroundDensities = {bg0},{bg1}
FluidsTau = {tau0},{tau1}
[{section}]
interactionFluid = {G}
interactionSolid = {Gs0},{Gs1}
potentialType = 'Simple'
[BoundaryDefinition]
BoundaryTypeInlet = 'Neumann'
BoundaryMethod = '{method}'
BoundaryTypeOutlet = '{outlet}'
[VelocityBoundary]
velocityX = 0.0,0.0
velocityY = {vy0},{vy1}
[PressureBoundary]
PressureInlet = 0.0, 0.0
PressureOutlet = 1.0, 0.0
[ForceScheme]
ExplicitScheme = {scheme}
[BodyForce]
Option = 'no'
forceXG = 0.0
forceYG = 0.0
[Time]
numberTimeStep = {steps}
"""

DEFAULTS = dict(image='no', nx=20, ny=48, inter='EFS', relax='SRT', rho0=1.0, rho1=1.0, bg0=0.02, bg1=0.02,
                tau0=1.0, tau1=1.0, G=0.20, Gs0=-0.14, Gs1=0.14, outlet='Dirichlet', method='ZouHe', vy0=0.0, vy1=-5.03e-4,
                steps=60, scheme=4)


def porous_image(nx, ny, seed, n_discs, rmin, rmax):
    rng = np.random.default_rng(seed)
    img = np.full((ny, nx), 255.0)
    yy, xx = np.mgrid[0:ny, 0:nx]
    for _ in range(n_discs):
        cx = rng.uniform(0, nx); cy = rng.uniform(0, ny); r = rng.uniform(rmin, rmax)
        img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0.0
    img[0, 0] = 0.0
    img[-1, -1] = 0.0
    return img


SCENARIOS = {
    "efs_srt_dirichlet": (dict(steps=80), (0, 1, 10, 80), None),
    "efs_mrt_dirichlet": (dict(steps=80, relax='MRT', tau0=1.0, tau1=0.8), (0, 1, 80), None),
    "efs_srt_convective": (dict(steps=60, outlet='Convective', tau0=0.9, tau1=1.1), (0, 1, 60), None),
    "efs_mrt_porous": (dict(steps=60, relax='MRT', image='yes'), (0, 1, 60),
                       dict(nx=34, ny=44, seed=5, n_discs=9, rmin=2.0, rmax=4.5)),
    # higher-isotropy force stencils (efs2D.ini [ForceScheme] ExplicitScheme = 8 | 10): scheme 8 moves the
    # boundary rows one row inwards and keeps two ghost rows; scheme 10 has no boundary kernels in the loop
    "efs_srt_iso8": (dict(steps=60, scheme=8), (0, 1, 60), None),
    "efs_mrt_iso8_porous": (dict(steps=60, scheme=8, relax='MRT', tau0=1.0, tau1=0.8, image='yes'), (0, 1, 60),
                            dict(nx=34, ny=44, seed=5, n_discs=9, rmin=2.0, rmax=4.5)),
    "efs_srt_iso10": (dict(steps=40, scheme=10), (0, 1, 40), None),
    # alternates of the explicit-forcing loop: the 'Freeflow' outlet (S:1865-1884: rows 2, 1, 0 take f-bar, F_i, f_eq and rho of
    # the row above BEFORE the collision) and Chang's velocity inlet (S:1999-2006)
    "efs_srt_freeflow": (dict(steps=60, outlet='Freeflow', tau0=0.9, tau1=1.1), (0, 1, 60), None),
    "efs_mrt_freeflow": (dict(steps=60, outlet='Freeflow', relax='MRT', tau0=1.0, tau1=0.8), (0, 1, 60), None),
    "efs_srt_chang": (dict(steps=60, method='Chang'), (0, 1, 60), None),
    # the 'Freeflow' rows are not scheme dependent in the loop (S:1865): with the wider force stencils too
    "efs_srt_iso8_freeflow": (dict(steps=40, scheme=8, outlet='Freeflow'), (0, 1, 40), None),
    "efs_srt_iso10_freeflow": (dict(steps=40, scheme=10, outlet='Freeflow'), (0, 1, 40), None),
    "sc_srt_convective": (dict(steps=80, inter='ShanChen', G=3.8, Gs0=-0.40, Gs1=0.40, bg0=0.06, bg1=0.06,
                               outlet='Convective', vy1=-1.01e-3), (1, 2, 10, 80), None),
    "sc_srt_chang": (dict(steps=60, inter='ShanChen', G=3.8, Gs0=-0.40, Gs1=0.40, bg0=0.06, bg1=0.06, method='Chang',
                          outlet='Convective', vy1=-1.01e-3), (1, 2, 10, 60), None),
    "sc_srt_porous": (dict(steps=60, inter='ShanChen', G=2.6, Gs0=-0.20, Gs1=0.20, bg0=0.15, bg1=0.15,
                           outlet='Convective', vy1=-1.01e-3, tau0=1.0, tau1=0.9, image='yes'), (1, 60),
                      dict(nx=34, ny=44, seed=9, n_discs=9, rmin=2.0, rmax=4.5)),
}


def run(name):
    overrides, snaps, image = SCENARIOS[name]
    par = dict(DEFAULTS); par.update(overrides)
    cuda = refenv.setup()
    import importlib
    import scipy.ndimage as sciimage
    img = None
    if image is not None:
        img = porous_image(**image)
        sciimage.imread = lambda path, flatten=True: np.array(img, copy=True)
    efs = par["inter"] == "EFS"
    files = {"basicsetup.ini": BASIC_INI.format(**par), "twophasesetup.ini": TWOPHASE_INI.format(**par)}
    files["efs2D.ini" if efs else "shanchen2D.ini"] = MODEL_INI.format(
        section="EFSParameters" if efs else "ShanChenParameters", **par)
    inidir = refenv.write_ini_dir(files)
    SC = importlib.import_module("ShanChenD2Q9")
    sim = SC.ShanChenD2Q9(inidir)

    state = {"iter": -1, "last": {}}
    out = {}
    cp = lambda a: np.array(a, copy=True).view(np.ndarray)

    def snapshot(k):
        L = state["last"]
        v = L["calPhysicalVelocity"]          # (N, nF, xDim, pdf, rho, Fx, Fy, vx, vy)
        rec = dict(f=v[3], rho=v[4], Fx=v[5], Fy=v[6], vx=v[7], vy=v[8])
        if efs:
            rec.update(ueqx=L["calEquilibriumFuncEFGPU"][7], ueqy=L["calEquilibriumFuncEFGPU"][8],
                       feq=L["calEquilibriumFuncEFGPU"][9], fforce=L["calForceDistrGPU"][11])
        for key, val in rec.items():
            out["s%d_%s" % (k, key)] = cp(val)

    def post(kname, args):
        state["last"][kname] = args
        if efs:
            # an iteration of the EFS loop ends with calForceDistrGPU (ShanChenD2Q9.py:2084);
            # the first call is the pre-loop one (:1765)
            if kname == "calForceDistrGPU":
                if state["iter"] >= 0 and state["iter"] in snaps:
                    snapshot(state["iter"])
                state["iter"] += 1
            if kname == "transformPDFGPU":        # pre-loop transform (:1770): initial f-bar
                out["pre_fbar"] = cp(args[3])
        else:
            # the SC loop ends with calPhysicalVelocity (:1626)
            if kname == "calPhysicalVelocity":
                state["iter"] = state["iter"] + 1 if state["iter"] >= 0 else 1
                if state["iter"] in snaps:
                    snapshot(state["iter"])

    def pre(kname, args):
        if "init_f" in out:
            return
        if kname == "calFluidPotentialGPUEql" and efs:       # first kernel of the EFS run
            out["init_rho"] = cp(args[3])
        if kname in ("calExplicit4thOrderScheme",) and efs and "init_f" not in out:
            pass
        if kname == "calEquilibriumVEFGPU" or kname == "transformEquilibriumVelocity":
            out["init_f"] = cp(args[9] if kname == "calEquilibriumVEFGPU" else args[8])
        if kname == "constantVelocityZouHeBoundaryHigher" and not efs:
            out["init_f"] = cp(args[8]); out["init_rho"] = cp(args[7])
        if kname == "calVelocityBoundaryHigherChangGPU" and not efs:          # the same place with BoundaryMethod = 'Chang'
            out["init_f"] = cp(args[11]); out["init_rho"] = cp(args[7])
    cuda.POST_LAUNCH_HOOK = post
    cuda.PRE_LAUNCH_HOOK = pre
    t0 = time.time()
    sim.runTypeSCmodel()
    cuda.POST_LAUNCH_HOOK = None
    cuda.PRE_LAUNCH_HOOK = None
    refenv.say("%s: %.1f s, N=%d, snapshots %s" % (name, time.time() - t0, sim.fluidNodes.size,
                                                   sorted(k for k in out if k.endswith("_rho"))))
    out.update(isDomain=np.array(sim.isDomain, dtype=np.uint8), fluidNodes=sim.fluidNodes,
               neighboringNodes=sim.neighboringNodes, snaps=np.array(snaps, dtype=np.int64))
    if par["relax"] == "MRT":
        out["collisionMatrix"] = np.array(sim.collisionMatrix)
        out["M"] = np.array(sim.transformationMatrix)
    if img is not None:
        out["image"] = img
    for key, val in refenv.H5_CAPTURE.items():
        out["h5|" + key] = val
    for key, val in par.items():
        out["par_" + key] = np.array(val)
    np.savez_compressed(os.path.join(OUT, "sc_%s.npz" % name), **out)


if __name__ == "__main__":
    names = sys.argv[1:] or list(SCENARIOS)
    if len(names) == 1:
        run(names[0])
    else:
        import subprocess
        procs = [subprocess.Popen([sys.executable, __file__, n]) for n in names]
        sys.exit(max(p.wait() for p in procs))
