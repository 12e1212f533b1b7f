#!/usr/bin/env python3
"""Generate colour-gradient (RKCG2D) golden vectors by running the REAL reference driver.

Container-only (needs /root/reference).  Usage:
    python tests/golden/gen/make_golden_rk.py [scenario ...]
Writes tests/golden/rk_<scenario>.npz.  Each file holds the ini parameters, the geometry,
the reference's compaction tables, the initial state and end-of-step snapshots of every
device array of RKColorGradientLBM.runRKColorGradient2DCSF (RKD2Q9.py:1225-1490).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

OUT = os.environ.get("LBMPM_GOLDEN_OUT") or os.path.dirname(HERE)      # (redirected by tests/test_golden_provenance.py)

INI_TEMPLATE = """[ImageSetup]
Existance = '{image}'

[DomainSize]
xDomain = {nx}
yDomain = {ny}
numBufferingLayers = {nbuf}
ratioTopToBottom = {ratio}

[SurfaceTension]
SurfaceTensionType = 'CSF'
SurfaceTensionValue = {sigma}
ContactAngle = {theta}
WettingType = {wetting}

[RKParameters]
AlphaR = 0.44444444
AlphaB = 0.44444444
BetaThickness = {beta}
AkR = 1.4e-1
AkB = 1.4e-1
DeltaValue = {delta}

[FluidParameters]
TauR = {tauR}
TauB = {tauB}
InitialRhoR = 1.0
InitialRhoB = 1.0
TauType = {tautype}

[BodyForce]
isBodyForce = 'no'
bodyForceX = 0.0
bodyForceY = 0.0

[SolidBoundarySetup]
SolidColorDiff = 0.5

[BoundaryCondition]
BoundaryTypeInlet = '{inlet}'
NeumannType = 'ZouHe'
velocityYR = {vyR}
velocityYB = {vyB}
densityBH = {rhoBH}
densityRH = {rhoRH}
BoundaryTypeOutlet = '{outlet}'
densityBL = {rhoBL}
densityRL = {rhoRL}

[GradientType]
Type = 'Isotropic'

[TimeSetup]
TimeSteps = {steps}
TimeInterval = {interval}

[Parallelism]
Parallel = 'yes'
xDimension = 128
ThreadsNum = 32

[RelaxationType]
Type = '{relax}'

[CyclesSetup]
IsCycle = 'no'
LastStep = 100
"""

DEFAULTS = dict(image='no', nx=20, ny=48, nbuf=6, ratio=0.5, sigma=0.1, theta=60, wetting=2,
                beta=0.7, delta=0.98, tauR=1.0, tauB=1.0, tautype=2, inlet='Neumann',
                vyR=-1.0e-4, vyB=0.0, rhoBH=5e-8, rhoRH=1.00536, outlet='Dirichlet',
                rhoBL=1.0, rhoRL=5e-8, steps=60, interval=1000, relax='MRT')


def porous_image(nx, ny, seed, n_discs, rmin, rmax):
    """Synthetic binary pore image: 0 = solid, 255 = void (what RKD2Q9.py:382-399 expects).
    One solid pixel is kept in two opposite corners so the loader's bounding-box crop
    (RKD2Q9.py:387-399) keeps the full frame."""
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
    # name: (ini overrides, snapshot steps, synthetic image parameters or None)
    "csf_mrt_capillary": (dict(nx=20, ny=48, steps=200), (1, 2, 10, 50, 200), None),
    "csf_mrt_tauratio": (dict(nx=16, ny=44, steps=120, tauR=1.0, tauB=0.7, tautype=2, theta=120),
                         (1, 120), None),
    "csf_mrt_tautype1": (dict(nx=16, ny=44, steps=100, tauR=0.9, tauB=0.65, tautype=1, delta=0.9),
                         (1, 100), None),
    "csf_srt_capillary": (dict(nx=16, ny=44, steps=120, relax='SRT', tauR=1.0, tauB=0.8),
                          (1, 120), None),
    "csf_mrt_wetting1": (dict(nx=16, ny=44, steps=100, wetting=1, theta=45), (1, 100), None),
    "csf_mrt_convective": (dict(nx=16, ny=44, steps=100, outlet='Convective'), (1, 100), None),
    "csf_mrt_pinlet": (dict(nx=16, ny=44, steps=100, inlet='Dirichlet', rhoRH=1.002, rhoBH=5e-8),
                       (1, 100), None),
    "csf_mrt_porous": (dict(image='yes', nbuf=4, steps=150, theta=50),
                       (1, 2, 50, 150), dict(nx=34, ny=44, seed=7, n_discs=10, rmin=2.0, rmax=4.5)),
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
    inidir = refenv.write_ini_dir({"RKtwophasesetup2D.ini": INI_TEMPLATE.format(**par)})
    RKD2Q9 = importlib.import_module("RKD2Q9")
    sim = RKD2Q9.RKColorGradientLBM(inidir)

    state = {"step": 0, "last": {}}
    out = {}
    # argument positions (reference signatures, AcceleratedRKGPU2D.py)
    def post(kname, args):
        state["last"][kname] = args
        if kname == "calMacroDensityRKGPU2D":           # last launch of a step (RKD2Q9.py:1487)
            state["step"] += 1
            k = state["step"]
            if k in snaps:
                L = state["last"]
                fR, fB, rhoR, rhoB = args[2], args[3], args[4], args[5]
                v = L["calPhysicalVelocityRKGPU2DNew1"]
                g = L["calRKInitialGradient"]
                rec = dict(fR=fR, fB=fB, rhoR=rhoR, rhoB=rhoB, vx=v[5], vy=v[6],
                           phi=g[8], phiSolid=g[9], Gx=g[10], Gy=g[11])
                for fk in ("calForceTermInColorGradientNew2D", "calForceTermInColorGradient2D"):
                    if fk in L:
                        rec.update(Fx=L[fk][9], Fy=L[fk][10], K=L[fk][11])
                for key, val in rec.items():
                    out["s%d_%s" % (k, key)] = np.array(val, copy=True).view(np.ndarray)
    cuda.POST_LAUNCH_HOOK = post

    def pre(kname, args):
        # true initial compact state = device arrays at the very first launch of the loop
        if "init_fR" in out:
            return
        if kname == "constantTotalVelocityInlet":        # A:2348 argument order
            out["init_fR"] = np.array(args[9], copy=True).view(np.ndarray)
            out["init_fB"] = np.array(args[10], copy=True).view(np.ndarray)
        elif kname == "calConstPressureInletGPU":        # A:925 argument order
            out["init_fB"] = np.array(args[9], copy=True).view(np.ndarray)
            out["init_fR"] = np.array(args[10], copy=True).view(np.ndarray)
    cuda.PRE_LAUNCH_HOOK = pre

    # capture the initial compact state just before the time loop starts: the first
    # to_device of the run (RKD2Q9.py:1243) happens after all host-side set-up.
    t0 = time.time()
    sim.runRKColorGradient2D()
    cuda.POST_LAUNCH_HOOK = None
    cuda.PRE_LAUNCH_HOOK = None
    refenv.say("%s: %d steps in %.1f s, N=%d" % (name, state["step"], time.time() - t0,
                                                   sim.fluidNodes.size))
    out.update(
        isDomain=np.array(sim.isDomain, dtype=np.uint8),
        fluidNodes=sim.fluidNodes, neighboringNodes=sim.neighboringNodes,
        wettingSolidNodes=sim.wettingSolidNodes,
        neighboringWettingSolidNodes=sim.neighboringWettingSolidNodes,
        M=sim.transformationM if hasattr(sim, "transformationM") else np.zeros((9, 9)),
        Minv=sim.invTransformationM if hasattr(sim, "invTransformationM") else np.zeros((9, 9)),
        snaps=np.array(snaps, dtype=np.int64),
        steps=np.int64(par["steps"]),
    )
    if sim.wettingSolidNodes.size > 0:
        out.update(fluidNodesWithSolidGPU=sim.fluidNodesWithSolidGPU,
                   fluidNodesWithSolidOriginal=sim.fluidNodesWithSolidOriginal,
                   nsX=sim.nsX, nsY=sim.nsY)
    if img is not None:
        out["image"] = img
    # HDF5 record 0 as written by resultInHDF5 (RKD2Q9.py:938-957) at iStep-1 == 0
    for key, val in refenv.H5_CAPTURE.items():
        out["h5|" + key] = val
    for key, val in par.items():
        out["par_" + key] = np.array(val)
    np.savez_compressed(os.path.join(OUT, "rk_%s.npz" % name), **out)


if __name__ == "__main__":
    names = sys.argv[1:] or list(SCENARIOS)
    if len(names) == 1:
        run(names[0])
    else:   # one process per scenario: the reference modules keep module-level state
        import subprocess
        procs = [subprocess.Popen([sys.executable, __file__, n]) for n in names]
        sys.exit(max(p.wait() for p in procs))
