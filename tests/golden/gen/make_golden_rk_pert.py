#!/usr/bin/env python3
"""Golden vectors for the PERTURBATION colour-gradient path of RKCG2D (the path the D3Q19 model extends).

Container-only (needs /root/reference).  Usage:
    python tests/golden/gen/make_golden_rk_pert.py [kernels | <scenario> ...]

Two kinds of fixture:

1. ``rkpert_kernels.npz`` -- the kernels of AcceleratedRKGPU2D.py that only the perturbation loop uses,
   ONE BY ONE on seeded random inputs over a small porous domain (real kernel bodies under the numba
   stand-in): calRKCollision1GPU2DSRTNew (A:1125), calRKCollision1GPU2DMRTNew (A:1272),
   calRKCollision23GPUNew (A:1169), constantVelocityZHBoundaryHigherRK (A:657),
   ghostPointsConstantVelocityRK (A:607), calConstPressureLowerGPU (A:1008),
   ghostPointsConstPressureLowerRK (A:1045), calPhysicalVelocityRKGPU2D (A:125), calPhaseFieldPhi (A:1348),
   calTotalFluidPDF (A:1414), calMacroDensityRKGPU2D (A:103).

2. ``rkpert_<scenario>.npz`` -- the REAL driver RKColorGradientLBM.runRKColorGradient2DPerturbation
   (RKD2Q9.py:978-1223) run to completion.  As shipped it dies at its first time step (SURVEY.md
   Appendix B-7); it runs with these in-memory repairs, none of which touches a kernel body and all of
   which are recorded in the fixture (key ``repairs``):
     R1  RKD2Q9.py:1099 passes 10 arguments to ghostPointsConstantVelocityRK (A:607 takes 12; the two
         missing ones, forceX/forceY, are never read by the kernel)            -> two dummies appended
     R2  RKD2Q9.py:1120 passes 10 arguments to calPhysicalVelocityRKGPU2D (A:125 takes 8: the driver
         adds xDomain and fluidNodes, which the kernel has no parameter for)   -> those two dropped
     R3  RKD2Q9.py:1065 builds fluidPDFTotal = fR + fB right after streaming, i.e. BEFORE the boundary
         kernels and before collision 1, and calRKCollision23GPUNew then recolours from that sum: the
         inlet/outlet rows and (SRT) the BGK relaxation of RKD2Q9.py:1170 would be discarded.  The sum is
         taken where the loop needs it: after calRKCollision1GPU2DSRTNew (which relaxes the two colours in
         place) for SRT, immediately before calRKCollision1GPU2DMRTNew (which relaxes the sum) for MRT.
     R4  RKD2Q9.py:1186 reads self.bodyFX / self.bodyFY, which only exist when the ini says
         isBodyForce = 'yes' (RKD2Q9.py:197-207)                               -> set to 0.0 (MRT only)
   The trailing calRecoloringProcess (RKD2Q9.py:1218) is left in: its collisionTotal1/2 and gradient inputs
   are never written, so it adds zero to both colours (checked here: populations identical before/after).

The initial populations are replaced (input data, first launch) by the pre-image of the rest state under streaming,
see unstream() below.  Each scenario file holds the ini parameters, geometry, compaction tables, the initial compact state and
snapshots (fR, fB, rhoR, rhoB, vx, vy, phi, fTot) at the END of the listed time steps.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402
import make_golden_rk as G  # noqa: E402  (ini template, porous image)

OUT = os.environ.get("LBMPM_GOLDEN_OUT") or os.path.dirname(HERE)

REPAIRS = ("R1 ghostPointsConstantVelocityRK: +2 unused arguments (RKD2Q9.py:1099 vs A:607)",
           "R2 calPhysicalVelocityRKGPU2D: arguments 1 and 3 dropped (RKD2Q9.py:1120 vs A:125)",
           "R3 calTotalFluidPDF deferred: after collision 1 (SRT) / just before collision 1 (MRT) instead of RKD2Q9.py:1065",
           "R4 bodyFX = bodyFY = 0.0 (RKD2Q9.py:1186, MRT only)")

SCENARIOS = {
    # name: (ini overrides, snapshot steps, synthetic image parameters or None)
    "srt_capillary": (dict(nx=20, ny=48, steps=80, relax='SRT', tauR=1.0, tauB=0.8, beta=0.9,
                           rhoRL=0.02, rhoBL=1.0), (1, 2, 10, 40, 80), None),
    "srt_porous": (dict(image='yes', nbuf=4, steps=60, relax='SRT', tauR=0.9, tauB=1.1, beta=1.0,
                        vyR=0.0, vyB=-2.0e-3, rhoRL=1.0, rhoBL=0.02), (1, 2, 30, 60),
                   dict(nx=30, ny=40, seed=11, n_discs=8, rmin=2.0, rmax=4.0)),
    # 64 columns: the width at which the D3Q19 code switches to its compact storage (rk3dc_fused)
    "srt_porous64": (dict(image='yes', nbuf=4, steps=50, relax='SRT', tauR=1.0, tauB=0.75, beta=1.0,
                          vyR=0.0, vyB=-1.0e-3, rhoRL=1.0, rhoBL=0.05), (1, 25, 50),
                     dict(nx=64, ny=30, seed=23, n_discs=14, rmin=2.0, rmax=4.5)),
    "mrt_capillary": (dict(nx=16, ny=44, steps=60, relax='MRT', tauR=1.0, tauB=0.8, beta=0.9,
                           rhoRL=0.02, rhoBL=1.0), (1, 60), None),
}
# The LITERAL order of the loop (no R3): calTotalFluidPDF runs where RKD2Q9.py:1065 has it.  Same ini and input as srt_capillary;
# the fixture's `repairs` lists R1, R2 (and R4) only.  It exists to make R3's effect a number.
SCENARIOS["srt_capillary_literal"] = SCENARIOS["srt_capillary"]
SCENARIOS["mrt_capillary_literal"] = SCENARIOS["mrt_capillary"]
AK = dict(AkR=7.0e-3, AkB=9.0e-3, solidPhi=0.5)


class _Pad:
    def __init__(self, k, n):
        self.k, self.n = k, n

    def __getitem__(self, cfg):
        launch = self.k[cfg]

        def go(*a):
            a = list(a)
            while len(a) < self.n:
                a.append(np.zeros(1))
            return launch(*a)
        return go


class _Drop:
    def __init__(self, k, idx):
        self.k, self.idx = k, idx

    def __getitem__(self, cfg):
        launch = self.k[cfg]
        return lambda *a: launch(*[v for i, v in enumerate(a) if i not in self.idx])


def ini_text(par):
    t = G.INI_TEMPLATE.format(**par)
    t = t.replace("'CSF'", "'Perturbation'")
    t = t.replace("AlphaR = 0.44444444", "AlphaR = 0.0").replace("AlphaB = 0.44444444", "AlphaB = 0.0")
    t = t.replace("AkR = 1.4e-1", "AkR = %r" % AK["AkR"]).replace("AkB = 1.4e-1", "AkB = %r" % AK["AkB"])
    t = t.replace("SolidColorDiff = 0.5", "SolidColorDiff = %r" % AK["solidPhi"])
    return t


def run_loop(name):
    overrides, snaps, image = SCENARIOS[name]
    par = dict(G.DEFAULTS); par.update(overrides)
    cuda = refenv.setup()
    import importlib
    import scipy.ndimage as sciimage
    img = None
    if image is not None:
        img = G.porous_image(**image)
        sciimage.imread = lambda path, flatten=True: np.array(img, copy=True)
    inidir = refenv.write_ini_dir({"RKtwophasesetup2D.ini": ini_text(par)})
    RKD2Q9 = importlib.import_module("RKD2Q9")
    A = importlib.import_module("AcceleratedRKGPU2D")
    mrt = par["relax"] == 'MRT'
    # ---- R1, R2
    A.ghostPointsConstantVelocityRK = _Pad(A.ghostPointsConstantVelocityRK, 12)
    A.calPhysicalVelocityRKGPU2D = _Drop(A.calPhysicalVelocityRKGPU2D, (1, 3))
    literal = name.endswith("_literal")
    # ---- R3: the launch at RKD2Q9.py:1065 is recorded, not executed; it runs at its repaired place
    real_total = A.calTotalFluidPDF
    deferred = {}

    class _Defer:
        def __getitem__(self, cfg):
            def go(*a):
                deferred["cfg"], deferred["args"] = cfg, a
            return go
    if not literal:
        A.calTotalFluidPDF = _Defer()

    class _Around:
        def __init__(self, k, before):
            self.k, self.before = k, before

        def __getitem__(self, cfg):
            launch = self.k[cfg]

            def go(*a):
                if self.before:
                    real_total[deferred["cfg"]](*deferred["args"])
                launch(*a)
                if not self.before:
                    real_total[deferred["cfg"]](*deferred["args"])
            return go
    if literal:
        pass
    elif mrt:
        A.calRKCollision1GPU2DMRTNew = _Around(A.calRKCollision1GPU2DMRTNew, True)
    else:
        A.calRKCollision1GPU2DSRTNew = _Around(A.calRKCollision1GPU2DSRTNew, False)
    sim = RKD2Q9.RKColorGradientLBM(inidir)
    if mrt:
        sim.bodyFX = 0.0; sim.bodyFY = 0.0        # R4

    state = {"step": 0, "last": {}, "noop_checked": 0}
    out = {}

    def unstream(f, nbr):
        """the pre-image of the rest state under the reference's streaming (A:340-417): f''_i(n) = w_i rho(n + e_i),
        or w_i rho(n) where n + e_i is not fluid (that slot only ever bounces back).  The capture starts from it, so
        that the first distribution the loop's boundary / collision kernels see is f_i = w_i rho -- the state a solver
        that collides before it streams (the D3Q19 code) starts from.  Input data, not a code change."""
        rho = f.sum(axis=1)
        w = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
        g = np.empty_like(f)
        g[:, 0] = w[0] * rho
        nb = np.asarray(nbr).reshape(-1, 8)
        for i in range(1, 9):
            q = nb[:, i - 1]
            g[:, i] = w[i] * np.where(q >= 0, rho[np.maximum(q, 0)], rho)
        return g

    def pre(kname, args):
        if kname == "calStreaming1GPU" and "init_fR" not in out:
            out["init_rhoR"] = np.array(args[4]).sum(axis=1)          # compact densities of the rest state
            args[4][...] = unstream(np.array(args[4]), args[3])
            out["init_fR"] = np.array(args[4], copy=True).view(np.ndarray)
        elif kname == "calStreaming1GPU" and "init_fB" not in out:
            out["init_rhoB"] = np.array(args[4]).sum(axis=1)
            args[4][...] = unstream(np.array(args[4]), args[3])
            out["init_fB"] = np.array(args[4], copy=True).view(np.ndarray)
        if kname == "calRecoloringProcess":
            state["before"] = (np.array(args[12], copy=True), np.array(args[13], copy=True))

    def post(kname, args):
        state["last"][kname] = args
        if kname != "calRecoloringProcess":               # last launch of a step (RKD2Q9.py:1218)
            return
        b = state.pop("before")
        assert np.array_equal(b[0], args[12]) and np.array_equal(b[1], args[13]), "calRecoloringProcess is not a no-op here"
        state["noop_checked"] += 1
        state["step"] += 1
        k = state["step"]
        if k in snaps:
            L = state["last"]
            c23 = L["calRKCollision23GPUNew"]
            v = L["calPhysicalVelocityRKGPU2D"]
            rec = dict(fR=c23[18], fB=c23[19], fTot=c23[22], rhoR=c23[13], rhoB=c23[14], phi=c23[15], vx=v[6], vy=v[7])
            for key, val in rec.items():
                out["s%d_%s" % (k, key)] = np.array(val, copy=True).view(np.ndarray)
    cuda.PRE_LAUNCH_HOOK, cuda.POST_LAUNCH_HOOK = pre, post
    t0 = time.time()
    sim.runRKColorGradient2DPerturbation()
    cuda.PRE_LAUNCH_HOOK = cuda.POST_LAUNCH_HOOK = None
    refenv.say("pert_%s: %d steps in %.1f s, N=%d" % (name, state["step"], time.time() - t0, sim.fluidNodes.size))
    assert state["step"] == par["steps"] and state["noop_checked"] == par["steps"]
    out.update(isDomain=np.array(sim.isDomain, dtype=np.uint8), fluidNodes=sim.fluidNodes,
               neighboringNodes=sim.neighboringNodes, snaps=np.array(snaps, dtype=np.int64),
               steps=np.int64(par["steps"]), constantB=np.array(sim.constantBNew), solidPhi=np.float64(sim.solidPhi),
               AkR=np.float64(sim.AkR), AkB=np.float64(sim.AkB),
               repairs=np.array([r for r in REPAIRS if not (literal and r.startswith("R3"))]))
    if mrt:
        out.update(M=sim.transformationM, Minv=sim.invTransformationM, S=np.array(sim.collisionS))
    if img is not None:
        out["image"] = img
    for key, val in par.items():
        out["par_" + key] = np.array(val)
    np.savez_compressed(os.path.join(OUT, "rkpert_%s.npz" % name), **out)


def run_kernels():
    refenv.setup()
    import importlib
    A = importlib.import_module("AcceleratedRKGPU2D")
    rng = np.random.default_rng(20260928)
    nx, ny = 18, 22
    dom = np.ones((ny, nx), dtype=np.int64)
    dom[4:-4, 0] = 0; dom[4:-4, -1] = 0
    yy, xx = np.mgrid[0:ny, 0:nx]
    for cx, cy, r in ((5.0, 8.0, 2.4), (12.0, 13.0, 2.9)):
        dom[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0
    fluidNodes = np.flatnonzero(dom.reshape(-1) == 1).astype(np.int64)
    N = fluidNodes.size
    newIndex = -np.ones((ny, nx), dtype=np.int64)
    newIndex.reshape(-1)[fluidNodes] = np.arange(N)
    xDim, grid, block = 128, (4, int(np.ceil(N / 128))), (32, 1)
    nbr = np.zeros(8 * N, dtype=np.int64)
    A.fillNeighboringNodes[grid, block](N, nx, ny, xDim, fluidNodes, newIndex, nbr)
    out = dict(isDomain=dom.astype(np.uint8), fluidNodes=fluidNodes, nbr=nbr.copy())
    EX = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); EY = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
    W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    CR = np.zeros(9); CB = np.zeros(9)                      # RKD2Q9.py:143-151 with AlphaR = AlphaB = 0
    CR[1:5] = 1. / 5.; CR[5:] = 1. / 20.; CB[:] = CR
    Bnew = np.ones(9); Bnew[0] = -2. / 9.; Bnew[1:5] = 1. / 9.; Bnew[5:] = 1. / 36.      # RKD2Q9.py:131-133
    scheme = np.ones(9)
    fR = rng.uniform(0.01, 0.2, (N, 9)); fB = rng.uniform(0.01, 0.2, (N, 9))
    fB[::5] *= 1e-6                                          # nearly pure red nodes
    rhoR = np.zeros(N); rhoB = np.zeros(N)
    A.calMacroDensityRKGPU2D[grid, block](N, xDim, fR, fB, rhoR, rhoB)
    out.update(fR=fR.copy(), fB=fB.copy(), rhoR=rhoR.copy(), rhoB=rhoB.copy())
    fT = np.zeros((N, 9))
    A.calTotalFluidPDF[grid, block](N, xDim, fR, fB, fT)
    out["fT"] = fT.copy()
    vx = np.zeros(N); vy = np.zeros(N)
    A.calPhysicalVelocityRKGPU2D[grid, block](N, xDim, fR, fB, rhoR, rhoB, vx, vy)
    out.update(vx=vx.copy(), vy=vy.copy())
    phi = np.zeros(N)
    A.calPhaseFieldPhi[grid, block](N, xDim, rhoR, rhoB, phi)
    out["phi"] = phi.copy()
    # collision 1, SRT (A:1125): relaxes both colours in place
    tauR, tauB, delta = 0.9, 1.3, 0.98
    c1R = fR.copy(); c1B = fB.copy(); dummy = np.zeros((N, 9))
    A.calRKCollision1GPU2DSRTNew[grid, block](N, xDim, delta, tauR, tauB, EX, EY, CR, CB, W, vx, vy, rhoR, rhoB, phi,
                                               c1R, c1B, dummy, dummy.copy())
    out.update(tauR=np.float64(tauR), tauB=np.float64(tauB), col1_fR=c1R.copy(), col1_fB=c1B.copy())
    # collision 1, MRT (A:1272) on the sum, with and without a body force
    M = np.zeros([9, 9])                                     # statements of RKD2Q9.py:308-336
    M[0, :] = 1.
    M[1, :] = -1.; M[1, 0] = -4.; M[1, 5:] = 2.
    M[2, :] = 1.; M[2, 0] = 4.; M[2, 1:5] = -2.
    M[3, 1] = 1.; M[3, 3] = -1.; M[3, 5] = 1.; M[3, 6:-1] = -1.; M[3, -1] = 1.
    M[4, 1] = -2.; M[4, 3] = 2.; M[4, 5] = 1.; M[4, 6:-1] = -1.; M[4, -1] = 1.
    M[5, 2] = 1.; M[5, 4] = -1.; M[5, 5:7] = 1.; M[5, 7:] = -1.
    M[6, 2] = -2.; M[6, 4] = 2.; M[6, 5:7] = 1.; M[6, 7:] = -1.
    M[7, 1] = 1.; M[7, 2] = -1.; M[7, 3] = 1.; M[7, 4] = -1.
    M[8, 5] = 1.; M[8, 6] = -1.; M[8, 7] = 1.; M[8, 8] = -1.
    Minv = np.linalg.inv(M)
    S = np.zeros(9); S[1] = 1.64; S[2] = 1.54; S[4] = 1.9; S[6] = 1.9
    m1 = fT.copy()
    A.calRKCollision1GPU2DMRTNew[grid, block](N, xDim, delta, tauR, tauB, 0.0, 0.0, EX, EY, CR, CB, W, vx, vy, rhoR, rhoB,
                                               phi, m1, M, Minv, S.copy())
    m2 = fT.copy()
    A.calRKCollision1GPU2DMRTNew[grid, block](N, xDim, delta, tauR, tauB, 1.0e-5, -2.0e-5, EX, EY, CR, CB, W, vx, vy, rhoR,
                                               rhoB, phi, m2, M, Minv, S.copy())
    out.update(M=M, Minv=Minv, S=S, mrt1_fT=m1.copy(), mrt1_fT_force=m2.copy(), bodyF=np.array([1.0e-5, -2.0e-5]))
    # collisions 2 + 3 (A:1169) from the SRT result
    beta, AkR, AkB, solidPhi = 0.9, 7.0e-3, 9.0e-3, 0.4
    t23 = c1R + c1B
    r23 = c1R.copy(); b23 = c1B.copy(); cgx = np.zeros(nx * ny); cgy = np.zeros(nx * ny)
    phi_z = phi.copy()
    A.calRKCollision23GPUNew[grid, block](N, xDim, beta, AkR, AkB, solidPhi, fluidNodes, nbr, Bnew, W, EX, EY, scheme, rhoR,
                                           rhoB, phi_z, CR, CB, r23, b23, cgx, cgy, t23)
    out.update(beta=np.float64(beta), AkR=np.float64(AkR), AkB=np.float64(AkB), solidPhi=np.float64(solidPhi), constantB=Bnew,
               c23_in_fT=(c1R + c1B), c23_fT=t23.copy(), c23_fR=r23.copy(), c23_fB=b23.copy())
    # a uniform phase field equal to the solid value: the gradient cancels exactly -> zero-gradient branches (A:1225, A:1247)
    rU = np.full(N, 0.7); bU = np.full(N, 0.3)
    tU = fT.copy(); rUo = np.zeros((N, 9)); bUo = np.zeros((N, 9))
    A.calRKCollision23GPUNew[grid, block](N, xDim, beta, AkR, AkB, 0.4, fluidNodes, nbr,
                                           Bnew, W, EX, EY, scheme, rU, bU, phi_z, CR, CB, rUo, bUo, cgx, cgy, tU)
    out.update(c23u_fT=tU.copy(), c23u_fR=rUo.copy(), c23u_fB=bUo.copy())
    # boundary rows: velocity inlet per colour (A:657) + ghost row (A:607)
    zR = fR.copy(); zB = fB.copy(); zrR = rhoR.copy(); zrB = rhoB.copy()
    vyR, vyB = -1.0e-3, -4.0e-4
    A.constantVelocityZHBoundaryHigherRK[grid, block](N, nx, ny, xDim, vyR, vyB, fluidNodes, zrR, zrB, zR, zB)
    out.update(vyR=np.float64(vyR), vyB=np.float64(vyB), zh_fR=zR.copy(), zh_fB=zB.copy(), zh_rhoR=zrR.copy(), zh_rhoB=zrB.copy())
    A.ghostPointsConstantVelocityRK[grid, block](N, nx, ny, xDim, fluidNodes, nbr, zrR, zrB, zR, zB, np.zeros(1), np.zeros(1))
    out.update(zhg_fR=zR.copy(), zhg_fB=zB.copy(), zhg_rhoR=zrR.copy(), zhg_rhoB=zrB.copy())
    # pressure outlet per colour (A:1008) + ghost row (A:1045)
    pR = fR.copy(); pB = fB.copy(); prR = rhoR.copy(); prB = rhoB.copy()
    pLB, pLR = 0.98, 0.03
    A.calConstPressureLowerGPU[grid, block](N, nx, xDim, pLB, pLR, fluidNodes, prB, prR, pB, pR)
    out.update(pLB=np.float64(pLB), pLR=np.float64(pLR), pl_fR=pR.copy(), pl_fB=pB.copy(), pl_rhoR=prR.copy(), pl_rhoB=prB.copy())
    A.ghostPointsConstPressureLowerRK[grid, block](N, nx, xDim, fluidNodes, nbr, prR, prB, pR, pB)
    out.update(plg_fR=pR.copy(), plg_fB=pB.copy(), plg_rhoR=prR.copy(), plg_rhoB=prB.copy())
    np.savez_compressed(os.path.join(OUT, "rkpert_kernels.npz"), **out)
    refenv.say("rk_pert_kernels: N=%d" % N)


if __name__ == "__main__":
    names = sys.argv[1:] or (["kernels"] + list(SCENARIOS))
    if len(names) == 1:
        run_kernels() if names[0] == "kernels" else run_loop(names[0])
    else:
        import subprocess
        procs = [subprocess.Popen([sys.executable, __file__, n]) for n in names]
        sys.exit(max(p.wait() for p in procs))
