#!/usr/bin/env python3
"""Golden vectors for RKCG2D/RKGPU2DBoundary.py: every one of its sixteen kernels, executed (real kernel bodies,
numba stand-in) on seeded random inputs.  The module has no imports at all (`cuda` is an undefined name in it), so
refenv puts `cuda` into builtins before importing it.

The domain has a solid COLUMN x = 0: row 0 then holds nx - 1 fluid nodes, so the compact index range [nx, 2 nx) is not
grid row 1 -- which separates this module's grid-row tests (calConstPressureLowerGPU :414, ghostPointsConstPressureLowerRK
:452) from the compact-index tests of the same-named kernels in AcceleratedRKGPU2D.py (A:1008, A:1045).  Rows 0-3 and
ny-3..ny-1 are otherwise fluid (the kernels index their N / S neighbours without a solid test).

Container-only.  Writes tests/golden/rkb_kernels.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

OUT = os.environ.get("LBMPM_GOLDEN_OUT") or os.path.dirname(HERE)


def main():
    refenv.setup()
    import importlib
    A = importlib.import_module("AcceleratedRKGPU2D")
    B = importlib.import_module("RKGPU2DBoundary")
    rng = np.random.default_rng(414)
    nx, ny = 14, 17
    dom = np.ones((ny, nx), dtype=np.int64)
    dom[:, 0] = 0
    yy, xx = np.mgrid[0:ny, 0:nx]
    dom[(xx - 7.0) ** 2 + (yy - 8.0) ** 2 <= 2.2 ** 2] = 0
    fluidNodes = np.flatnonzero(dom.reshape(-1) == 1).astype(np.int64)
    N = fluidNodes.size
    newIndex = -np.ones((ny, nx), dtype=np.int64)
    newIndex.reshape(-1)[fluidNodes] = np.arange(N)
    xDim, grid, block = 128, (4, int(np.ceil(N / 128))), (32, 1)
    nbr = np.zeros(8 * N, dtype=np.int64)
    A.fillNeighboringNodes[grid, block](N, nx, ny, xDim, fluidNodes, newIndex, nbr)
    out = dict(isDomain=dom.astype(np.uint8), fluidNodes=fluidNodes, nbr=nbr.copy())
    fR0 = rng.uniform(0.01, 0.2, (N, 9)); fB0 = rng.uniform(0.01, 0.2, (N, 9))
    rR0 = fR0.sum(axis=1); rB0 = fB0.sum(axis=1)
    fT0 = fR0 + fB0
    vy0 = rng.uniform(-0.05, 0.05, N)
    oldR = rng.uniform(0.01, 0.2, (N, 9)); oldB = rng.uniform(0.01, 0.2, (N, 9))
    out.update(fR=fR0, fB=fB0, rhoR=rR0, rhoB=rB0, fT=fT0, vy=vy0, fROld=oldR, fBOld=oldB)
    c = lambda a: np.array(a, copy=True)

    def run(name, call, arrays):
        """arrays: name -> array; stored as <name>__<array> after the launch"""
        call()
        for k, v in arrays.items():
            out["%s__%s" % (name, k)] = c(v)

    vyR, vyB, pHB, pHR, pLB, pLR, pL, vIn = -1.0e-3, -4.0e-4, 0.6, 0.5, 0.98, 0.03, 1.01, -2.0e-3
    out.update(vyR=np.float64(vyR), vyB=np.float64(vyB), pHB=np.float64(pHB), pHR=np.float64(pHR), pLB=np.float64(pLB), pLR=np.float64(pLR),
               pL=np.float64(pL), vIn=np.float64(vIn))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("constantVelocityZHBoundaryHigherRK", lambda: B.constantVelocityZHBoundaryHigherRK[grid, block](N, nx, ny, xDim, vyR, vyB, fluidNodes, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    run("ghostPointsConstantVelocityRK", lambda: B.ghostPointsConstantVelocityRK[grid, block](N, nx, ny, xDim, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    for k in ("convectiveOutletGPU", "convectiveOutletGhost2GPU", "convectiveOutletGhost3GPU"):      # chained like the driver (RKD2Q9.py:1069-1081)
        run(k, lambda k=k: getattr(B, k)[grid, block](N, nx, xDim, fluidNodes, nbr, fR, fB, rR, rB), dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    fR, fB, vn = c(fR0), c(fB0), c(vy0)
    for k in ("convectiveAverageBoundaryGPU", "convectiveAverageBoundaryGPU2", "convectiveAverageBoundaryGPU3"):
        run(k, lambda k=k: getattr(B, k)[grid, block](N, nx, xDim, fluidNodes, nbr, vn, fR, fB, oldR, oldB), dict(fR=fR, fB=fB, vn=vn))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("calConstPressureInletGPU", lambda: B.calConstPressureInletGPU[grid, block](N, nx, ny, xDim, pHB, pHR, fluidNodes, rB, rR, fB, fR),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    run("ghostPointsConstPressureInletRK", lambda: B.ghostPointsConstPressureInletRK[grid, block](N, nx, ny, xDim, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("calConstPressureLowerGPU", lambda: B.calConstPressureLowerGPU[grid, block](N, nx, xDim, pLB, pLR, fluidNodes, rB, rR, fB, fR),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    run("ghostPointsConstPressureLowerRK", lambda: B.ghostPointsConstPressureLowerRK[grid, block](N, nx, xDim, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    # the same two through the namesakes in AcceleratedRKGPU2D.py: compact-index tests, a different answer on this domain
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("A_calConstPressureLowerGPU", lambda: A.calConstPressureLowerGPU[grid, block](N, nx, xDim, pLB, pLR, fluidNodes, rB, rR, fB, fR),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    run("A_ghostPointsConstPressureLowerRK", lambda: A.ghostPointsConstPressureLowerRK[grid, block](N, nx, xDim, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    assert not np.array_equal(out["A_calConstPressureLowerGPU__fR"], out["calConstPressureLowerGPU__fR"])
    fR, fB = c(fR0), c(fB0)
    run("calConstPressureHighGPU", lambda: B.calConstPressureHighGPU[grid, block](N, nx, ny, xDim, pHB, pHR, fluidNodes, fB, fR), dict(fR=fR, fB=fB))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("constantVelocityZHBoundaryHigherNewRK",
        lambda: B.constantVelocityZHBoundaryHigherNewRK[grid, block](N, nx, ny, xDim, vyR, vyB, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    fR, fB, rR, rB = c(fR0), c(fB0), c(rR0), c(rB0)
    run("A_constantVelocityZHBoundaryHigherNewRK",
        lambda: A.constantVelocityZHBoundaryHigherNewRK[grid, block](N, nx, ny, xDim, vyR, vyB, fluidNodes, nbr, rR, rB, fR, fB),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB))
    fR, fB, fT, vy = c(fR0), c(fB0), c(fT0), c(vy0)
    run("calConstPressureLowerGPUTotal", lambda: B.calConstPressureLowerGPUTotal[grid, block](N, nx, xDim, pL, fluidNodes, fT, vy, rR0, rB0, fR, fB),
        dict(fR=fR, fB=fB, fT=fT, vy=vy))
    fR, fB, rR, rB, fT, vy = c(fR0), c(fB0), c(rR0), c(rB0), c(fT0), c(vy0)
    run("constantTotalVelocityInlet", lambda: B.constantTotalVelocityInlet[grid, block](N, nx, ny, xDim, vIn, fluidNodes, nbr, rR, rB, fR, fB, fT, vy),
        dict(fR=fR, fB=fB, rhoR=rR, rhoB=rB, fT=fT, vy=vy))
    np.savez_compressed(os.path.join(OUT, "rkb_kernels.npz"), **out)
    refenv.say("rkb_kernels: N=%d, %d arrays" % (N, len(out)))


if __name__ == "__main__":
    main()
