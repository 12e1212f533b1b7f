#!/usr/bin/env python3
"""Golden vectors for the tracer-transport kernels (RKCG2D/AccelerateTransport2DRK.py).

The coupled driver Transport2DRK.py cannot be imported (IndentationError at :1358, missing
transportsetup.ini, missing kernel at :1293 -- SURVEY.md Appendix B-14), so the kernels it calls in
its D2Q5-MRT path (Transport2DRK.py:1341-1418) are pinned ONE BY ONE: each is executed under the
numba stand-in on seeded random inputs over a small porous domain and its inputs/outputs stored.
The host-side matrices are rebuilt with the reference's own statements (Transport2DRK.py:313-347).

Container-only.  Writes tests/golden/tr_kernels.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

OUT = os.environ.get("LBMPM_GOLDEN_OUT") or os.path.dirname(HERE)      # (redirected by tests/test_golden_provenance.py)


def main():
    refenv.setup()
    import importlib
    import scipy.linalg as slin
    T = importlib.import_module("AccelerateTransport2DRK")
    rng = np.random.default_rng(77)
    nx, ny = 18, 26
    dom = np.ones((ny, nx), dtype=np.int64)
    dom[6:-6, 0] = 0; dom[6:-6, -1] = 0
    yy, xx = np.mgrid[0:ny, 0:nx]
    for cx, cy, r in ((5.0, 9.0, 2.6), (12.0, 15.0, 3.1), (8.5, 19.0, 1.9)):
        dom[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 0
    fluidNodes = np.flatnonzero(dom.reshape(-1) == 1).astype(np.int64)
    N = fluidNodes.size
    newIndex = -np.ones((ny, nx), dtype=np.int64)
    newIndex.reshape(-1)[fluidNodes] = np.arange(N)
    xDim, grid, block = 128, (4, int(np.ceil(N / 128))), (32, 1)
    out = dict(isDomain=dom.astype(np.uint8), fluidNodes=fluidNodes)

    nbr = np.zeros(4 * N, dtype=np.int64)
    T.fillNeighboringNodesTransport[grid, block](N, nx, ny, xDim, fluidNodes, newIndex, nbr)
    out["nbr"] = nbr.copy()

    nT = 2
    # host matrices, statements of Transport2DRK.py:313-347
    diffX = np.array([1. / 6., 0.12]); diffY = np.array([1. / 6., 0.2]); dXY = 0.01; dYX = 0.02
    w = np.array([1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.])
    M = np.ones([5, 5])
    M[1, 0] = 0; M[1, 2] = -1.; M[1, 3:] = 0.
    M[2, :3] = 0.; M[2, 4] = -1.
    M[3, 0] = 4.; M[3, 1:] = -1.
    M[4, 0] = 0.; M[4, 3:] = -1.
    Minv = slin.inv(M)
    S = np.zeros([nT, 5, 5])
    for i in range(nT):
        S[i, 1, 1] = (0.5 + 3. * diffX[i]); S[i, 2, 2] = (0.5 + 3. * diffY[i])
        S[i, 1, 2] = 3. * dXY; S[i, 2, 1] = 3. * dYX
    S[:, 0, 0] = 1.0; S[:, 3, 3] = 1.0; S[:, 4, 4] = 1.0
    A = np.zeros([nT, 5, 5])
    for i in range(nT):
        A[i] = -np.dot(Minv, slin.inv(S[i]))
    out.update(M=M, A=A, w=w, diffX=diffX, diffY=diffY, dXY=np.float64(dXY), dYX=np.float64(dYX))

    unitVX = np.array([0., 1., -1, 0., 0.]); unitVY = np.array([0., 0., 0., 1., -1.])
    unitEX = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); unitEY = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
    g = rng.uniform(0.02, 0.3, size=(nT, N, 5))
    conc = np.zeros((nT, N))
    T.calConcentrationGPU[grid, block](N, nT, xDim, 5, conc, g)
    out.update(conc_in_g=g.copy(), conc_out=conc.copy())

    vx = rng.uniform(-0.05, 0.05, N); vy = rng.uniform(-0.05, 0.05, N)
    g1 = g.copy()
    T.calCollisionTransportLinearEqlMRTGPU[grid, block](N, xDim, nT, unitVX, unitVY, vx, vy, conc, g1, M, A, w)
    out.update(col_vx=vx, col_vy=vy, col_conc=conc.copy(), col_in_g=g.copy(), col_out_g=g1.copy())

    rhoR = rng.uniform(0.0, 1.0, N)
    ind = np.zeros(N)
    T.calValueTransportDomain[grid, block](N, xDim, 0.5, ind, rhoR)
    out.update(ind_rhoR=rhoR, ind_out=ind.copy())

    Gx = rng.uniform(-0.4, 0.4, N); Gy = rng.uniform(-0.4, 0.4, N)
    Gx[::7] = 0.0; Gy[::7] = 0.0
    beta = np.array([1.0, 0.6])
    g2 = g1.copy()
    T.calTransportWithInterfaceD2Q5[grid, block](N, xDim, nT, beta, ind, unitEX, unitEY, Gx, Gy, w, conc, g2)
    out.update(itf_Gx=Gx, itf_Gy=Gy, itf_beta=beta, itf_in_g=g1.copy(), itf_out_g=g2.copy())

    g3 = g2.copy()
    T.calFreeConcBoundary3[grid, block](N, nT, nx, xDim, fluidNodes, nbr, conc, g3)
    out.update(free_in_g=g2.copy(), free_out_g=g3.copy())

    gNew = np.zeros_like(g3)
    g4 = g3.copy()
    T.calStreamingTransportGPU[grid, block](N, xDim, nT, nbr, g4, gNew)
    T.calStreamingTransport2GPU[grid, block](N, nT, xDim, gNew, g4)
    out.update(str_in_g=g3.copy(), str_out_g=g4.copy())

    cb = np.array([1.0, 0.25])
    g5 = g4.copy()
    T.calInamuroConstConcBoundary[grid, block](N, xDim, nT, ny, nx, fluidNodes, nbr, cb, w, g5)
    out.update(ina_cb=cb, ina_in_g=g4.copy(), ina_out_g=g5.copy())
    # reaction between tracers (A + B -> C, Transport2DRK.py:1358-1362 / AccelerateTransport2DRK.py:95-111)
    n3 = 3
    g6 = rng.uniform(0.02, 0.3, size=(n3, N, 5))
    conc3 = rng.uniform(0.0, 1.0, size=(n3, N))
    diffJ = np.array([1. / 3., 0.3, 0.4])
    diffJED = np.zeros([n3, 5])                      # statements of Transport2DRK.py:404-410
    for i in range(5):
        if i == 0:
            diffJED[:, i] = diffJ[:]
        else:
            diffJED[:, i] = (1. - diffJ[:]) / 4.
    rate = np.array([0.05])
    g7 = g6.copy()
    T.calReactionTracersGPU[grid, block](N, n3, xDim, rate, diffJED, conc3, g7)
    out.update(rea_rate=rate, rea_J=diffJED, rea_diffJ=diffJ, rea_conc=conc3, rea_in_g=g6.copy(), rea_out_g=g7.copy())
    np.savez_compressed(os.path.join(OUT, "tr_kernels.npz"), **out)
    refenv.say("tr_kernels: N=%d" % N)


if __name__ == "__main__":
    main()
