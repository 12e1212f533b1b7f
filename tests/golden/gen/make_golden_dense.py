#!/usr/bin/env python3
"""Known-answer vectors for the LEGACY dense kernels of ShanChen2D/AccelerateGPU2D.py: the
explicit-forcing pipeline on q-major dense arrays f[9][ny*nx] with boolean masks, one launch per kernel on a 16 x 24
two-fluid case (solid side columns + a solid disc; periodic in y), chained the way the dead dense driver chains them
(ShanChenD2Q9.py:1191 ff.):

    calInteractionForceEFGPU (:1392) -> calExternalForceSolidEF (:2257) -> calMacroVelocityEFGPU (:2460) x 2
    -> calEffectiveVGPU (:2309) -> calEquilibriumFuncEFGPU (:2354) x 2 -> calForcingTermEFGPU (:2403) x 2
    -> calTransformedDistrFuncGPU (:2444) x 2 -> calCollisionEFGPU (:2487) x 2
    -> calHalfWallBounceBack (:2698) x 2 -> calStreamingStep1 / Step2 (:1336 / :1372) x 2
    and calMacroDensityGPU1D (:54), calMacroVelocityGPU1D (:80) on the result; beside the chain one launch each of
    calExternalForceSolid (:2209) and calEffectiveVGPUMRT (:2332).

Container-only.  Writes tests/golden/dense_kernels.npz.
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
    D = importlib.import_module("AccelerateGPU2D")
    rng = np.random.default_rng(2698)
    nx, ny = 16, 24
    yy, xx = np.mgrid[0:ny, 0:nx]
    solid = np.zeros((ny, nx), dtype=bool)
    solid[:, 0] = True; solid[:, -1] = True
    solid[(xx - 8.0) ** 2 + (yy - 11.0) ** 2 <= 2.6 ** 2] = True
    dom = ~solid
    isDomain = dom.reshape(-1).copy(); isSolid = solid.reshape(-1).copy()
    n = nx * ny
    grid, block = (1, ny), (nx, 1)                    # id1D = by * nx + idX
    out = dict(isDomain=isDomain.reshape(ny, nx).astype(np.uint8), isSolid=isSolid.reshape(ny, nx).astype(np.uint8))
    tau = (1.0, 0.8); G = 0.2; Gs = (-0.14, 0.14); constC = 6.0
    out.update(tau=np.array(tau), G=np.float64(G), Gs=np.array(Gs), constC=np.float64(constC))
    f = [np.where(isDomain[None, :], rng.uniform(0.02, 0.2, (9, n)), 0.0) for _ in range(2)]
    f[1] *= 0.3
    rho = [np.zeros(n), np.zeros(n)]
    scratch = np.zeros((9, n))
    for k in range(2):
        D.calMacroDensityGPU1D[grid, block](nx, ny, rho[k], f[k], scratch, isDomain)
    out.update(f0=f[0].copy(), f1=f[1].copy(), rho0=rho[0].copy(), rho1=rho[1].copy())
    psi = [rho[0].copy(), rho[1].copy()]                      # potential = density (ShanChenD2Q9.py 'EFS')
    F = [np.zeros(n) for _ in range(4)]                       # F0x, F0y, F1x, F1y
    D.calInteractionForceEFGPU[grid, block](nx, ny, constC, G, psi[0], psi[1], F[0], F[1], F[2], F[3], isDomain, isSolid)
    out.update(Ff_0x=F[0].copy(), Ff_0y=F[1].copy(), Ff_1x=F[2].copy(), Ff_1y=F[3].copy())
    D.calExternalForceSolidEF[grid, block](nx, ny, Gs[0], Gs[1], psi[0], psi[1], F[0], F[1], F[2], F[3], isDomain, isSolid)
    out.update(F_0x=F[0].copy(), F_0y=F[1].copy(), F_1x=F[2].copy(), F_1y=F[3].copy())
    # side answer (not part of the chain): the fluid-solid force of the original Shan-Chen weights, calExternalForceSolid (:2209), on a copy
    Fs = [out["Ff_0x"].copy(), out["Ff_0y"].copy(), out["Ff_1x"].copy(), out["Ff_1y"].copy()]
    D.calExternalForceSolid[grid, block](nx, ny, Gs[0], Gs[1], psi[0], psi[1], Fs[0], Fs[1], Fs[2], Fs[3], isSolid)
    out.update(Fs_0x=Fs[0], Fs_0y=Fs[1], Fs_1x=Fs[2], Fs_1y=Fs[3])
    v = [np.zeros(n) for _ in range(4)]                       # v0x, v0y, v1x, v1y
    for k in range(2):
        D.calMacroVelocityEFGPU[grid, block](nx, ny, rho[k], F[2 * k], F[2 * k + 1], f[k], v[2 * k], v[2 * k + 1], isDomain)
    out.update(v_0x=v[0].copy(), v_0y=v[1].copy(), v_1x=v[2].copy(), v_1y=v[3].copy())
    ux, uy = np.zeros(n), np.zeros(n)
    D.calEffectiveVGPU[grid, block](nx, ny, tau[0], tau[1], rho[0], rho[1], v[0], v[1], v[2], v[3], ux, uy, isDomain)
    out.update(ueff_x=ux.copy(), ueff_y=uy.copy())
    # side answer: the MRT flavour of the effective velocity, calEffectiveVGPUMRT (:2332), conserveS = (1.0, 0.9)
    uxm, uym = np.zeros(n), np.zeros(n)
    D.calEffectiveVGPUMRT[grid, block](nx, ny, 1.0, 0.9, rho[0], rho[1], v[0], v[1], v[2], v[3], uxm, uym, isDomain)
    out.update(ueffm_x=uxm, ueffm_y=uym)
    feq = [np.zeros((9, n)), np.zeros((9, n))]; ff = [np.zeros((9, n)), np.zeros((9, n))]
    for k in range(2):
        D.calEquilibriumFuncEFGPU[grid, block](nx, ny, rho[k], ux, uy, feq[k], isDomain)
        D.calForcingTermEFGPU[grid, block](nx, ny, rho[k], F[2 * k], F[2 * k + 1], ux, uy, feq[k], ff[k], isDomain)
    out.update(feq0=feq[0].copy(), feq1=feq[1].copy(), ff0=ff[0].copy(), ff1=ff[1].copy())
    for k in range(2):
        D.calTransformedDistrFuncGPU[grid, block](nx, ny, f[k], ff[k], isDomain)
    out.update(ft0=f[0].copy(), ft1=f[1].copy())
    for k in range(2):
        D.calCollisionEFGPU[grid, block](nx, ny, tau[k], f[k], feq[k], ff[k], isDomain)
    out.update(fc0=f[0].copy(), fc1=f[1].copy())
    for k in range(2):
        D.calHalfWallBounceBack[grid, block](nx, ny, f[k], isDomain, isSolid)
    out.update(fb0=f[0].copy(), fb1=f[1].copy())
    for k in range(2):
        mid = np.zeros((9, n))
        D.calStreamingStep1[grid, block](nx, ny, f[k], mid)
        D.calStreamingStep2[grid, block](nx, ny, f[k], mid)
    out.update(fs0=f[0].copy(), fs1=f[1].copy())
    vx1, vy1 = np.zeros(n), np.zeros(n)
    D.calMacroDensityGPU1D[grid, block](nx, ny, rho[0], f[0], scratch, isDomain)
    D.calMacroVelocityGPU1D[grid, block](nx, ny, vx1, vy1, rho[0], f[0], isDomain)
    out.update(rho0_after=rho[0].copy(), vx_after=vx1.copy(), vy_after=vy1.copy())
    np.savez_compressed(os.path.join(OUT, "dense_kernels.npz"), **out)
    refenv.say("dense_kernels: %d fluid of %d nodes" % (int(isDomain.sum()), n))


if __name__ == "__main__":
    main()
