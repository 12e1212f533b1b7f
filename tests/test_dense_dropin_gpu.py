"""`openlbmpm_amd/dropin/AccelerateGPU2D.py`: same-named entry points for the explicit-forcing pipeline of the reference's legacy dense
kernel file ShanChen2D/AccelerateGPU2D.py (:1336-2487, calHalfWallBounceBack :2698), on the file's own dense
direction-major arrays f[9][ny * nx] with boolean masks.  The launch statements below are written the way the dead dense driver writes them
(ShanChenD2Q9.py:1191 ff.: `AccelerateGPU2D.kernel[grid, block](nx, ny, ...)` on numba.cuda device arrays); every stage is held to
tests/golden/dense_kernels.npz -- the real kernel bodies run under the stand-in (tests/golden/gen/make_golden_dense.py) -- at 1e-13,
including the two places where this file differs from the sparse path on purpose (the equilibrium of :2354 with its direction-7 typo, v_y of
:92 not divided by the density)."""
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-13


@pytest.fixture(scope="module")
def mods():
    sys.path.insert(0, os.path.join(ROOT, "openlbmpm_amd", "dropin"))
    import _runtime as cuda          # the numba.cuda-shaped facade (to_device, copy_to_host)
    import AccelerateGPU2D as D
    yield cuda, D
    sys.path.remove(os.path.join(ROOT, "openlbmpm_amd", "dropin"))


def test_the_module_holds_the_pipelines_kernels_under_their_own_names(mods):
    _cuda, D = mods
    for name in ("calMacroDensityGPU1D", "calMacroVelocityGPU1D", "calStreamingStep1", "calStreamingStep2", "calInteractionForceEFGPU",
                 "calExternalForceSolid", "calExternalForceSolidEF", "calEffectiveVGPU", "calEffectiveVGPUMRT", "calEquilibriumFuncEFGPU",
                 "calForcingTermEFGPU", "calTransformedDistrFuncGPU", "calMacroVelocityEFGPU", "calCollisionEFGPU", "calHalfWallBounceBack"):
        assert callable(getattr(D, name)[(1, 1), (1, 1)]), name


def test_dense_pipeline_through_the_dropin_module(mods):
    cuda, D = mods
    d = np.load(os.path.join(GOLDEN, "dense_kernels.npz"))
    ny, nx = d["isDomain"].shape
    n = nx * ny
    grid, block = (1, ny), (nx, 1)
    isDomain = cuda.to_device(d["isDomain"].reshape(-1).astype(np.bool_))
    isSolid = cuda.to_device(d["isSolid"].reshape(-1).astype(np.bool_))
    tau, G, Gs, constC = d["tau"], float(d["G"]), d["Gs"], float(d["constC"])
    dom = d["isDomain"].reshape(-1) == 1

    def same(dev, key, where=None, tol=TOL):
        got, want = dev.copy_to_host(), d[key]
        if where is not None:
            got, want = got[..., where], want[..., where]
        assert rel_err(got, want) < tol, key

    f = [cuda.to_device(d["f0"]), cuda.to_device(d["f1"])]
    rho = [cuda.to_device(np.zeros(n)), cuda.to_device(np.zeros(n))]
    scratch = cuda.to_device(np.zeros((9, n)))
    for k in range(2):
        D.calMacroDensityGPU1D[grid, block](nx, ny, rho[k], f[k], scratch, isDomain)
        same(rho[k], "rho%d" % k)
    assert np.array_equal(scratch.copy_to_host(), d["f1"])                      # the copy the kernel makes on the way
    psi = rho
    F = [cuda.to_device(np.zeros(n)) for _ in range(4)]
    D.calInteractionForceEFGPU[grid, block](nx, ny, constC, G, psi[0], psi[1], F[0], F[1], F[2], F[3], isDomain, isSolid)
    for a, key in zip(F, ("Ff_0x", "Ff_0y", "Ff_1x", "Ff_1y")):
        same(a, key)
    Fs = [cuda.to_device(d[key]) for key in ("Ff_0x", "Ff_0y", "Ff_1x", "Ff_1y")]
    D.calExternalForceSolid[grid, block](nx, ny, Gs[0], Gs[1], psi[0], psi[1], Fs[0], Fs[1], Fs[2], Fs[3], isSolid)
    for a, key in zip(Fs, ("Fs_0x", "Fs_0y", "Fs_1x", "Fs_1y")):
        same(a, key)
    D.calExternalForceSolidEF[grid, block](nx, ny, Gs[0], Gs[1], psi[0], psi[1], F[0], F[1], F[2], F[3], isDomain, isSolid)
    for a, key in zip(F, ("F_0x", "F_0y", "F_1x", "F_1y")):
        same(a, key)
    v = [cuda.to_device(np.zeros(n)) for _ in range(4)]
    for k in range(2):
        D.calMacroVelocityEFGPU[grid, block](nx, ny, rho[k], F[2 * k], F[2 * k + 1], f[k], v[2 * k], v[2 * k + 1], isDomain)
    for a, key in zip(v, ("v_0x", "v_0y", "v_1x", "v_1y")):
        same(a, key)
    ux, uy = cuda.to_device(np.zeros(n)), cuda.to_device(np.zeros(n))
    D.calEffectiveVGPU[grid, block](nx, ny, tau[0], tau[1], rho[0], rho[1], v[0], v[1], v[2], v[3], ux, uy, isDomain)
    same(ux, "ueff_x"); same(uy, "ueff_y")
    uxm, uym = cuda.to_device(np.zeros(n)), cuda.to_device(np.zeros(n))
    D.calEffectiveVGPUMRT[grid, block](nx, ny, 1.0, 0.9, rho[0], rho[1], v[0], v[1], v[2], v[3], uxm, uym, isDomain)
    same(uxm, "ueffm_x"); same(uym, "ueffm_y")
    feq = [cuda.to_device(np.zeros((9, n))) for _ in range(2)]
    ff = [cuda.to_device(np.zeros((9, n))) for _ in range(2)]
    for k in range(2):
        D.calEquilibriumFuncEFGPU[grid, block](nx, ny, rho[k], ux, uy, feq[k], isDomain)
        D.calForcingTermEFGPU[grid, block](nx, ny, rho[k], F[2 * k], F[2 * k + 1], ux, uy, feq[k], ff[k], isDomain)
        same(feq[k], "feq%d" % k); same(ff[k], "ff%d" % k)
    # the typo of :2393 is in: direction 7 differs from the standard second-order term wherever v_x != v_y
    e7 = feq[0].copy_to_host()[7]
    vx, vy, r = d["ueff_x"], d["ueff_y"], d["rho0"]
    std7 = 1. / 36. * r * (1.5 + 3. * (-vx - vy) + 4.5 * (-vx - vy) * (-vx - vy) - (vx * vx + vy * vy) / (2. / 3.))
    assert np.max(np.abs(e7[dom] - std7[dom])) > 1e-6
    for k in range(2):
        D.calTransformedDistrFuncGPU[grid, block](nx, ny, f[k], ff[k], isDomain)
        same(f[k], "ft%d" % k)
    for k in range(2):
        D.calCollisionEFGPU[grid, block](nx, ny, tau[k], f[k], feq[k], ff[k], isDomain)
        same(f[k], "fc%d" % k)
    for k in range(2):
        D.calHalfWallBounceBack[grid, block](nx, ny, f[k], isDomain, isSolid)
        same(f[k], "fb%d" % k)
    for k in range(2):
        mid = cuda.to_device(np.zeros((9, n)))
        D.calStreamingStep1[grid, block](nx, ny, f[k], mid)
        D.calStreamingStep2[grid, block](nx, ny, f[k], mid)
        same(f[k], "fs%d" % k)
    vx1, vy1 = cuda.to_device(np.zeros(n)), cuda.to_device(np.zeros(n))
    D.calMacroDensityGPU1D[grid, block](nx, ny, rho[0], f[0], scratch, isDomain)
    D.calMacroVelocityGPU1D[grid, block](nx, ny, vx1, vy1, rho[0], f[0], isDomain)
    same(rho[0], "rho0_after")
    same(vx1, "vx_after", where=dom)          # (solid nodes: 0 / 0 in the reference and here; compared on the domain)
    same(vy1, "vy_after")
    # quirk of :92: v_y is the bare momentum, not divided by the density
    assert rel_err(vy1.copy_to_host()[dom], (d["vy_after"] * 1.0)[dom]) < TOL and np.max(np.abs(d["vy_after"][dom] / d["rho0_after"][dom] - d["vy_after"][dom])) > 1e-3


def test_argument_checks_of_the_dense_entry_points(mods):
    cuda, D = mods
    a = cuda.to_device(np.zeros(8))
    m = cuda.to_device(np.zeros(8, dtype=np.bool_))
    with pytest.raises(TypeError):
        D.calStreamingStep2[(1, 1), (1, 1)](2, 4, a)                      # argument count, like Numba's explicit signature
    with pytest.raises(TypeError):
        D.calTransformedDistrFuncGPU[(1, 1), (1, 1)](2, 4, a, a, a)       # float64 array where boolean[:] is declared
    with pytest.raises(TypeError):
        D.calCollisionEFGPU[(1, 1), (1, 1)](2, 4, 1.0, np.zeros(8), a, a, m)   # a host array
