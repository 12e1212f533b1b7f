"""BASELINE configs[0]: original Shan-Chen D2Q9, 128 x 128 static droplet in a fully periodic box
(the case of the reference's CPU path SimpleD2Q9; shanchen2D.ini: G = 3.8, densities 1.0 / 0.06,
tau = 1).  The GPU solver runs it with the boundary kernels switched off (outlet='Periodic'):
  * parity: same kernels as runOptimizedLBM minus the boundary kernels, against the CPU oracle;
  * conservation of both components and persistence of the two phases."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

PAR = dict(inter="ShanChen", relax="SRT", tau0=1.0, tau1=1.0, G=3.8, Gs0=-0.40, Gs1=0.40, outlet="Periodic", vy0=0.0, vy1=0.0)


def droplet(n, r, hi=1.0, lo=0.06):
    yy, xx = np.mgrid[0:n, 0:n]
    inside = (xx - n / 2) ** 2 + (yy - n / 2) ** 2 <= r * r
    return np.where(inside, hi, lo), np.where(inside, lo, hi)


def test_periodic_droplet_vs_oracle():
    from openlbmpm_amd.sc2d import SC2DSolver
    from oracle.sc import SCOracle
    n = 128
    dom = np.ones((n, n), dtype=np.uint8)
    r0, r1 = droplet(n, 20)
    s = SC2DSolver(dom, PAR, diagnostics=True)
    s.set_density(r0, r1)
    o = SCOracle(dom, dict(PAR, rho0=1.0, rho1=1.0, bg0=0.06, bg1=0.06), rho_init=np.stack([r0, r1]))
    for steps in (1, 50, 250):
        s.step(steps); o.run(steps)
        for k in range(2):
            assert rel_err(s.get_compact("rho%d" % k), o.rho[k]) < 1e-9, (steps, k)
            assert rel_err(s.get_compact("f%d" % k), o.f[k]) < 1e-9, (steps, k)
    m = s.get("rho0").sum() + s.get("rho1").sum()
    assert abs(m - (r0 + r1).sum()) / m < 1e-12          # nothing enters or leaves a periodic box
    s.close()


def test_droplet_stays_a_droplet_and_conserves_both_components():
    """Laplace-law measurements are not attempted: with the reference's collision kernel (velocity shift
    tau F / rho_k on each component, O:1274) and its ini parameters the periodic droplet keeps strong
    spurious currents and drifts; the CPU oracle, pinned to that kernel, blows up for most radii after a
    few thousand steps.  What is held here: over the first 600 steps both components are conserved to
    round-off, everything stays finite, and the phases stay separated."""
    from openlbmpm_amd.sc2d import SC2DSolver
    n = 128
    dom = np.ones((n, n), dtype=np.uint8)
    r0, r1 = droplet(n, 22)
    s = SC2DSolver(dom, PAR, diagnostics=True)
    s.set_density(r0, r1)
    s.step(600)
    a, b = s.get("rho0"), s.get("rho1")
    s.close()
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert abs(a.sum() - r0.sum()) / r0.sum() < 1e-11 and abs(b.sum() - r1.sum()) / r1.sum() < 1e-11
    area0, area = np.count_nonzero(r0 > 0.5), np.count_nonzero(a > 0.5)      # the droplet drifts: no fixed probe points
    assert a.max() > 0.9 and b.max() > 0.9 and 0.5 * area0 < area < 1.5 * area0


def test_laplace_law_at_the_nearest_stable_coupling():
    """configs[0]'s Laplace check on the GPU twin, at G = 2.6 (with the ini's 3.8 the pinned kernel holds no static droplet:
    tests/test_oracle_sc.py::test_configs0_shipped_coupling_does_not_hold_a_static_droplet): three radii, 10^4 steps each, the droplet
    stays centred, dp > 0, dp * R within +-5 % of each other, and each equals the CPU oracle's measurement of the same run to 1e-6"""
    from openlbmpm_amd.sc2d import SC2DSolver
    from test_oracle_sc import c1_laplace_numbers, _c1_droplet
    n = 128
    dom = np.ones((n, n), dtype=np.uint8)
    out = []
    for radius in (14, 20, 28):
        r0, r1 = droplet(n, radius)
        s = SC2DSolver(dom, dict(PAR, G=2.6), diagnostics=True)
        s.set_density(r0, r1)
        s.step(10000)
        d = c1_laplace_numbers(s.get("rho0"), s.get("rho1"), 2.6)
        s.close()
        assert abs(d["cx"] - 64.0) < 1e-6 and abs(d["cy"] - 64.0) < 1e-6 and d["dp"] > 0.0, d
        out.append(d)
    s = [d["dpR"] for d in out]
    assert (max(s) - min(s)) / np.mean(s) < 0.10 and 0.15 < np.mean(s) < 0.25, s
    ref = _c1_droplet(2.6, 20, 10000)
    assert abs(out[1]["dpR"] - ref["dpR"]) / ref["dpR"] < 1e-6 and abs(out[1]["R"] - ref["R"]) < 1e-9
