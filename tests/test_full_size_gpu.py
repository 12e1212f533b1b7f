"""Size-independent properties at the BASELINE sizes of configs 3, 4 and 5 (config 2 lives in
tests/test_rk2d_gpu.py::test_full_size_properties): what cannot be compared with the CPU oracle in
seconds is held to conservation bounds, finiteness and schedule equivalence on exactly the
workloads bench.py times (same builders, same seed)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


def test_c3_efs_2048_mass_per_component():
    """explicit-forcing Shan-Chen, 2048 x 2048 pore image: each component's mass changes only through
    the open rows (velocity inlet 5.03e-4 on top, pressure outlet at the bottom)"""
    s, m0, _ = bench.build_c3(2048, 2048, 0)
    s.enable_diagnostics(True)                       # end-of-iteration densities of the EFS loop (ShanChenD2Q9.py:2022-2087)
    s.step(1)
    r0a, r1a = s.get("rho0"), s.get("rho1")
    steps = 60
    s.step(steps)
    r0, r1 = s.get("rho0"), s.get("rho1")
    assert np.isfinite(r0).all() and np.isfinite(r1).all() and r0.min() >= 0.0 and r1.min() >= 0.0
    bound = 8.0 * 5.03e-4 * 2048 * steps          # |v_in| * nx * steps, same again for the outlet, x4 slack
    assert abs(r0.sum() - r0a.sum()) < bound and abs(r1.sum() - r1a.sum()) < bound
    assert abs((r0 + r1).sum() - m0) / m0 < 2e-3
    s.close()


def test_c4_tracer_2048_mass_through_inlet_only():
    """colour gradient + D2Q5 tracer, 2048 x 2048: tracer mass changes only through the inlet row, the
    flow's mass only by the inlet flux; no NaN; undershoot of the anti-diffusive interface term ~1 %"""
    s, m0, mass = bench.build_c4(2048, 2048, 0)
    c0 = s.get_tracer(0)
    steps = 60
    s.step(steps)
    c = s.get_tracer(0)
    assert np.isfinite(c).all()
    # no upper bound: the anti-diffusive interface term (beta = 1) piles tracer up against the moving
    # interface (the CPU oracle shows the same: 3.4 after 20 steps on a 256^2 sample); undershoot stays ~1 %
    assert c.min() > -0.05
    assert abs(c.sum() - c0.sum()) < 2.0 * 2048 * steps * 1.0       # <= one inlet row of concentration 1 per step (generous)
    assert abs(mass() - m0) / m0 < 4.0 * 1.0e-4 * 2048 * steps / m0 + 1e-12
    s.close()


def test_c5_512_cubed_storage_layouts_agree_and_mass_bounded(monkeypatch):
    """D3Q19 colour gradient (MRT, the bench configuration) at 512^3 (88 M fluid cells): the 38-value compact storage == dense
    storage bit for bit after 3 steps (phase field, colour densities, velocity); the bench kernel (rk3dq_fused, 23 stored values per
    cell: the same step up to round-off) within 1e-11 of them; the total mass moves only by the inlet flux"""
    from openlbmpm_amd.rk3d import RK3DSlab
    size = (512, 512, 512)
    dom = bench.c5_domain(size)
    rR, rB = bench.c5_densities(dom, 0, size[2])
    m0 = float((rR + rB).sum())
    out = []
    for layout in ("q23", "compact", "dense"):
        if layout == "compact":
            monkeypatch.setenv("LBMPM_RK3D_STORAGE", "38")
        if layout == "dense":
            monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
        s = RK3DSlab(dom, 0, size[2], dict(relax="MRT"))
        assert s.dominant_kernel == {"q23": "rk3dq_fused", "compact": "rk3dc_fused", "dense": "rk3d_fused"}[layout]
        s.set_density(rR, rB)
        s.step_single(3)
        s.phase_field(diagnostics=True)
        out.append({f: s.get(f) for f in ("phi", "rhoR", "rhoB", "vz")})
        s.close()
    for f in out[1]:
        assert np.array_equal(out[1][f], out[2][f]), f
        scale = max(float(np.max(np.abs(out[2][f]))), 1e-300) if f != "vz" else 1e-4      # vz against the inlet velocity
        assert float(np.max(np.abs(out[0][f] - out[2][f]))) / scale < 1e-11, f
    rho = out[0]["rhoR"] + out[0]["rhoB"]
    assert np.isfinite(rho).all()
    assert abs(float(rho.sum()) - m0) / m0 < 4.0 * 1.0e-4 * 512 * 512 * 3 / m0


def test_c5_lattice_under_the_3d_csf_model_bulk_path_equals_full_path():
    """the bench's 512^3 porous lattice under the 3-D CSF model (bench.py's CSF legs): the bulk path (variant 0) against every cell on the
    full path (variant 1), bit for bit after 12 steps -- phase field, colour densities of the next record, force; total mass moves only
    by the inlet flux; most of the lattice is on the bulk path"""
    from openlbmpm_amd.rk3dcsf import RK3DCSFSolver
    size = (512, 512, 512)
    dom = bench.c5_domain(size)
    dom[0] = dom[1]; dom[-1] = dom[-2]
    rR, rB = bench.c5_densities(dom, 0, size[2])
    m0 = float((rR + rB).sum())
    out = []
    for variant in (0, 1):
        s = RK3DCSFSolver(dom, dict(relax="MRT", tauB=0.8, variant=variant))
        s.set_macro(rR, rB)
        s.step(12)
        out.append({f: s.get(f) for f in ("phi", "rec_rhoR", "rec_rhoB", "Fz")})
        if variant == 0:
            assert s.bulk_cells > 0.8 * s.num_fluid_nodes
            bulk = s.bulk_cells
        else:
            assert s.bulk_cells == 0
        s.close()
    for f in out[0]:
        assert np.array_equal(out[0][f], out[1][f]), f
    # ... and cut into 8 z-slabs (the node's rank count, here contexts of one process on this GPU): the same bits, the bulk path up to the faces
    from openlbmpm_amd.rk3dcsf import RK3DCSFCluster
    c = RK3DCSFCluster(dom, dict(relax="MRT", tauB=0.8), nslabs=8)
    c.set_macro(rR, rB)
    c.step(12)
    for f in out[0]:
        assert np.array_equal(out[0][f], c.get(f)), ("8 slabs", f)
    assert c.bulk_cells > 0.97 * bulk
    c.close()
    rho = out[0]["rec_rhoR"] + out[0]["rec_rhoB"]
    assert np.isfinite(rho).all()
    assert abs(float(rho.sum()) - m0) / m0 < 4.0 * 1.0e-4 * 512 * 512 * 12 / m0
