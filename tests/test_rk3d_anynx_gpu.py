"""The compact 23-value storage of the D3Q19 solver for ANY nx (the reference's own 3-D ini is 32 x 32 x 96,
IniFiles/RKtwophasesetup3D.ini:5-7).  Row segments hold nx / ceil(nx / 64) cells (csrc/rk3d.hip::seg_x0); rk3dq_fused<.., RAGGED>
keeps the x-periodic wrap through the segment records.  Held to the oracle (1e-10), to the dense 38-value layout with both colours in
every cell (1e-11) and, slab by slab, to the single domain bit for bit -- on lattices WITHOUT side walls too, where every row wraps."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

FIELDS = ("rhoR", "rhoB", "phi", "vx", "vy", "vz")


def _dom(nx, ny, nz, walls, seed=None):
    from openlbmpm_amd.geometry import porous_spheres
    return porous_spheres(nx, ny, nz, porosity=0.72, rmin=2.0, rmax=5.0, seed=nx + 7 * ny if seed is None else seed, nbuf=3, walls=walls)


def _two_colours(dom, seed=3):
    w = 0.2 + 0.6 * np.random.default_rng(seed).random(dom.shape)
    return np.where(dom == 1, w, 0.0), np.where(dom == 1, 1.0 - w, 0.0)


SIZES = [(32, 32, 24), (100, 14, 20), (65, 9, 18), (127, 10, 16), (129, 11, 16), (37, 13, 18), (500, 9, 14), (4, 5, 10), (63, 8, 12)]


@pytest.mark.parametrize("walls", [True, False], ids=["walls", "periodic"])
@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("nx,ny,nz", SIZES, ids=["%dx%dx%d" % s for s in SIZES])
def test_any_nx_against_the_oracle(nx, ny, nz, relax, walls):
    from openlbmpm_amd.rk3d import RK3DCluster
    from openlbmpm_amd.geometry import initial_densities_rk3d
    from oracle.rk3d import RK3DOracle
    dom = _dom(nx, ny, nz, walls) if nx >= 32 else np.ones((nz, ny, nx), dtype=np.uint8)
    par = dict(tauR=0.9, tauB=1.15, relax=relax)
    for colours in ("front", "mixed"):
        rR, rB = initial_densities_rk3d(dom, 3) if colours == "front" else _two_colours(dom)
        c = RK3DCluster(dom, 1, par)
        assert c.slabs[0].dominant_kernel == "rk3dq_fused"
        c.set_density(rR, rB)
        o = RK3DOracle(dom, rR, rB, par)
        for n in (1, 9):
            c.step(n); o.run(n)
            c.observe(); o.macro()
            umax = max(max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz")), 1e-300)
            for f in FIELDS:
                e = rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None)
                assert e < 1e-10, (f, colours, c.slabs[0].steps_done, e)
        c.close()


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("nx", [100, 32, 65])
def test_any_nx_equals_the_dense_38_value_layout_to_roundoff(nx, relax, monkeypatch):
    """both colours in every cell (no single-colour shortcut anywhere), no side walls, 30 steps"""
    from openlbmpm_amd.rk3d import RK3DCluster
    dom = _dom(nx, 24, 36, walls=False)
    rR, rB = _two_colours(dom)
    par = dict(relax=relax, tauR=0.9, tauB=0.75)
    out = []
    for layout in ("q23", "dense"):
        if layout == "dense":
            monkeypatch.setenv("LBMPM_RK3D_LAYOUT", "dense")
        c = RK3DCluster(dom, 1, par)
        assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
        c.set_density(rR, rB)
        c.step(30); c.observe()
        out.append({f: c.get(f) for f in FIELDS})
        c.close()
    umax = max(float(np.max(np.abs(out[1][f]))) for f in ("vx", "vy", "vz"))
    for f in FIELDS:
        assert rel_err(out[0][f], out[1][f], scale=umax if f[0] == "v" else None) < 1e-11, f


@pytest.mark.parametrize("env,k", [({}, 2), ({}, 3), ({"LBMPM_RK3D_CHUNK": "7"}, 2)],
                         ids=lambda e: ",".join("%s=%s" % (k[11:], v) for k, v in e.items()) if isinstance(e, dict) else "k%d" % e)
@pytest.mark.parametrize("nx,walls", [(96, False), (100, True), (33, False), (200, False)])
def test_any_nx_slabs_equal_the_single_domain_bitwise(nx, walls, env, k, monkeypatch, knobs):
    from openlbmpm_amd.rk3d import RK3DCluster
    from openlbmpm_amd.geometry import initial_densities_rk3d
    dom = _dom(nx, 19, 38, walls)
    rR, rB = initial_densities_rk3d(dom, 5)
    par = dict(relax="MRT", tauR=0.9, tauB=0.7)
    ref = RK3DCluster(dom, 1, par)
    ref.set_density(rR, rB)
    knobs(env)
    c = RK3DCluster(dom, k, par)
    assert c.slabs[0].dominant_kernel == "rk3dq_fused" and c.slabs[0].one_exchange
    c.set_density(rR, rB)
    for n in (1, 12):
        ref.step(n); ref.observe(); c.step(n); c.observe()
        for f in FIELDS:
            assert np.array_equal(ref.get(f), c.get(f)), (f, n)
    ref.close(); c.close()


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("nx,layout", [(64, "q23"), (100, "q23"), (40, "dense"), (64, "compact38")])
def test_pressure_inlet_against_the_oracle(nx, layout, relax, monkeypatch):
    """BoundaryTypeInlet = 'Dirichlet' (Zou-He pressure per colour on the plane nz-2; rk3dq_fused<.., PIN>): every storage vs the oracle,
    two slabs; the prescribed densities stand on the inlet plane"""
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    for k, v in {"dense": {"LBMPM_RK3D_LAYOUT": "dense"}, "compact38": {"LBMPM_RK3D_STORAGE": "38"}}.get(layout, {}).items():
        monkeypatch.setenv(k, v)
    dom = _dom(nx, 14, 30, walls=True)
    rR, rB = _two_colours(dom)
    par = dict(tauR=0.9, tauB=1.1, relax=relax, inlet="Dirichlet", densityRH=0.3, densityBH=0.75)
    c = RK3DCluster(dom, 2, par)
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    for n in (1, 14):
        c.step(n); o.run(n)
        c.observe(); o.macro()
        umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
        for f in FIELDS:
            assert rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None) < 1e-10, (f, n)
    top = dom[-2] == 1
    assert np.all(c.get("rhoR")[-2][top] == 0.3) and np.allclose(c.get("rhoB")[-2][top], 0.75, rtol=0, atol=1e-15)
    c.close()


@pytest.mark.parametrize("relax", ["SRT", "MRT"])
@pytest.mark.parametrize("nx,layout,k", [(64, "q23", 1), (100, "q23", 3), (37, "q23", 2), (40, "dense", 2), (64, "compact38", 1)])
def test_convective_outlet_against_the_oracle(nx, layout, k, relax, monkeypatch):
    """BoundaryTypeOutlet = 'Convective' (AcceleratedRKGPU2D.py:700-784 as z planes: the planes 2, 1, 0 take plane 3's streamed
    populations).  The 23-value storage computes those planes outside its marching kernel (rk3dq_conv_*), the dense layout through the
    source plane of its pulls; LBMPM_RK3D_STORAGE=38 runs the dense layout here.  Against the oracle from the first step on, k slabs;
    then a restart from the stored state continues bit for bit."""
    from openlbmpm_amd.rk3d import RK3DCluster
    from oracle.rk3d import RK3DOracle
    for kk, v in {"dense": {"LBMPM_RK3D_LAYOUT": "dense"}, "compact38": {"LBMPM_RK3D_STORAGE": "38"}}.get(layout, {}).items():
        monkeypatch.setenv(kk, v)
    from openlbmpm_amd.geometry import porous_spheres
    dom = porous_spheres(nx, 14, 34, porosity=0.72, rmin=2.0, rmax=5.0, seed=nx, nbuf=4, walls=True)
    assert all(np.array_equal(dom[0], dom[z]) for z in (1, 2, 3))
    rR, rB = _two_colours(dom)
    par = dict(tauR=0.9, tauB=1.1, relax=relax, outlet="Convective", velocityZB=-2.0e-3)
    c = RK3DCluster(dom, k, par)
    assert c.slabs[0].dominant_kernel == ("rk3dq_fused" if layout == "q23" else "rk3d_fused")
    c.set_density(rR, rB)
    o = RK3DOracle(dom, rR, rB, par)
    for n in (0, 1, 1, 14):
        c.step(n); o.run(n)
        c.observe(); o.macro()
        umax = max(float(np.max(np.abs(o.field(f)))) for f in ("vx", "vy", "vz"))
        for f in FIELDS:
            assert rel_err(c.get(f), o.field(f), scale=umax if f[0] == "v" else None) < 1e-10, (f, c.slabs[0].steps_done)
    got = {f: c.get(f) for f in FIELDS}
    for f in ("rhoR", "rhoB", "phi"):          # the copied planes
        assert np.array_equal(got[f][0], got[f][3]) and np.array_equal(got[f][1], got[f][3]) and np.array_equal(got[f][2], got[f][3]), f
    st, info = c.get_state()
    c.step(9); want = {f: v for f, v in ((f, None) for f in FIELDS)}
    c.observe(); want = {f: c.get(f) for f in FIELDS}
    c.close()
    b = RK3DCluster(dom, 1, par)
    b.set_state(st, info["steps"], info["post_collision"])
    b.step(9); b.observe()
    for f in FIELDS:
        assert np.array_equal(b.get(f), want[f]), ("restart", f)
    b.close()


def test_convective_outlet_needs_equal_masks_on_the_copied_planes():
    from openlbmpm_amd._lib import LbmpmError, ERR_INVALID
    from openlbmpm_amd.rk3d import RK3DSlab
    dom = np.ones((20, 8, 64), dtype=np.uint8)
    dom[2, 3, 10] = 0
    with pytest.raises(LbmpmError) as e:
        RK3DSlab(dom, 0, 20, dict(outlet="Convective"))
    assert e.value.status == ERR_INVALID and "masks must coincide" in str(e.value)
    with pytest.raises(LbmpmError) as e:          # a cut through the copied planes
        RK3DSlab(np.ones((20, 8, 64), dtype=np.uint8), 0, 4, dict(outlet="Convective"))
    assert e.value.status == ERR_INVALID


def test_a_lattice_wider_than_the_packed_coordinates_is_refused():
    """advisor, round 5: rk3dq_fused packs a thread's lattice coordinates into signed 16-bit fields"""
    from openlbmpm_amd._lib import LbmpmError, ERR_INVALID
    from openlbmpm_amd.rk3d import RK3DSlab
    for shape in ((8, 4, 32768), (8, 32768, 4)):
        with pytest.raises(LbmpmError) as e:
            RK3DSlab(np.ones(shape, dtype=np.uint8), 0, 8)
        assert e.value.status == ERR_INVALID and "32767" in str(e.value)


def test_500_cubed_sample_runs_at_the_rate_of_the_512_case():
    """500^3 porous (eight row segments of 62 / 63 cells) within 10 % of the per-cell rate of 512^3 -- here at 500 x 500 x 96 vs
    512 x 512 x 96 (the march cost per plane is what nx changes; tools/k3bench.py runs the cubes)"""
    import bench
    from openlbmpm_amd.rk3d import RK3DSlab
    rate = {}
    for n in (512, 500):
        dom = bench.c5_domain((n, n, 96))
        rR, rB = bench.c5_densities(dom, 0, 96)
        s = RK3DSlab(dom, 0, 96, dict(relax="MRT"))
        assert s.dominant_kernel == "rk3dq_fused"
        s.set_density(rR, rB)
        s.step_single(20)
        s.sync()
        ms, _ = s.step_timed(40)
        rate[n] = s.num_fluid_nodes * 40 / ms
        s.close()
    assert rate[500] > 0.90 * rate[512], rate
