"""Closed forms used by the HIP kernels instead of the reference's dense 9x9 loops, checked
against the matrices the reference builds (RKD2Q9.py:308-340, SimpleD2Q9.py:107-124)."""
import numpy as np
import pytest

from oracle.rk import mrt_matrices
from oracle.sc import transformation_matrix

EX = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1.]); EY = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1.])
W = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)


def test_both_drivers_use_the_same_basis_with_orthogonal_rows():
    M, Minv, _ = mrt_matrices()
    assert np.array_equal(M, transformation_matrix())
    n2 = (M * M).sum(axis=1)
    assert np.array_equal(n2, [9, 36, 36, 6, 12, 6, 12, 4, 4])
    assert np.allclose(M @ M.T, np.diag(n2))
    assert np.max(np.abs(Minv - M.T / n2)) < 1e-15


def test_equilibrium_and_guo_source_moments():
    M, _, _ = mrt_matrices()
    rng = np.random.default_rng(0)
    for _ in range(20):
        rho, ux, uy, Fx, Fy = rng.uniform(0.5, 2), *rng.uniform(-0.1, 0.1, 2), *rng.uniform(-0.01, 0.01, 2)
        eu = EX * ux + EY * uy
        feq = rho * W * (1 + 3 * eu + 4.5 * eu ** 2 - 1.5 * (ux * ux + uy * uy))
        usq = ux * ux + uy * uy
        meq = rho * np.array([1, -2 + 3 * usq, 1 - 3 * usq, ux, -ux, uy, -uy, ux * ux - uy * uy, ux * uy])
        assert np.max(np.abs(M @ feq - meq)) < 1e-14
        src = W * (3 * EX * Fx + 3 * EY * Fy + 9 * (EX * EX - 1 / 3) * ux * Fx + 9 * EX * EY * uy * Fx +
                   9 * EY * EX * ux * Fy + 9 * (EY * EY - 1 / 3) * uy * Fy)
        uF = ux * Fx + uy * Fy
        ms = np.array([0, 6 * uF, -6 * uF, Fx, -Fx, Fy, -Fy, 2 * (ux * Fx - uy * Fy), ux * Fy + uy * Fx])
        assert np.max(np.abs(M @ src - ms)) < 1e-15


def test_recolouring_cosine_identity():
    """cos(phi_i)|e_i| = (e_i . G)/|G| for every D2Q9 direction (used by the HIP recolouring)."""
    g = np.array([0.3, -0.7])
    gn = np.hypot(*g)
    for ex, ey in zip(EX[1:], EY[1:]):
        en = np.hypot(ex, ey)
        assert abs((ex * g[0] + ey * g[1]) / (en * gn) * en - (ex * g[0] + ey * g[1]) / gn) < 1e-15


def test_duplicated_pore_image_is_mirror_tiled():
    """[DuplicateDomain] Option = 'yes' (ShanChenD2Q9.py:513-541)"""
    import numpy as np
    from openlbmpm_amd.geometry import expand_image_domain, image_domain
    a = np.arange(12.0).reshape(3, 4)
    e = expand_image_domain(a, 3, 2)
    assert e.shape == (6, 12)
    assert np.array_equal(e[:3, :4], a) and np.array_equal(e[:3, 4:8], a[:, ::-1]) and np.array_equal(e[:3, 8:], a)
    assert np.array_equal(e[3:, :4], a[::-1]) and np.array_equal(e[3:, 4:8], a[::-1, ::-1])
    img = np.full((10, 8), 255.0); img[0, 0] = 0.0; img[-1, -1] = 0.0; img[4:6, 3:5] = 0.0
    d1, d2 = image_domain(img, 20, 0.5), image_domain(img, 20, 0.5, duplicate=(2, 3))
    assert d1.shape == (50, 8) and d2.shape == (70, 16)
    assert d2[:, 0].sum() == 40 and d2[:, -1].sum() == 40          # side walls, buffer rows stay open


def test_structure_image_file_is_read_as_greyscale(tmp_path):
    """the PNG the reference drivers read from ~/StructureImage (RKD2Q9.py:382, ShanChenD2Q9.py:549):
    0 = solid, anything else = pore, through the same crop / wall / buffer rules as an array"""
    PIL = pytest.importorskip("PIL.Image")
    from openlbmpm_amd.RKD2Q9 import load_structure_image
    from openlbmpm_amd.geometry import image_domain
    rng = np.random.default_rng(1)
    img = (rng.random((40, 30)) > 0.3).astype(np.uint8) * 255
    path = str(tmp_path / "structure.png")
    PIL.fromarray(img).save(path)
    back = load_structure_image(path)
    assert back.shape == img.shape and np.array_equal(back > 0, img > 0)
    assert np.array_equal(image_domain(back, 6, 0.5), image_domain(img.astype(np.float64), 6, 0.5))


def test_voxel_domain_frames_a_sample_like_the_2d_image_rules():
    from openlbmpm_amd.geometry import voxel_domain
    rng = np.random.default_rng(2)
    vox = (rng.random((12, 9, 11)) > 0.4).astype(np.uint8) * 7
    dom = voxel_domain(vox, nbuf=3)
    assert dom.shape == (18, 9, 11) and dom.dtype == np.uint8
    assert dom[:3].all() and dom[-3:].all()                                   # buffer planes: all fluid
    core = dom[3:-3]
    assert not core[:, 0, :].any() and not core[:, -1, :].any() and not core[:, :, 0].any() and not core[:, :, -1].any()
    assert np.array_equal(core[:, 1:-1, 1:-1], (vox != 0)[:, 1:-1, 1:-1])
