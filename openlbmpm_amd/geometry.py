"""Geometry and initial-condition helpers of the host side (own code; behaviour follows the
reference's drivers, cited per function).  Pure numpy, set-up only -- no time-stepping here."""
import numpy as np


def simple_geometry(nx, ny):
    """All void; solid side walls x=0 and x=nx-1 on rows 10..ny-11
    (reference ShanChen2D/SimpleGeometry.py:11-27, used by RKD2Q9.py:429-431)."""
    dom = np.ones((ny, nx), dtype=np.uint8)
    dom[10:-10, 0] = 0
    dom[10:-10, -1] = 0
    return dom


def expand_image_domain(array, x_number, y_number):
    """[DuplicateDomain] Option = 'yes' (ShanChenD2Q9.py:513-541): tile the cropped image x_number x y_number
    times, every second copy mirrored so that pores stay connected across the seams."""
    a = np.asarray(array)
    if x_number < 1 or y_number < 1:
        raise ValueError("duplication numbers must be >= 1")
    rows = []
    for i in range(int(y_number)):
        base = a if i % 2 == 0 else np.flipud(a)
        rows.append(np.hstack([base if j % 2 == 0 else np.fliplr(base) for j in range(int(x_number))]))
    return np.vstack(rows)


def image_domain(image, num_buffering_layers, ratio_top_to_bottom, duplicate=None):
    """Binary pore image (0 = solid) -> isDomain following RKD2Q9.py:373-414 and :432-443:
    crop to the bounding box of solid pixels, [duplicate = (x_number, y_number): mirror-tile it,
    ShanChenD2Q9.py:569-576], force the first/last column solid, then add 2*numBufferingLayers
    all-void rows, int(2*n*ratio) of them before row 0."""
    img = np.asarray(image, dtype=np.float64)
    ys, xs = np.nonzero(img == 0.0)
    if ys.size == 0:
        raise ValueError("image has no solid (zero) pixel")
    eff = np.array(img[ys.min():ys.max() + 1, xs.min():xs.max() + 1], copy=True)
    if duplicate is not None:
        eff = np.array(expand_image_domain(eff, duplicate[0], duplicate[1]), copy=True)
    eff[:, 0] = 0.0
    eff[:, -1] = 0.0
    n_before = int(2 * num_buffering_layers * ratio_top_to_bottom)
    n_after = 2 * num_buffering_layers - n_before
    void = np.full((1, eff.shape[1]), 255.0)
    eff = np.vstack([void] * n_before + [eff] + [void] * n_after)
    return (eff != 0.0).astype(np.uint8)


def porous_disks(nx, ny, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, max_disks=1000000):
    """Synthetic pore image: union of uniformly placed discs added until the
    void fraction drops to `porosity`.  Returns float image, 0 = solid / 255 = void, with one
    solid pixel in two opposite corners so image_domain() keeps the full frame."""
    rng = np.random.default_rng(seed)
    solid = np.zeros((ny, nx), dtype=bool)
    target = (1.0 - porosity) * nx * ny
    count = 0
    n = 0
    while count < target and n < max_disks:
        cx, cy, r = rng.uniform(0, nx), rng.uniform(0, ny), rng.uniform(rmin, rmax)
        x0, x1 = max(int(cx - r), 0), min(int(cx + r) + 2, nx)
        y0, y1 = max(int(cy - r), 0), min(int(cy + r) + 2, ny)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        solid[y0:y1, x0:x1] |= (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        n += 1
        if n % 16 == 0:
            count = int(solid.sum())
    img = np.where(solid, 0.0, 255.0)
    img[0, 0] = 0.0
    img[-1, -1] = 0.0
    return img


def initial_densities_rk(is_domain, image, num_buffering_layers, rho_r=1.0, rho_b=1.0, mode=None):
    """Initial colour distribution.
    mode None  : the reference's live code -- no image: red disc r<=16 at the domain centre
                 (RKD2Q9.py:459-490); image: red below the top buffer rows, blue in them
                 (RKD2Q9.py:511-531).
    mode 'intrusion': red in the top quarter of the domain (rows >= 3*ny/4), blue below --
                 the capillary-intrusion set-up used by bench.py (documented there)."""
    ny, nx = is_domain.shape
    ii, jj = np.mgrid[0:ny, 0:nx]
    if mode == "intrusion":
        red = ii >= (3 * ny) // 4
    elif not image:
        cy, cx = int(ny / 2), int(nx / 2)
        red = np.sqrt((ii - cy) * (ii - cy) + (jj - cx) * (jj - cx)) <= 16.0
    else:
        red = ii < ny - num_buffering_layers
    fluid = is_domain == 1
    rR = np.where(fluid & red, rho_r, 0.0)
    rB = np.where(fluid & ~red, rho_b, 0.0)
    return rR, rB


def porous_spheres(nx, ny, nz, porosity=0.65, rmin=6.0, rmax=20.0, seed=20260928, nbuf=10, walls=True):
    """Synthetic 3-D pore space: union of uniformly placed spheres added until
    the void fraction of the core drops to `porosity`; then the 3-D analogue of the reference's
    image rules (RKD2Q9.py:407-414): solid side walls (x = 0, nx-1 and y = 0, ny-1) and `nbuf`
    all-fluid buffer planes at the bottom and at the top.  Returns isDomain [nz, ny, nx] uint8."""
    rng = np.random.default_rng(seed)
    solid = np.zeros((nz, ny, nx), dtype=bool)
    core = slice(nbuf, nz - nbuf)
    target = (1.0 - porosity) * nx * ny * (nz - 2 * nbuf)
    n = 0
    count = 0
    while count < target and n < 10 ** 7:
        cx, cy, cz, r = rng.uniform(0, nx), rng.uniform(0, ny), rng.uniform(nbuf, nz - nbuf), rng.uniform(rmin, rmax)
        x0, x1 = max(int(cx - r), 0), min(int(cx + r) + 2, nx)
        y0, y1 = max(int(cy - r), 0), min(int(cy + r) + 2, ny)
        z0, z1 = max(int(cz - r), nbuf), min(int(cz + r) + 2, nz - nbuf)
        if z1 > z0:
            zz, yy, xx = np.ogrid[z0:z1, y0:y1, x0:x1]
            ball = (xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2 <= r * r
            sub = solid[z0:z1, y0:y1, x0:x1]
            count += int(np.count_nonzero(ball & ~sub))
            sub |= ball
        n += 1
    if walls:
        solid[core, :, 0] = True; solid[core, :, -1] = True
        solid[core, 0, :] = True; solid[core, -1, :] = True
    solid[:nbuf] = False
    solid[nz - nbuf:] = False
    return (~solid).astype(np.uint8)


def voxel_domain(voxels, nbuf=10, walls=True):
    """isDomain [nz + 2 nbuf][ny][nx] from a voxel array (non-zero = pore): the image rules of
    RKD2Q9.py:407-414 carried to 3-D as in porous_spheres -- solid side walls around the sample,
    `nbuf` all-fluid buffer planes added below and above it."""
    v = np.asarray(voxels)
    if v.ndim != 3:
        raise TypeError("voxels must be a [nz][ny][nx] array")
    core = (v != 0).astype(np.uint8)
    if walls:
        core[:, :, 0] = core[:, :, -1] = 0
        core[:, 0, :] = core[:, -1, :] = 0
    buf = np.ones((int(nbuf),) + core.shape[1:], dtype=np.uint8)
    return np.ascontiguousarray(np.concatenate([buf, core, buf], axis=0))


def initial_densities_rk3d(is_domain, nbuf, rho_r=1.0, rho_b=1.0):
    """3-D analogue of RKD2Q9.py:511-531: red below the top buffer planes, blue in them."""
    nz = is_domain.shape[0]
    zz = np.arange(nz)[:, None, None]
    red = zz < nz - nbuf
    fluid = is_domain == 1
    return np.where(fluid & red, rho_r, 0.0), np.where(fluid & ~red, rho_b, 0.0)
