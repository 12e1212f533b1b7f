"""Image ingest: geometry.image_domain on the pore images stored in the golden fixtures must give the isDomain the REAL
reference drivers built from the same images (crop to the bounding box of the solid pixels, solid first / last column,
all-void buffer rows: RKD2Q9.py:373-443 with the ini's numBufferingLayers / ratioTopToBottom, ShanChenD2Q9.py:513-585
with its fixed 20 + 20 rows)."""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_params
from openlbmpm_amd.geometry import image_domain, simple_geometry

FIXTURES = [f for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))) if "image" in np.load(f).files]


def test_there_are_image_fixtures_of_both_driver_families():
    names = [os.path.basename(f) for f in FIXTURES]
    assert any(n.startswith("rk_") for n in names) and any(n.startswith("sc_") for n in names) and len(names) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-4])
def test_image_domain_equals_the_reference_drivers(path):
    d = np.load(path)
    p = load_params(d)
    if os.path.basename(path).startswith("sc_"):
        dom = image_domain(d["image"], 20, 0.5)
    else:
        dom = image_domain(d["image"], p["nbuf"], p["ratio"])
    assert dom.dtype == np.uint8 and np.array_equal(dom, d["isDomain"])


def test_simple_geometry_equals_the_reference():
    """ShanChen2D/SimpleGeometry.py:11-27 through the fixtures without an image"""
    for f in sorted(glob.glob(os.path.join(GOLDEN, "rk_csf_*capillary.npz"))):
        d = np.load(f)
        ny, nx = d["isDomain"].shape
        assert np.array_equal(simple_geometry(nx, ny), d["isDomain"])
