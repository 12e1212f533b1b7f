"""CPU-only checks of the drop-in boundary: liblbmpm_hip.so loads without a GPU and exports every
symbol that include/lbmpm.h declares; the ctypes signatures in openlbmpm_amd/_lib.py cover them;
config structs have the C layout; and the product fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbmpm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbmpm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from openlbmpm_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(L, n), "liblbmpm_hip.so does not export %s" % n
        assert n in _lib._SIGNATURES, "no ctypes signature for %s" % n
    assert sorted(_lib._SIGNATURES) == names
    assert L.lbmpm_version().startswith(b"liblbmpm_hip")


def test_kernel_level_symbols_exported():
    from openlbmpm_amd import _lib
    from openlbmpm_amd._kernel_specs import KERNELS
    L = _lib.lib()
    text = open(os.path.join(ROOT, "include", "lbmpm_kernels.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(lbmpm_[a-zA-Z0-9_]+)\s*\(", text)))
    assert len(names) >= 65
    for n in names:
        assert hasattr(L, n), "liblbmpm_hip.so does not export %s" % n
    assert {v[0] for v in KERNELS.values()} <= set(names)


def test_config_struct_layout_matches_c():
    """Compile a tiny C program against include/lbmpm.h and compare sizeof/offsetof."""
    from openlbmpm_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "lbmpm.h"
int main(void) {
  printf("%zu %zu %zu\n", sizeof(lbmpm_rk2d_config), offsetof(lbmpm_rk2d_config, beta), offsetof(lbmpm_rk2d_config, device));
  printf("%zu %zu %zu\n", sizeof(lbmpm_sc2d_config), offsetof(lbmpm_sc2d_config, g_solid), offsetof(lbmpm_sc2d_config, inlet_velocity_y));
  printf("%zu %zu %zu %zu\n", sizeof(lbmpm_rk3d_config), offsetof(lbmpm_rk3d_config, solid_phi), offsetof(lbmpm_rk3d_config, device), offsetof(lbmpm_rk3d_config, recolor_diag));
  printf("%zu %zu %zu\n", sizeof(lbmpm_rk2d_perturbation), offsetof(lbmpm_rk2d_perturbation, solid_phi), offsetof(lbmpm_rk2d_perturbation, outlet_rho_b));
  return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    got = list(map(int, out))
    want = [C.sizeof(_lib.RK2DConfig), _lib.RK2DConfig.beta.offset, _lib.RK2DConfig.device.offset,
            C.sizeof(_lib.SC2DConfig), _lib.SC2DConfig.g_solid.offset, _lib.SC2DConfig.inlet_velocity_y.offset,
            C.sizeof(_lib.RK3DConfig), _lib.RK3DConfig.solid_phi.offset, _lib.RK3DConfig.device.offset, _lib.RK3DConfig.recolor_diag.offset,
            C.sizeof(_lib.RK2DPerturbation), _lib.RK2DPerturbation.solid_phi.offset, _lib.RK2DPerturbation.outlet_rho_b.offset]
    assert got == want


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the solvers must raise (status LBMPM_ERR_HIP), never compute."""
    from openlbmpm_amd import _lib
    if _lib.lib().lbmpm_device_count() > 0:
        pytest.skip("a GPU is present")
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.sc2d import SC2DSolver
    dom = np.ones((16, 16), dtype=np.uint8)
    with pytest.raises(_lib.LbmpmError):
        RK2DSolver(dom)
    with pytest.raises(_lib.LbmpmError):
        SC2DSolver(dom)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openlbmpm_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liblbmpm_oracle" not in text, f


def test_wrapper_argument_checks():
    from openlbmpm_amd.rk2d import RK2DSolver
    from openlbmpm_amd.sc2d import SC2DSolver
    with pytest.raises(TypeError):
        RK2DSolver(np.ones(16, dtype=np.uint8))
    with pytest.raises(KeyError):
        RK2DSolver(np.ones((16, 16), dtype=np.uint8), dict(nosuch=1))
    with pytest.raises(ValueError):
        RK2DSolver(np.ones((16, 16), dtype=np.uint8), dict(relax="TRT"))
    with pytest.raises(ValueError):
        SC2DSolver(np.ones((16, 16), dtype=np.uint8), dict(inter="Other"))
