"""The committed fixtures under tests/golden/ are what the committed generators produce from the REAL
reference (its drivers and kernels run under tests/golden/gen's numba stand-in): one scenario per
generator is regenerated here and compared bit for bit.  Container-only: skipped where
/root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tests", "golden", "gen")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is not on this machine")


@pytest.mark.parametrize("script,args,fixture", [
    ("make_golden_rk.py", ["csf_mrt_convective"], "rk_csf_mrt_convective.npz"),
    ("make_golden_sc.py", ["efs_srt_convective"], "sc_efs_srt_convective.npz"),
    ("make_golden_tr.py", [], "tr_kernels.npz"),
    ("make_golden_rk_pert.py", ["kernels"], "rkpert_kernels.npz"),
    ("make_golden_rk_pert.py", ["srt_porous"], "rkpert_srt_porous.npz"),
    ("make_golden_rkb.py", [], "rkb_kernels.npz"),
    ("make_golden_dense.py", [], "dense_kernels.npz"),
    ("make_golden_tr_coupled.py", ["capillary"], "trc_capillary.npz"),
])
def test_fixture_is_reproduced_from_the_reference(tmp_path, script, args, fixture):
    env = dict(os.environ, LBMPM_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(GEN, script)] + args, env=env, cwd=str(tmp_path),
                          stdout=subprocess.DEVNULL, timeout=600)
    new, old = np.load(tmp_path / fixture), np.load(os.path.join(ROOT, "tests", "golden", fixture))
    assert set(old.files) <= set(new.files)            # (later generator versions may record more parameters)
    for k in old.files:
        assert np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == "f"), k
