"""A/B of two builds of the 3-D CSF kernels on one box: python tools/dev/ab_csf.py <csf3d_bench arguments>, LBMPM_LIBRARY naming the build
(an older build lacks the newer entry points: they are dropped from the binding table for the run)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openlbmpm_amd import _lib

if "old" in os.environ.get("LBMPM_LIBRARY", ""):
    for k in [k for k in _lib._SIGNATURES if k.startswith("lbmpm_rk3dcsf_stage") or k.startswith("lbmpm_rk3dcsf_face")]:
        del _lib._SIGNATURES[k]
sys.argv = ["csf3d_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csf3d_bench.py"), run_name="__main__")
