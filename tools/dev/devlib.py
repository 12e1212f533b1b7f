"""dev: build the -DLBMPM_DEV flavour of the library (time stamps LBMPM_RK3D_TRACE, knock-outs LBMPM_RK3D_DBG / LBMPM_RK3D_COMM_CUS) into
tools/dev/_build/ and make this process load it instead of the product library.  Import BEFORE anything of openlbmpm_amd:

    import devlib      # noqa: F401   (tools/dev on sys.path)
"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "dev", "_build", "liblbmpm_hip_dev.so")


def ensure():
    from openlbmpm_amd import build
    deps = build.sources() + glob.glob(os.path.join(build.CSRC, "*.h"))
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        build.build_dev(OUT)
    os.environ["LBMPM_LIBRARY"] = OUT
    return OUT


if "openlbmpm_amd._lib" in sys.modules:
    raise ImportError("import devlib before openlbmpm_amd (the library path is read when openlbmpm_amd._lib is imported)")
ensure()
