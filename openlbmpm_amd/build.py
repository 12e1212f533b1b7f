"""Build recipe for liblbmpm_hip.so (hipcc, gfx950 only, in-tree so it travels with gpurun).

    python -m openlbmpm_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "liblbmpm_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: keep the reference's evaluation order (no FMA contraction); the
# kernels are HBM-bound, so this costs nothing measurable and tightens parity.
# -mllvm -disable-machine-licm: the marching kernels re-derive per-thread values inside their loops on purpose (DESIGN.md section 4:
# hoisted, they cost registers the kernels do not have); with machine LICM off hipcc stops hoisting address arithmetic as well --
# rk3dq_fused 253 -> 246 VGPRs, c5 MRT + 4 %, SRT + 1.4 % (A/B on one box, three runs each), the 2-D kernels unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=" + os.environ.get("LBMPM_FP_CONTRACT", "off"),
         "-mllvm", "-disable-machine-licm",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_and_link(out, extra, objdir, verbose):
    """one object per source (compiled side by side, recompiled only when the source, a header or this recipe is newer), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h")) + [os.path.abspath(__file__)]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    compile_flags = [f for f in FLAGS if f != "-shared"] + extra
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append([HIPCC] + compile_flags + ["-c", src, "-o", obj])
    if verbose:
        for j in jobs:
            print("[openlbmpm_amd.build]", " ".join(j), flush=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print("[openlbmpm_amd.build]", " ".join(link), flush=True)
    subprocess.check_call(link)
    return out


def build_dev(out, verbose=True, extra=()):
    """The development build (-DLBMPM_DEV): per-workgroup time stamps of rk3dq_fused (LBMPM_RK3D_TRACE) and the timing knock-outs of the
    slab step (LBMPM_RK3D_DBG, LBMPM_RK3D_COMM_CUS).  Never the product: tools/dev/devlib.py builds it beside the tools and points
    LBMPM_LIBRARY at it; `build()` below does not define LBMPM_DEV, and tests/test_codeobj.py checks that the product library holds
    none of those switches."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    return _compile_and_link(out, ["-DLBMPM_DEV"] + list(extra), os.path.join(os.path.dirname(out), "obj" + "".join(e.replace("-D", "_") for e in extra)), verbose)


DEV_LIB = os.path.join(os.path.dirname(PKG), "tools", "dev", "_build", "liblbmpm_hip_dev.so")


def build_dev_if_stale(verbose=True):
    """tools/dev/_build/liblbmpm_hip_dev.so: the product's sources with the tuning knobs (LBMPM_RK3D_TILE | CHUNK | FILL | BOUNDARY | XCC |
    SLAB_SCHEDULE, LBMPM_RK2D_SHAPE, LBMPM_IPC_LAND) and the instrumentation compiled in.  Tests that vary a knob load it (tests/conftest.py
    `knobs`); nothing of the package does."""
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(PKG), "include", "*.h")) + [os.path.abspath(__file__)]
    if not os.path.exists(DEV_LIB) or any(os.path.getmtime(d) > os.path.getmtime(DEV_LIB) for d in deps):
        build_dev(DEV_LIB, verbose)
    return DEV_LIB


RELAXED_LIB = os.path.join(os.path.dirname(PKG), "tools", "dev", "_build", "liblbmpm_hip_relaxed.so")


def build_relaxed(verbose=True):
    """A MEASUREMENT build, never shipped or loaded by the package (tools/relaxed_parity.py): FMA contraction in every file and the
    algebraic form of the wetting rule's cos(acos) / sin(acos) -- the departures from the reference's rounding that the suite's 1e-9
    does not allow and the north star's 1e-6 might (review of round 5, item 4)."""
    flags = [f for f in FLAGS if f != "-shared" and not f.startswith("-ffp-contract=")] + ["-ffp-contract=fast", "-DLBMPM_RELAXED"]
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(os.path.dirname(RELAXED_LIB), "obj_relaxed")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", RELAXED_LIB])
    return RELAXED_LIB


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(PKG, "_obj")
    if force:
        for o in glob.glob(os.path.join(objdir, "*.o")):
            os.remove(o)
    return _compile_and_link(LIB, [], objdir, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
