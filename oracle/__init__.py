"""CPU parity oracle for the openLBMPM hot path -- TEST INFRASTRUCTURE ONLY.

Plain-C restatements (rk_oracle.c, sc_oracle.c) of the reference's algorithms in the
reference's own sparse AoS data layout, plus thin ctypes wrappers.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package `openlbmpm_amd` never does (and fails loudly without its HIP library).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liblbmpm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rk_oracle.c", "rk_pert_oracle.c", "sc_oracle.c", "rk3d_oracle.c", "rk3d_csf_oracle.c", "tr_oracle.c", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return so


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores inside a quota-limited container, and an OpenMP
    team larger than the quota spins itself to a crawl)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:    # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                quota = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_model():
    """CPU model string of this box (for the bench line)"""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def lib(threads=None):
    """Load the oracle.  OpenMP team: LBMPM_ORACLE_THREADS if set, else every usable core (affinity mask and
    cgroup quota, usable_cores()).  `threads` overrides for this and later calls (the checker of the small
    test problems asks for a small team: waking 100+ threads per loop costs more than the loops)."""
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        env = os.environ.get("LBMPM_ORACLE_THREADS")
        _LIB.rk_oracle_set_threads(ctypes.c_int(int(env) if env else usable_cores()))
    if threads is not None:
        _LIB.rk_oracle_set_threads(ctypes.c_int(int(threads)))
    return _LIB
