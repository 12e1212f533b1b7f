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
    srcs = [os.path.join(_HERE, f) for f in ("rk_oracle.c", "sc_oracle.c", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB
