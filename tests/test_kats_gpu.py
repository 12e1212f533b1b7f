"""Known-answer tests of the kernel-level entry points whose reference kernels no working loop launches (earlier
generations of the colour-gradient step, Chang / free-flow boundary rows, the D2Q9 tracer scheme, ...).
tests/golden/kats_<module>.npz holds, per case, every argument of ONE launch of the real kernel body (numba stand-in,
tests/golden/gen/make_golden_kats.py) and every array argument after it; the same-named HIP entry point is launched
on the same arguments -- picked by the reference kernel's parameter names -- and every array compared."""
import glob
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-14                   # relative to the largest entry of the array: a few roundings (pow vs x*x, sum orders kept)


def cases():
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN, "kats_*.npz"))):
        d = np.load(path)
        for key in d.files:
            if key.endswith("|kernel"):
                out.append((os.path.basename(path), key[:-len("|kernel")]))
    return out


@pytest.fixture(scope="module")
def rt():
    sys.path.insert(0, os.path.join(ROOT, "openlbmpm_amd", "dropin"))
    import _runtime
    yield _runtime
    sys.path.remove(os.path.join(ROOT, "openlbmpm_amd", "dropin"))


@pytest.mark.parametrize("fixture,case", cases())
def test_kernel_reproduces_the_reference_kernel(rt, fixture, case):
    d = np.load(os.path.join(GOLDEN, fixture))
    module, kernel = str(d[case + "|module"]), str(d[case + "|kernel"])
    names = [str(n) for n in d[case + "|args"]]
    from openlbmpm_amd._kernel_specs import KERNELS
    assert tuple(names) == KERNELS[(module, kernel)][2], "the entry point must take the reference kernel's arguments, in its order"
    values, arrays = {}, {}
    for n in names:
        v = d["%s|in|%s" % (case, n)]
        if v.ndim == 0:
            values[n] = v.item()
        else:
            arrays[n] = values[n] = rt.to_device(np.ascontiguousarray(v))
    rt.launch_by_name(module, kernel, values)
    changed = 0
    for n, dev in arrays.items():
        want, got, before = d["%s|out|%s" % (case, n)], dev.copy_to_host(), d["%s|in|%s" % (case, n)]
        if want.dtype.kind in "ib":
            assert np.array_equal(got, want), (case, n)
        else:
            assert np.array_equal(np.isnan(got), np.isnan(want)), (case, n)
            scale = np.nanmax(np.abs(want)) or 1.0
            err = np.nanmax(np.abs(got - want)) / scale if want.size else 0.0
            assert err < TOL, (case, n, err)
        changed += not np.array_equal(want, before, equal_nan=True)
    assert bool(d[case + "|noop"]) == (changed == 0)        # (two reference kernels change nothing at all: O:320 with two fluids, E:38)


def test_kernels_that_cannot_run_in_the_reference_say_so(rt):
    """AccelerateTransport2DRK.py:927 raises IndexError under the stand-in for every input (9 weights into a 5-entry shared
    array, :938-939; recorded by the generator), :596 for every node beyond the fifth (unitVY[node], :624): the entry
    points refuse instead of guessing"""
    from openlbmpm_amd._lib import LbmpmError
    d = np.load(os.path.join(GOLDEN, "kats_tr.npz"))
    assert str(d["calUpdateConcInTransportDomainByVQ9|raises"]) == "IndexError"
    case = "calCollisionTransportQuadraticEqlMRTGPU"
    values = {}
    for n in d[case + "|args"]:
        v = d["%s|in|%s" % (case, n)]
        values[str(n)] = v.item() if v.ndim == 0 else rt.to_device(np.ascontiguousarray(v))
    assert values["totalNodes"] == 5
    with pytest.raises(LbmpmError, match="totalNodes <= 5"):
        rt.launch_by_name("tr", case, dict(values, totalNodes=6))
    N = 8
    dev = lambda a: rt.to_device(np.ascontiguousarray(a))
    z = dict(totalNodes=N, numTracers=1, xDim=64, totalTracer=dev(np.ones(1)), totalOld=dev(np.ones(1)), transportDomain=dev(np.ones(N, dtype=bool)),
             physicalVX=dev(np.zeros(N)), physicalVY=dev(np.zeros(N)), unitVX=dev(np.zeros(9)), unitVY=dev(np.zeros(9)), weightsCoeff=dev(np.zeros(9)),
             tracerConc=dev(np.zeros((1, N))), tracerPDF=dev(np.zeros((1, N, 9))))
    with pytest.raises(LbmpmError, match="cannot run in the reference"):
        rt.launch_by_name("tr", "calUpdateConcInTransportDomainByVQ9", z)
    with pytest.raises(TypeError, match="bool"):
        rt.launch_by_name("tr", "calUpdateConcInTransportDomainByVQ9", dict(z, transportDomain=dev(np.ones(N))))
