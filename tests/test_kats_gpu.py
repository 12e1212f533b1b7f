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


def test_lattice_constant_arrays_are_checked_not_ignored(rt):
    """The reference hands its direction vectors and weights to every kernel as device arrays; the kernels here have them built in.
    An array with other contents must not be swallowed: the entry point refuses it (LBMPM_ERR_UNSUPPORTED, the differing entry named);
    the reference's own tables pass, also from a second allocation (the verdict is cached per device pointer)."""
    N = 6
    rng = np.random.default_rng(5)
    W = np.array([4. / 9.] + [1. / 9.] * 4 + [1. / 36.] * 4)
    EX = np.array([0., 1., 0., -1., 0., 1., -1., -1., 1.]); EY = np.array([0., 0., 1., 0., -1., 1., 1., -1., -1.])
    rR, rB = rng.uniform(0.2, 0.8, N), rng.uniform(0.2, 0.8, N)
    base = dict(totalNodes=N, xDim=64, betaValue=0.7, fluidRhoR=rR, fluidRhoB=rB, gradientX=rng.normal(size=N), gradientY=rng.normal(size=N),
                fluidPDFR=np.zeros((N, 9)), fluidPDFB=np.zeros((N, 9)), fluidPDFTotal=rng.uniform(0.05, 0.2, (N, 9)))

    def launch(**lattice):
        v = dict(base, **lattice)
        dev = {k: (rt.to_device(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a) for k, a in v.items()}
        rt.launch_by_name("rk", "calRecoloringProcessM", dev)                           # A:1857-1899
        return dev["fluidPDFR"].copy_to_host()

    good = launch(weightsCoeff=W, unitEX=EX, unitEY=EY)
    assert np.isfinite(good).all() and np.abs(good).max() > 0
    again = launch(weightsCoeff=W.copy(), unitEX=EX.copy(), unitEY=EY.copy())
    assert np.array_equal(good, again)
    for name, bad in (("weightsCoeff", W * np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.000001])), ("unitEX", -EX), ("unitEY", EY[::-1].copy())):
        with pytest.raises(RuntimeError) as err:
            launch(**dict(dict(weightsCoeff=W, unitEX=EX, unitEY=EY), **{name: bad}))
        assert name in str(err.value) and "built in" in str(err.value), str(err.value)
