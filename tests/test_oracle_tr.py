"""Pins the tracer oracle (oracle/tr_oracle.c) kernel by kernel to vectors produced by the real
reference kernels of RKCG2D/AccelerateTransport2DRK.py (tests/golden/gen/make_golden_tr.py).
The coupled driver does not parse, so the sub-step ORDER is pinned by reading only."""
import ctypes as C
import os

import numpy as np

from helpers import GOLDEN, rel_err
from oracle import lib
from oracle.tr import tracer_matrices

F64P = C.POINTER(C.c_double); I64P = C.POINTER(C.c_int64)
P = lambda a, t=F64P: a.ctypes.data_as(t)
d = np.load(os.path.join(GOLDEN, "tr_kernels.npz"))
N = int(d["fluidNodes"].size); ny, nx = d["isDomain"].shape; nT = 2
L = lib()
TOL = 1e-13


def test_neighbour_table_and_matrices():
    newidx = -np.ones(nx * ny, dtype=np.int64); newidx[d["fluidNodes"]] = np.arange(N)
    nbr = np.empty(4 * N, np.int64)
    fl = np.ascontiguousarray(d["fluidNodes"])
    L.tr_fill_neighbors(C.c_int64(N), C.c_int64(nx), C.c_int64(ny), P(fl, I64P), P(newidx, I64P), P(nbr, I64P))
    assert np.array_equal(nbr, d["nbr"])
    M, A = tracer_matrices(d["diffX"], d["diffY"], float(d["dXY"]), float(d["dYX"]))
    assert np.array_equal(M, d["M"]) and rel_err(A, d["A"]) < 1e-15


def test_each_kernel():
    g = np.ascontiguousarray(d["conc_in_g"]); conc = np.zeros((nT, N))
    L.tr_concentration(C.c_int64(N), C.c_int(nT), P(conc), P(g))
    assert rel_err(conc, d["conc_out"]) < TOL
    g = np.ascontiguousarray(d["col_in_g"])
    M = np.ascontiguousarray(d["M"]); A = np.ascontiguousarray(d["A"])
    L.tr_collide_mrt(C.c_int64(N), C.c_int(nT), P(np.ascontiguousarray(d["col_vx"])), P(np.ascontiguousarray(d["col_vy"])),
                     P(np.ascontiguousarray(d["col_conc"])), P(g), P(M), P(A))
    assert rel_err(g, d["col_out_g"]) < TOL
    ind = np.zeros(N)
    L.tr_indicator(C.c_int64(N), C.c_double(0.5), P(ind), P(np.ascontiguousarray(d["ind_rhoR"])))
    assert np.array_equal(ind, d["ind_out"])
    g = np.ascontiguousarray(d["itf_in_g"])
    L.tr_interface(C.c_int64(N), C.c_int(nT), P(np.ascontiguousarray(d["itf_beta"])), P(ind),
                   P(np.ascontiguousarray(d["itf_Gx"])), P(np.ascontiguousarray(d["itf_Gy"])),
                   P(np.ascontiguousarray(d["col_conc"])), P(g))
    assert rel_err(g, d["itf_out_g"]) < TOL
    fl = np.ascontiguousarray(d["fluidNodes"]); nbr = np.ascontiguousarray(d["nbr"])
    g = np.ascontiguousarray(d["free_in_g"])
    L.tr_free_outlet(C.c_int64(N), C.c_int(nT), C.c_int64(nx), P(fl, I64P), P(nbr, I64P), P(g))
    assert np.array_equal(g, d["free_out_g"])
    g = np.ascontiguousarray(d["str_in_g"]); gn = np.zeros_like(g)
    L.tr_stream(C.c_int64(N), C.c_int(nT), P(nbr, I64P), P(g), P(gn))
    assert np.array_equal(g, d["str_out_g"])
    g = np.ascontiguousarray(d["ina_in_g"])
    L.tr_inlet_inamuro(C.c_int64(N), C.c_int(nT), C.c_int64(ny), C.c_int64(nx), P(fl, I64P),
                       P(np.ascontiguousarray(d["ina_cb"])), P(g))
    assert rel_err(g, d["ina_out_g"]) < TOL


def test_reaction_kernel():
    g = np.ascontiguousarray(d["rea_in_g"])
    L.tr_reaction(C.c_int64(N), C.c_int(3), P(np.ascontiguousarray(d["rea_rate"])), P(np.ascontiguousarray(d["rea_J"])),
                  P(np.ascontiguousarray(d["rea_conc"])), P(g))
    assert rel_err(g, d["rea_out_g"]) < TOL
    # A + B -> C: what the two reactants lose the product gains (sum_j J_ij = 1)
    dg = (g - d["rea_in_g"]).sum(axis=2)
    assert np.allclose(dg[0], dg[1], rtol=0, atol=1e-15) and np.allclose(dg[0], -dg[2], rtol=0, atol=1e-15)


def test_coupled_loop_conserves_tracer_in_closed_box():
    """no inlet/outlet for the tracer: total tracer mass is conserved by collide + interface +
    stream (bounce-back) to round-off, whatever the flow does."""
    from oracle.tr import CoupledOracle
    from oracle.rk import simple_geometry
    dom = simple_geometry(20, 48)
    ii, jj = np.mgrid[0:48, 0:20]
    rR = np.where((dom == 1) & (ii >= 30), 1.0, 0.0); rB = np.where((dom == 1) & (ii < 30), 1.0, 0.0)
    c0 = np.where((dom == 1) & (ii < 30), 0.5, 0.0)[None]
    o = CoupledOracle(dom, None, rR, rB, c0, dict(free_outlet=False, dirichlet_inlet=False))
    m0 = o.C.sum()
    o.run(50)
    assert np.isfinite(o.C).all() and abs(o.C.sum() - m0) / m0 < 1e-12
