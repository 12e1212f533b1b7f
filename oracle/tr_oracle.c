/*
 * oracle/tr_oracle.c -- CPU restatement of the reference's D2Q5 tracer-transport kernels
 * (RKCG2D/AccelerateTransport2DRK.py, "T:") and of the tracer sub-step of the coupled
 * colour-gradient + transport loop (RKCG2D/Transport2DRK.py:1341-1418, "TD:").
 *
 * TEST INFRASTRUCTURE ONLY (rules in oracle/rk_oracle.c).
 *
 * Parity status: every kernel is PINNED individually against vectors produced by the real
 * reference kernels (tests/golden/gen/make_golden_tr.py -> tests/golden/tr_kernels.npz).  The
 * ORDER of the sub-step inside the flow step is pinned by two captures of the real driver
 * runTransport2DMPMCRKNew, which runs once three listed defects are repaired in memory (:1358 indentation,
 * :1293 undefined kernel name, the missing transportsetup.ini): tests/golden/gen/make_golden_tr_coupled.py ->
 * tests/golden/trc_*.npz, compared in tests/test_tr_coupled.py.
 *
 * Layout = the reference's: g[nT][N][5] (0 rest, 1 E, 2 W, 3 N, 4 S; TD:60-61), C[nT][N],
 * nbr4[4N] in the order E,W,N,S with -1 for every non-fluid neighbour (T:51-75).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef int64_t i64;

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

static const double VX[5] = {0., 1., -1., 0., 0.};
static const double VY[5] = {0., 0., 0., 1., -1.};
static const double WT5[5] = {1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.};
#define Gq(g, t, n, j) (g)[((size_t)(t) * N + (n)) * 5 + (j)]
#define Cc(c, t, n) (c)[(size_t)(t) * N + (n)]

/* T:51-75 fillNeighboringNodesTransport */
void tr_fill_neighbors(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *newIndex, i64 *nbr)
{
    PARFOR
    for (i64 k = 0; k < N; ++k) {
        i64 loc = fluidNodes[k], i = loc / nx, j = loc % nx;
        i64 Fw = (j < nx - 1) ? j + 1 : 0, Bw = (j > 0) ? j - 1 : nx - 1;
        i64 Up = (i < ny - 1) ? i + 1 : 0, Lo = (i > 0) ? i - 1 : ny - 1;
        nbr[4 * k] = newIndex[i * nx + Fw];
        nbr[4 * k + 1] = newIndex[i * nx + Bw];
        nbr[4 * k + 2] = newIndex[Up * nx + j];
        nbr[4 * k + 3] = newIndex[Lo * nx + j];
    }
}

/* T:78-90 calConcentrationGPU */
void tr_concentration(i64 N, int nT, double *C, const double *g)
{
    for (int t = 0; t < nT; ++t) {
        PARFOR
        for (i64 n = 0; n < N; ++n) {
            double c = 0.;
            for (int j = 0; j < 5; ++j) c += Gq(g, t, n, j);
            Cc(C, t, n) = c;
        }
    }
}

/* T:535-590 calCollisionTransportLinearEqlMRTGPU: g += A (M g - M g_eq), A = -M^-1 S^-1,
 * g_eq = C w (1 + 3 e.u) */
void tr_collide_mrt(i64 N, int nT, const double *vx, const double *vy, const double *C, double *g,
                    const double *M /*[5][5]*/, const double *A /*[nT][5][5]*/)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int t = 0; t < nT; ++t) {
            double eq[5], diff[5], d[5];
            for (int j = 0; j < 5; ++j)
                eq[j] = Cc(C, t, n) * WT5[j] * (1. + 3. * (VX[j] * vx[n] + VY[j] * vy[n]));
            for (int j = 0; j < 5; ++j) {
                double ve = 0., vp = 0.;
                for (int k = 0; k < 5; ++k) { ve += M[5 * j + k] * eq[k]; vp += Gq(g, t, n, k) * M[5 * j + k]; }
                diff[j] = vp - ve;
            }
            for (int j = 0; j < 5; ++j) {
                double v = 0.;
                for (int k = 0; k < 5; ++k) v += A[(size_t)t * 25 + 5 * j + k] * diff[k];
                d[j] = v;
            }
            for (int j = 0; j < 5; ++j) Gq(g, t, n, j) = Gq(g, t, n, j) + d[j];
        }
}

/* T:957-970 calValueTransportDomain */
void tr_indicator(i64 N, double crit, double *ind, const double *rhoR)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) ind[n] = (rhoR[n] > crit) ? -(1. - 1.) : -(1. - 0.);
}

/* T:976-1013 calTransportWithInterfaceD2Q5 */
void tr_interface(i64 N, int nT, const double *beta, const double *ind, const double *Gx, const double *Gy,
                  const double *C, double *g)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double gn = sqrt(Gx[n] * Gx[n] + Gy[n] * Gy[n]);
        double ux, uy, un;
        if (gn > 1.0e-8) { ux = -Gx[n] / gn; uy = -Gy[n] / gn; un = sqrt(ux * ux + uy * uy); }
        else { ux = 0.; uy = 0.; un = 0.; }
        for (int t = 0; t < nT; ++t)
            for (int j = 0; j < 4; ++j) {
                double eq = WT5[j + 1] * Cc(C, t, n);
                double en = sqrt(VX[j + 1] * VX[j + 1] + VY[j + 1] * VY[j + 1]);
                double c = 0.;
                if (un > 1.0e-8 && en > 1.0e-8) c = (VX[j + 1] * ux + VY[j + 1] * uy) / (en * un);
                Gq(g, t, n, j + 1) = Gq(g, t, n, j + 1) + beta[t] * ind[n] * eq * c;
            }
    }
}

/* T:461-478 calFreeConcBoundary3: row 0 <- N neighbour (all five populations) */
void tr_free_outlet(i64 N, int nT, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *g)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < nx && loc >= 0)) continue;
        i64 q = nbr[4 * n + 2];
        for (int t = 0; t < nT; ++t)
            for (int j = 0; j < 5; ++j) Gq(g, t, n, j) = Gq(g, t, q, j);
    }
}

/* T:139-182 calStreamingTransportGPU + T:184-194 calStreamingTransport2GPU */
void tr_stream(i64 N, int nT, const i64 *nbr, double *g, double *gNew)
{
    static const int OPP5[5] = {0, 2, 1, 4, 3};
    for (int t = 0; t < nT; ++t) {
        PARFOR
        for (i64 n = 0; n < N; ++n)
            for (int j = 1; j < 5; ++j) {
                i64 q = nbr[4 * n + j - 1];
                if (q != -1) Gq(gNew, t, q, j) = Gq(g, t, n, j);
                else Gq(gNew, t, n, OPP5[j]) = Gq(g, t, n, j);
            }
    }
    for (int t = 0; t < nT; ++t) {
        PARFOR
        for (i64 n = 0; n < N; ++n)
            for (int j = 1; j < 5; ++j) Gq(g, t, n, j) = Gq(gNew, t, n, j);
    }
}

/* T:682-698 calInamuroConstConcBoundary (ghost row ny-1): g_4 = C_in - (g0+g1+g2+g3) */
void tr_inlet_inamuro(i64 N, int nT, i64 ny, i64 nx, const i64 *fluidNodes, const double *cb, double *g)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= (ny - 1) * nx && loc < ny * nx)) continue;
        for (int t = 0; t < nT; ++t) {
            double s = Gq(g, t, n, 0) + Gq(g, t, n, 1) + Gq(g, t, n, 2) + Gq(g, t, n, 3);
            double u = (cb[t] - s) / WT5[4];
            Gq(g, t, n, 4) = WT5[4] * u;
        }
    }
}

/* T:95-111 calReactionTracersGPU: A + B -> C between tracers 0, 1 (consumed) and 2 (produced);
 * source k C_0 C_1 spread over the populations with the weights J (diffJED, TD:404-410) */
void tr_reaction(i64 N, int nT, const double *rate, const double *J /*[nT][5]*/, const double *C, double *g)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double S[3];
        S[0] = -rate[0] * C[0 * N + n] * C[1 * N + n];
        S[1] = -rate[0] * C[0 * N + n] * C[1 * N + n];
        S[2] = rate[0] * C[0 * N + n] * C[1 * N + n];
        for (int i = 0; i < nT; ++i)
            for (int j = 0; j < 5; ++j) g[((i64)i * N + n) * 5 + j] = g[((i64)i * N + n) * 5 + j] + J[i * 5 + j] * S[i];
    }
}

typedef struct {
    i64 N, nx, ny;
    int nT, freeOutlet, dirichletInlet;
    const i64 *fluidNodes, *nbr4;
    const double *M, *A, *beta, *cb;
    double crit;
    double *g, *gNew, *C, *ind;
    int reaction;            /* [SystemType] Reaction = 'yes' (needs nT == 3) */
    double rate[1], J[4 * 5];
} tr_sim;

/* tracer sub-step of the coupled loop, TD:1341-1418 (D2Q5, MRT): uses rhoR, physical velocity and
 * the wetting-corrected colour gradient of the CURRENT flow step */
void tr_substep(tr_sim *s, const double *rhoR, const double *vx, const double *vy, const double *Gx, const double *Gy)
{
    tr_indicator(s->N, s->crit, s->ind, rhoR);
    tr_collide_mrt(s->N, s->nT, vx, vy, s->C, s->g, s->M, s->A);
    tr_interface(s->N, s->nT, s->beta, s->ind, Gx, Gy, s->C, s->g);
    if (s->reaction) tr_reaction(s->N, s->nT, s->rate, s->J, s->C, s->g);       /* TD:1358-1362 */
    if (s->freeOutlet) tr_free_outlet(s->N, s->nT, s->nx, s->fluidNodes, s->nbr4, s->g);
    tr_stream(s->N, s->nT, s->nbr4, s->g, s->gNew);
    if (s->dirichletInlet) tr_inlet_inamuro(s->N, s->nT, s->ny, s->nx, s->fluidNodes, s->cb, s->g);
    tr_concentration(s->N, s->nT, s->C, s->g);
}
