/*
 * oracle/rk_pert_oracle.c -- CPU restatement of the reference's PERTURBATION colour-gradient D2Q9 path
 * (RKColorGradientLBM.runRKColorGradient2DPerturbation, RKCG2D/RKD2Q9.py:978-1223, and the kernels of
 * RKCG2D/AcceleratedRKGPU2D.py only that loop calls).  This is the path the D3Q19 model extends, and
 * the pin of oracle/rk3d_oracle.c: a y-uniform D3Q19 lattice must reproduce this D2Q9 loop
 * (tests/test_rk3d_reduction.py).
 *
 * TEST INFRASTRUCTURE ONLY (rules in oracle/rk_oracle.c).
 *
 * Parity status: PINNED.  Kernel by kernel against tests/golden/rkpert_kernels.npz and loop against
 * tests/golden/rkpert_{srt_capillary,srt_porous,mrt_capillary}.npz, all produced by the real reference
 * kernels / the real driver under the numba stand-in (tests/golden/gen/make_golden_rk_pert.py, which lists
 * the four in-memory repairs the dead driver needs; R3 fixes where fluidPDFTotal is summed).
 *
 * Layout and citations as in oracle/rk_oracle.c ("A:" = RKCG2D/AcceleratedRKGPU2D.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef int64_t i64;

static const double EX[9] = {0., 1., 0., -1., 0., 1., -1., -1., 1.};
static const double EY[9] = {0., 0., 1., 0., -1., 1., 1., -1., -1.};
static const double WT[9] = {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9.,
                             1. / 36., 1. / 36., 1. / 36., 1. / 36.};

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

/* from rk_oracle.c */
void rk_stream1(i64 N, const i64 *nbr, const double *f, double *fNew);
void rk_stream2(i64 N, const double *fNew, double *f);
void rk_macro_density(i64 N, const double *fR, const double *fB, double *rhoR, double *rhoB);
void rk_total_pdf(i64 N, const double *fR, const double *fB, double *fT);
void rk_phase_field(i64 N, const double *rhoR, const double *rhoB, double *phi);
void rk_ghost_inlet_velocity(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr,
                             double *rhoR, double *rhoB, double *fR, double *fB);
void rk_ghost_outlet_pressure(i64 N, i64 nx, const i64 *nbr, double *rhoR, double *rhoB,
                              double *fR, double *fB);

/* A:170-176 calEquilibriumRK2D */
static inline double feq(double rho, double w, double ex, double ey, double vx, double vy)
{
    return rho * w * (1 + (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) * (ex * vx + ey * vy) -
                           1.5 * (vx * vx + vy * vy)));
}

/* A:125-147 calPhysicalVelocityRKGPU2D: u = sum e (fR + fB) / (rhoB + rhoR), no force term */
void rk_pert_velocity(i64 N, const double *fR, const double *fB, const double *rhoR, const double *rhoB,
                      double *vx, double *vy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        const double *r = fR + 9 * n, *b = fB + 9 * n;
        double rho = rhoB[n] + rhoR[n];
        double tx = r[1] - r[3] + r[5] - r[6] - r[7] + r[8] + b[1] - b[3] + b[5] - b[6] - b[7] + b[8];
        vx[n] = tx / rho;
        double ty = r[2] - r[4] + r[5] + r[6] - r[7] - r[8] + b[2] - b[4] + b[5] + b[6] - b[7] - b[8];
        vy[n] = ty / rho;
    }
}

/* tau of A:1144-1145 / A:1307-1308: harmonic in phi, no delta window */
static inline double tau_phi(double phi, double tauR, double tauB)
{
    return 0.5 + 1. / ((1. + phi) / (2. * (tauR - 0.5)) + (1. - phi) / (2. * (tauB - 0.5)));
}

/* A:1125-1163 calRKCollision1GPU2DSRTNew: BGK on each colour, in place */
void rk_pert_collide1_srt(i64 N, double tauR, double tauB, const double *vx, const double *vy, const double *rhoR,
                          const double *rhoB, const double *phi, double *fR, double *fB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double tau = tau_phi(phi[n], tauR, tauB);
        for (int i = 0; i < 9; ++i) {
            double eR = feq(rhoR[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            double cR = -1. / tau * (fR[9 * n + i] - eR);
            double eB = feq(rhoB[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            double cB = -1. / tau * (fB[9 * n + i] - eB);
            fR[9 * n + i] = fR[9 * n + i] + cR;
            fB[9 * n + i] = fB[9 * n + i] + cB;
        }
    }
}

/* A:1272-1343 calRKCollision1GPU2DMRTNew: fT += -Minv S M (fT - feq) + wF_i (e_i . bodyF); S[7] = S[8] = 1/tau */
void rk_pert_collide1_mrt(i64 N, double tauR, double tauB, double bodyFX, double bodyFY, const double *vx, const double *vy,
                          const double *rhoR, const double *rhoB, const double *phi, double *fT, const double *M,
                          const double *Minv, const double *S)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double cs[9], eq[9], m[9];
        for (int i = 0; i < 9; ++i) cs[i] = S[i];
        double tau = tau_phi(phi[n], tauR, tauB);
        cs[7] = 1. / tau; cs[8] = 1. / tau;
        double rho = rhoB[n] + rhoR[n];
        for (int i = 0; i < 9; ++i) eq[i] = feq(rho, WT[i], EX[i], EY[i], vx[n], vy[n]);
        for (int i = 0; i < 9; ++i) {
            double a = 0., b = 0.;
            for (int j = 0; j < 9; ++j) { a += M[9 * i + j] * fT[9 * n + j]; b += M[9 * i + j] * eq[j]; }
            m[i] = a - b;
        }
        for (int i = 0; i < 9; ++i) m[i] = m[i] * cs[i];
        for (int i = 0; i < 9; ++i) {
            double a = 0.;
            for (int j = 0; j < 9; ++j) a += Minv[9 * i + j] * m[j];
            double wf = i == 0 ? 0. : (i < 5 ? 1. / 3. : 1. / 12.);
            fT[9 * n + i] = -a + wf * (EX[i] * bodyFX + EY[i] * bodyFY) + fT[9 * n + i];
        }
    }
}

/* A:1169-1267 calRKCollision23GPUNew: colour gradient from the neighbours' densities (solid neighbours
 * carry solidPhi), perturbation (AkR + AkB)/2 |G| (w (e.G)^2/|G|^2 - B_i) added to fT, recolouring of fT.
 * rw[9] = the weights of the recolouring term: the reference's are w_i (rw = NULL); a y-uniform D3Q19
 * lattice projects onto other ones (tests/test_rk3d_reduction.py). */
void rk_pert_collide23(i64 N, double beta, double AkR, double AkB, double solidPhi, const i64 *nbr, const double *Bc,
                       const double *rhoR, const double *rhoB, double *fR, double *fB, double *fT, double *Gx, double *Gy,
                       const double *rw)
{
    if (!rw) rw = WT;
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double gx = 0., gy = 0.;
        for (int i = 0; i < 8; ++i) {
            i64 q = nbr[8 * n + i];
            double ph = (q != -1) ? (rhoR[q] - rhoB[q]) / (rhoR[q] + rhoB[q]) : solidPhi;
            gx += 3. * WT[i + 1] * EX[i + 1] * ph;
            gy += 3. * WT[i + 1] * EY[i + 1] * ph;
        }
        if (Gx) { Gx[n] = gx; Gy[n] = gy; }
        double g2 = gx * gx + gy * gy, gn = sqrt(g2);
        double rs = rhoR[n] + rhoB[n], rm = rhoR[n] * rhoB[n], rs2 = rs * rs;
        for (int i = 0; i < 9; ++i) {
            double c2 = 0.;
            if (g2 != 0.) {
                double part = WT[i] * ((EX[i] * gx + EY[i] * gy) * (EX[i] * gx + EY[i] * gy)) / g2;
                c2 = (AkR + AkB) * 0.5 * gn * (part - Bc[i]);
            }
            fT[9 * n + i] += c2;
        }
        for (int i = 0; i < 9; ++i) {
            double en = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            double c = 0.;
            if (!(en == 0. || gn == 0.)) c = (EX[i] * gx + EY[i] * gy) / (sqrt(EX[i] * EX[i] + EY[i] * EY[i]) * gn);
            fR[9 * n + i] = rhoR[n] / rs * fT[9 * n + i] + (beta * rm / rs2) * rw[i] * c;
            fB[9 * n + i] = rhoB[n] / rs * fT[9 * n + i] - (beta * rm / rs2) * rw[i] * c;
        }
    }
}

/* A:657-695 constantVelocityZHBoundaryHigherRK: Zou-He velocity inlet per colour on grid row ny-2 */
void rk_pert_inlet_velocity(i64 N, i64 nx, i64 ny, double vyR, double vyB, const i64 *fluidNodes, double *rhoR, double *rhoB,
                            double *fR, double *fB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (ny - 1) * nx && loc >= (ny - 2) * nx)) continue;
        double *r = fR + 9 * n, *b = fB + 9 * n;
        rhoR[n] = (r[0] + r[1] + r[3] + 2. * (r[2] + r[5] + r[6])) / (1. + vyR);
        r[4] = r[2] - 2. / 3. * rhoR[n] * vyR;
        r[7] = r[5] + (r[1] - r[3]) / 2. - 1. / 6. * rhoR[n] * vyR;
        r[8] = r[6] - (r[1] - r[3]) / 2. - 1. / 6. * rhoR[n] * vyR;
        rhoB[n] = (b[0] + b[1] + b[3] + 2. * (b[2] + b[5] + b[6])) / (1. + vyB);
        b[4] = b[2] - 2. / 3. * rhoB[n] * vyB;
        b[7] = b[5] + (b[1] - b[3]) / 2. - 1. / 6. * rhoB[n] * vyB;
        b[8] = b[6] - (b[1] - b[3]) / 2. - 1. / 6. * rhoB[n] * vyB;
    }
}

/* A:1008-1039 calConstPressureLowerGPU: Zou-He pressure outlet per colour; the row test is on the COMPACT
 * index (nx <= n < 2 nx; equals grid row 1 when rows 0 and 1 are all fluid) */
void rk_pert_outlet_pressure(i64 N, i64 nx, double pLB, double pLR, double *rhoB, double *rhoR, double *fB, double *fR)
{
    PARFOR
    for (i64 n = nx; n < 2 * nx; ++n) {
        if (n >= N) continue;
        double *b = fB + 9 * n, *r = fR + 9 * n;
        double vB = 1. - 1. / pLB * (b[0] + b[1] + b[3] + 2. * (b[4] + b[7] + b[8]));
        b[2] = b[4] + 2. / 3. * (pLB * vB);
        b[5] = b[7] + 0.5 * (b[3] - b[1]) + 1. / 6. * pLB * vB;
        b[6] = b[8] + 0.5 * (b[1] - b[3]) + 1. / 6. * pLB * vB;
        rhoB[n] = pLB;
        double vR = 1. - 1. / pLR * (r[0] + r[1] + r[3] + 2. * (r[4] + r[7] + r[8]));
        r[2] = r[4] + 2. / 3. * pLR * vR;
        r[5] = r[7] + 0.5 * (r[3] - r[1]) + 1. / 6. * pLR * vR;
        r[6] = r[8] + 0.5 * (r[1] - r[3]) + 1. / 6. * pLR * vR;
        rhoR[n] = pLR;
    }
}

typedef struct {
    i64 N, nx, ny;
    const i64 *fluidNodes, *nbr;
    double beta, AkR, AkB, solidPhi, tauR, tauB, vyR, vyB, pLB, pLR;
    int mrt;
    const double *Bc, *rw, *M, *Minv, *S;
    double *fR, *fB, *fRn, *fBn, *fT, *rhoR, *rhoB, *vx, *vy, *phi, *Gx, *Gy;
} rk_pert_sim;

/* one time step in the order of RKD2Q9.py:1046-1223 (velocity inlet + pressure outlet) with repair R3 of
 * tests/golden/gen/make_golden_rk_pert.py: fT summed after collision 1 (SRT) / just before it (MRT) */
void rk_pert_step(rk_pert_sim *s)
{
    i64 N = s->N;
    rk_stream1(N, s->nbr, s->fR, s->fRn);
    rk_stream1(N, s->nbr, s->fB, s->fBn);
    rk_stream2(N, s->fRn, s->fR);
    rk_stream2(N, s->fBn, s->fB);
    rk_pert_outlet_pressure(N, s->nx, s->pLB, s->pLR, s->rhoB, s->rhoR, s->fB, s->fR);
    rk_ghost_outlet_pressure(N, s->nx, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_pert_inlet_velocity(N, s->nx, s->ny, s->vyR, s->vyB, s->fluidNodes, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_ghost_inlet_velocity(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_macro_density(N, s->fR, s->fB, s->rhoR, s->rhoB);
    rk_pert_velocity(N, s->fR, s->fB, s->rhoR, s->rhoB, s->vx, s->vy);
    rk_phase_field(N, s->rhoR, s->rhoB, s->phi);
    if (s->mrt) {
        rk_total_pdf(N, s->fR, s->fB, s->fT);
        rk_pert_collide1_mrt(N, s->tauR, s->tauB, 0., 0., s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fT, s->M, s->Minv, s->S);
    } else {
        rk_pert_collide1_srt(N, s->tauR, s->tauB, s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fR, s->fB);
        rk_total_pdf(N, s->fR, s->fB, s->fT);
    }
    rk_pert_collide23(N, s->beta, s->AkR, s->AkB, s->solidPhi, s->nbr, s->Bc, s->rhoR, s->rhoB, s->fR, s->fB, s->fT, s->Gx, s->Gy,
                      s->rw);
}

void rk_pert_run(rk_pert_sim *s, i64 nsteps)
{
    for (i64 k = 0; k < nsteps; ++k) rk_pert_step(s);
}

/* The same time step in the LITERAL order of RKD2Q9.py:1046-1223, i.e. without repair R3: fluidPDFTotal = fR + fB is summed right
 * after streaming (RKD2Q9.py:1065), before the boundary kernels and before collision 1, and calRKCollision23GPUNew recolours from
 * that sum -- the inlet / outlet rows and (SRT) the BGK relaxation of the two colours never reach the populations the next step
 * streams.  Kept so that the effect of R3 is a number (tests/test_oracle_rk_pert.py), not an argument. */
void rk_pert_step_literal(rk_pert_sim *s)
{
    i64 N = s->N;
    rk_stream1(N, s->nbr, s->fR, s->fRn);
    rk_stream1(N, s->nbr, s->fB, s->fBn);
    rk_stream2(N, s->fRn, s->fR);
    rk_stream2(N, s->fBn, s->fB);
    rk_total_pdf(N, s->fR, s->fB, s->fT);                                    /* RKD2Q9.py:1065 */
    rk_pert_outlet_pressure(N, s->nx, s->pLB, s->pLR, s->rhoB, s->rhoR, s->fB, s->fR);
    rk_ghost_outlet_pressure(N, s->nx, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_pert_inlet_velocity(N, s->nx, s->ny, s->vyR, s->vyB, s->fluidNodes, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_ghost_inlet_velocity(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    rk_macro_density(N, s->fR, s->fB, s->rhoR, s->rhoB);
    rk_pert_velocity(N, s->fR, s->fB, s->rhoR, s->rhoB, s->vx, s->vy);
    rk_phase_field(N, s->rhoR, s->rhoB, s->phi);
    if (s->mrt) rk_pert_collide1_mrt(N, s->tauR, s->tauB, 0., 0., s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fT, s->M, s->Minv, s->S);
    else rk_pert_collide1_srt(N, s->tauR, s->tauB, s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fR, s->fB);
    rk_pert_collide23(N, s->beta, s->AkR, s->AkB, s->solidPhi, s->nbr, s->Bc, s->rhoR, s->rhoB, s->fR, s->fB, s->fT, s->Gx, s->Gy,
                      s->rw);
}

void rk_pert_run_literal(rk_pert_sim *s, i64 nsteps)
{
    for (i64 k = 0; k < nsteps; ++k) rk_pert_step_literal(s);
}
