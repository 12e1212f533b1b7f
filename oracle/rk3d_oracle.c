/*
 * oracle/rk3d_oracle.c -- CPU statement of the D3Q19 colour-gradient model (perturbation
 * operator) that the reference only ships an ini for (IniFiles/RKtwophasesetup3D.ini; the
 * module RKColorGradientD3Q19 imported by main.py:22 is absent from the tree).
 *
 * TEST INFRASTRUCTURE ONLY (rules in oracle/rk_oracle.c).
 *
 * Parity status: PINNED BY REDUCTION.  There is no 3-D reference code, but there is the 2-D loop this model
 * extends (runRKColorGradient2DPerturbation, RKD2Q9.py:978-1223).  A y-uniform D3Q19 lattice projects onto
 * D2Q9 term by term -- weights, the B_i below onto the reference's constantBNew (RKD2Q9.py:131-133), gradient,
 * BGK, perturbation, both Zou-He closures, bounce-back -- except for the |e_i| inside cos(theta_i) of the
 * recolouring; with that one pair of weights set to its projection-exact value (rcA below) this file
 * reproduces the captures of the REAL 2-D driver (tests/golden/rkpert_srt_*.npz) to 1e-10 on rhoR, rhoB,
 * phi, u over 50-80 steps, inlet/outlet/ghost planes and solid-phi walls included, and with its own weights
 * the pinned D2Q9 oracle run with the projected ones (tests/test_rk3d_reduction.py).  What the reduction cannot
 * see: anything that vanishes on y-uniform fields (the y-components of gradient and closures; covered by the
 * x<->y symmetry tests of tests/test_oracle_rk3d.py) and the MRT option (the D2Q9 and D3Q19 moment bases do
 * not project onto each other; held to the BGK limit, basis orthogonality and physics tests).
 * The model is the D3Q19 extension of the reference's 2-D kernels, operator by operator:
 *   colour gradient   G = 3 sum_i w_i e_i phi(x+e_i), solid neighbours carry the constant
 *                     phi_s = (SolidRhoR-SolidRhoB)/(SolidRhoR+SolidRhoB)
 *                     (AcceleratedRKGPU2D.py:1199-1219)
 *   BGK on f_R, f_B   tau = 1/2 + 1/((1+phi)/(2(tauR-1/2)) + (1-phi)/(2(tauB-1/2)))   (A:1144-1160)
 *   MRT option        ([RelaxationType] Type = 'MRT', RKtwophasesetup3D.ini:53-55) as the 2-D MRT
 *                     kernel does it: f_tot -= M^-1 S M (f_tot - f_eq) (A:1938-2025), with the D3Q19
 *                     moment basis of d'Humieres et al. 2002, rates s_e 1.19, s_eps 1.4, s_pi 1.4, s_q = s_m 1.2
 *                     (his s_m = 1.98 is unstable at the open planes), conserved moments 0, stress moments 1/tau
 *   perturbation      f += (AkR+AkB)/2 |G| (w_i (e_i.G)^2/|G|^2 - B_i)                 (A:1225-1239)
 *                     with the D3Q19 B_i of Liu, Valocchi & Kang 2012: -1/3, 1/18, 1/36
 *   recolouring       f_R = rhoR/rho f + beta rhoR rhoB/rho^2 w_i cos(theta_i)         (A:1241-1267)
 *   streaming         push + in-place half-way bounce-back == pull (A:340-417)
 *   inlet  (z=nz-2)   Zou-He velocity per colour (A:657-695 -> Hecht & Harting 2010 D3Q19),
 *                     ghost plane nz-1 = copy with rho re-summed (A:607-650)
 *   outlet (z=1)      Zou-He pressure per colour (A:1008-1039), ghost plane 0 = copy (A:1045-1081)
 * Further self-consistency / physics tests: tests/test_oracle_rk3d.py.
 *
 * Layout here: dense AoS f[c][z][y][x][19] for clarity; x,y periodic, no wrap in z.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

#define Q 19
static const int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
static const int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
static const int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};
static const int OPP[Q] = {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17};
static double WT(int i) { return i == 0 ? 1. / 3. : (i < 7 ? 1. / 18. : 1. / 36.); }
static double BI(int i) { return i == 0 ? -1. / 3. : (i < 7 ? 1. / 18. : 1. / 36.); }

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

typedef struct {
    i64 nx, ny, nz;
    const uint8_t *dom;          /* [nz][ny][nx] 1 = fluid */
    double akR, akB, beta, tauR, tauB, solidPhi, vzR, vzB, rhoOutR, rhoOutB;
    int mrt;
    double *fR, *fB, *gR, *gB;   /* [nz*ny*nx][19] current / scratch */
    double *rhoR, *rhoB, *phi, *vx, *vy, *vz, *Gx, *Gy, *Gz;
    double rcA, rcD;             /* weights w_i / |e_i| of the recolouring term for |e_i| = 1 and sqrt 2; 0 = the model's
                                    own (1/18, 1/(36 sqrt 2)).  Other values exist for ONE purpose: rcA = 1/9 - 2/(36 sqrt 2)
                                    makes a y-uniform lattice project exactly onto the reference's D2Q9 loop
                                    (tests/test_rk3d_reduction.py; the D3Q19 w_i, B_i, gradient and Zou-He closures project
                                    onto their D2Q9 counterparts by themselves, the |e_i| of cos(theta_i) does not) */
    int inletP;                  /* 0: Zou-He velocity inlet (vzR, vzB); 1: Zou-He PRESSURE inlet per colour (rhoInR, rhoInB), the z-plane
                                    form of calConstPressureInletGPU, AcceleratedRKGPU2D.py:925-962, ghost plane as :968-1002 */
    double rhoInR, rhoInB;
    int conv;                    /* 1: convective outlet, AcceleratedRKGPU2D.py:700-784 as z planes: the planes 2, 1, 0 take the streamed
                                    populations of plane 3 (cell by cell; same masks) and re-sum their densities; no pressure rule on plane 1 */
} rk3d_sim;

static i64 wrap(i64 v, i64 n) { return v < 0 ? v + n : (v >= n ? v - n : v); }

static void zouhe_inlet(double uz, double *f, double *rho_out)
{   /* top plane, unknown e_z = -1: 6, 12, 13, 16, 17 */
    double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    double sp = f[5] + f[11] + f[14] + f[15] + f[18];
    double rho = (s0 + 2. * sp) / (1. + uz);
    double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[6] = f[5] - 1. / 3. * rho * uz;
    f[12] = f[11] - 1. / 6. * rho * uz + Nx;
    f[13] = f[14] - 1. / 6. * rho * uz - Nx;
    f[16] = f[15] - 1. / 6. * rho * uz + Ny;
    f[17] = f[18] - 1. / 6. * rho * uz - Ny;
    *rho_out = rho;
}

/* AcceleratedRKGPU2D.py:925-962 in 3-D: u_z = -1 + (S0 + 2 S+) / rho, unknown e_z = -1 as in zouhe_inlet */
static void zouhe_inlet_pressure(double rho, double *f)
{
    double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    double sp = f[5] + f[11] + f[14] + f[15] + f[18];
    double uz = -1. + (s0 + 2. * sp) / rho;
    double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[6] = f[5] - 1. / 3. * rho * uz;
    f[12] = f[11] - 1. / 6. * rho * uz + Nx;
    f[13] = f[14] - 1. / 6. * rho * uz - Nx;
    f[16] = f[15] - 1. / 6. * rho * uz + Ny;
    f[17] = f[18] - 1. / 6. * rho * uz - Ny;
}

static void zouhe_outlet(double rho, double *f)
{   /* bottom plane, unknown e_z = +1: 5, 11, 14, 15, 18 */
    double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    double sm = f[6] + f[12] + f[13] + f[16] + f[17];
    double uz = 1. - 1. / rho * (s0 + 2. * sm);
    double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[5] = f[6] + 1. / 3. * rho * uz;
    f[11] = f[12] + 1. / 6. * rho * uz - Nx;
    f[14] = f[13] + 1. / 6. * rho * uz + Nx;
    f[15] = f[16] + 1. / 6. * rho * uz - Ny;
    f[18] = f[17] + 1. / 6. * rho * uz + Ny;
}

static double sum19(const double *f) { double r = 0.; for (int i = 0; i < Q; ++i) r += f[i]; return r; }

/* boundary planes + macroscopic fields of the current (post-streaming) populations */
static void rk3d_bc_and_macro(rk3d_sim *s)
{
    i64 nx = s->nx, ny = s->ny, nz = s->nz, pl = nx * ny;
    PARFOR
    for (i64 z = 0; z < nz; ++z)
        for (i64 k = 0; k < pl; ++k) {
            i64 n = z * pl + k;
            if (!s->dom[n]) continue;
            s->rhoR[n] = sum19(s->fR + Q * n);
            s->rhoB[n] = sum19(s->fB + Q * n);
        }
    for (i64 k = 0; k < pl; ++k) {          /* inlet plane nz-2, ghost nz-1 */
        i64 n = (nz - 2) * pl + k, g = (nz - 1) * pl + k;
        if (s->dom[n]) {
            if (s->inletP) {
                zouhe_inlet_pressure(s->rhoInR, s->fR + Q * n); s->rhoR[n] = s->rhoInR;
                zouhe_inlet_pressure(s->rhoInB, s->fB + Q * n); s->rhoB[n] = s->rhoInB;
            } else {
                zouhe_inlet(s->vzR, s->fR + Q * n, &s->rhoR[n]);
                zouhe_inlet(s->vzB, s->fB + Q * n, &s->rhoB[n]);
            }
        }
        if (s->dom[g]) {
            memcpy(s->fR + Q * g, s->fR + Q * n, sizeof(double) * Q);
            memcpy(s->fB + Q * g, s->fB + Q * n, sizeof(double) * Q);
            if (s->inletP) { s->rhoR[g] = s->rhoR[n]; s->rhoB[g] = s->rhoB[n]; }      /* the ghost copies the densities too (A:968-1002) */
            else { s->rhoR[g] = sum19(s->fR + Q * g); s->rhoB[g] = sum19(s->fB + Q * g); }
        }
    }
    if (s->conv) {
        for (i64 z = 2; z >= 0; --z)
            for (i64 k = 0; k < pl; ++k) {
                i64 n = z * pl + k, up = (z + 1) * pl + k;
                if (!s->dom[n]) continue;
                memcpy(s->fR + Q * n, s->fR + Q * up, sizeof(double) * Q);
                memcpy(s->fB + Q * n, s->fB + Q * up, sizeof(double) * Q);
                s->rhoR[n] = sum19(s->fR + Q * n);
                s->rhoB[n] = sum19(s->fB + Q * n);
            }
    }
    for (i64 k = 0; k < pl && !s->conv; ++k) {          /* outlet plane 1, ghost 0 */
        i64 n = pl + k, g = k;
        if (s->dom[n]) {
            zouhe_outlet(s->rhoOutR, s->fR + Q * n); s->rhoR[n] = s->rhoOutR;
            zouhe_outlet(s->rhoOutB, s->fB + Q * n); s->rhoB[n] = s->rhoOutB;
        }
        if (s->dom[g]) {
            memcpy(s->fR + Q * g, s->fR + Q * n, sizeof(double) * Q);
            memcpy(s->fB + Q * g, s->fB + Q * n, sizeof(double) * Q);
            s->rhoR[g] = s->rhoR[n]; s->rhoB[g] = s->rhoB[n];
        }
    }
    PARFOR
    for (i64 n = 0; n < nz * pl; ++n) {
        if (!s->dom[n]) continue;
        const double *r = s->fR + Q * n, *b = s->fB + Q * n;
        double mx = 0., my = 0., mz = 0.;
        for (int i = 0; i < Q; ++i) { double t = r[i] + b[i]; mx += CX[i] * t; my += CY[i] * t; mz += CZ[i] * t; }
        double rho = s->rhoR[n] + s->rhoB[n];
        s->vx[n] = mx / rho; s->vy[n] = my / rho; s->vz[n] = mz / rho;
        s->phi[n] = (s->rhoR[n] - s->rhoB[n]) / (s->rhoR[n] + s->rhoB[n]);
    }
}

/* D3Q19 moment basis of d'Humieres, Ginzburg, Krafczyk, Lallemand & Luo (2002): rows
 * rho, e, eps, jx, qx, jy, qy, jz, qz, 3pxx, 3pixx, pww, piww, pxy, pyz, pxz, mx, my, mz */
static void mrt_basis(double M[Q][Q])
{
    for (int i = 0; i < Q; ++i) {
        double x = CX[i], y = CY[i], z = CZ[i], c2 = x * x + y * y + z * z;
        M[0][i] = 1.;
        M[1][i] = 19. * c2 - 30.;
        M[2][i] = (21. * c2 * c2 - 53. * c2 + 24.) / 2.;
        M[3][i] = x;  M[4][i] = (5. * c2 - 9.) * x;
        M[5][i] = y;  M[6][i] = (5. * c2 - 9.) * y;
        M[7][i] = z;  M[8][i] = (5. * c2 - 9.) * z;
        M[9][i] = 3. * x * x - c2;   M[10][i] = (3. * c2 - 5.) * (3. * x * x - c2);
        M[11][i] = y * y - z * z;    M[12][i] = (3. * c2 - 5.) * (y * y - z * z);
        M[13][i] = x * y; M[14][i] = y * z; M[15][i] = x * z;
        M[16][i] = (y * y - z * z) * x; M[17][i] = (z * z - x * x) * y; M[18][i] = (x * x - y * y) * z;
    }
}

/* |row k|^2 of the basis (small integers, exact in floating point) */
static void mrt_norms(const double M[Q][Q], double nrm[Q])
{
    for (int k = 0; k < Q; ++k) {
        double a = 0.;
        for (int i = 0; i < Q; ++i) a += M[k][i] * M[k][i];
        nrm[k] = a;
    }
}

/* d <- M^-1 S M d; the rows of M are mutually orthogonal, so M^-1 = M^T diag(1/|row|^2) */
static void mrt_relax_with(const double M[Q][Q], const double nrm[Q], const double S[Q], double d[Q])
{
    double m[Q], out[Q];
    for (int k = 0; k < Q; ++k) {
        double acc = 0.;
        for (int i = 0; i < Q; ++i) acc += M[k][i] * d[i];
        m[k] = S[k] * acc / nrm[k];
    }
    for (int i = 0; i < Q; ++i) {
        double acc = 0.;
        for (int k = 0; k < Q; ++k) acc += M[k][i] * m[k];
        out[i] = acc;
    }
    for (int i = 0; i < Q; ++i) d[i] = out[i];
}

/* s_e, s_eps, s_q, s_pi, s_m.  d'Humieres' own s_m = 1.98 lets a mode grow at the Zou-He planes (a 24^3
 * static droplet is gone after ~1500 steps; tests/test_oracle_rk3d.py); the two third-order moment
 * families therefore share 1.2. */
static double RATES[5] = {1.19, 1.4, 1.2, 1.4, 1.2};
void rk3d_set_mrt_rates_public(const double *r) { for (int i = 0; i < 5; ++i) RATES[i] = r[i]; }
static void mrt_relax_matrix(const double M[Q][Q], const double nrm[Q], double inv_tau, double d[Q])
{
    const double S[Q] = {0., RATES[0], RATES[1], 0., RATES[2], 0., RATES[2], 0., RATES[2], inv_tau, RATES[3], inv_tau, RATES[3],
                         inv_tau, inv_tau, inv_tau, RATES[4], RATES[4], RATES[4]};
    mrt_relax_with(M, nrm, S, d);
}

/* The same operator without the two 19 x 19 products, for d = f - feq (no mass, no momentum) and s_q = s_m:
 * the odd part of d (differences of opposite pairs) then holds only q and m moments and relaxes at s_q as a
 * whole; of the even part the stress moments relax at 1/tau, so
 *   Delta = s_q odd(d) + (1/tau) even(d) + sum_{k in e, eps, pi_xx, pi_ww} (s_k - 1/tau) M_k (M_k . d) / |M_k|^2.
 * (What the timed CPU baseline runs; tests/test_oracle_rk3d.py holds it to the matrix form.) */
static void mrt_relax_fast(const double M[Q][Q], const double nrm[Q], double inv_tau, double d[Q])
{
    static const int ROWS[4] = {1, 2, 10, 12};
    const double rate[4] = {RATES[0], RATES[1], RATES[3], RATES[3]};
    double c[4];
    for (int a = 0; a < 4; ++a) {
        const double *m = M[ROWS[a]];
        double acc = 0.;
        for (int i = 0; i < Q; ++i) acc += m[i] * d[i];
        c[a] = (rate[a] - inv_tau) * acc / nrm[ROWS[a]];
    }
    double out[Q];
    for (int i = 0; i < Q; ++i) {
        double ev = 0.5 * (d[i] + d[OPP[i]]), od = 0.5 * (d[i] - d[OPP[i]]);
        double corr = 0.;
        for (int a = 0; a < 4; ++a) corr += M[ROWS[a]][i] * c[a];
        out[i] = RATES[2] * od + inv_tau * ev + corr;
    }
    for (int i = 0; i < Q; ++i) d[i] = out[i];
}

static void mrt_relax(const double M[Q][Q], const double nrm[Q], double inv_tau, double d[Q])
{
    if (RATES[2] == RATES[4]) mrt_relax_fast(M, nrm, inv_tau, d);
    else mrt_relax_matrix(M, nrm, inv_tau, d);
}
void rk3d_mrt_relax_both_public(double inv_tau, const double *d_in, double *d_matrix, double *d_fast)
{
    double M[Q][Q], nrm[Q];
    mrt_basis(M);
    mrt_norms(M, nrm);
    memcpy(d_matrix, d_in, sizeof(double) * Q); memcpy(d_fast, d_in, sizeof(double) * Q);
    mrt_relax_matrix(M, nrm, inv_tau, d_matrix);
    mrt_relax_fast(M, nrm, inv_tau, d_fast);
}

/* test hooks: the basis itself, and d <- M^-1 diag(S) M d for any S */
void rk3d_mrt_basis_public(double *M19x19) { mrt_basis((double (*)[Q])M19x19); }
void rk3d_mrt_relax_public(const double *S, double *d)
{
    double M[Q][Q], nrm[Q];
    mrt_basis(M);
    mrt_norms(M, nrm);
    mrt_relax_with(M, nrm, S, d);
}

static void rk3d_collide_stream(rk3d_sim *s)
{
    i64 nx = s->nx, ny = s->ny, nz = s->nz, pl = nx * ny;
    double M[Q][Q], nrm[Q];
    mrt_basis(M);
    mrt_norms(M, nrm);
    PARFOR
    for (i64 z = 0; z < nz; ++z)
        for (i64 y = 0; y < ny; ++y)
            for (i64 x = 0; x < nx; ++x) {
                i64 n = z * pl + y * nx + x;
                if (!s->dom[n]) continue;
                i64 nb[Q];
                double gx = 0., gy = 0., gz = 0.;
                for (int i = 1; i < Q; ++i) {
                    i64 zz = z + CZ[i];
                    nb[i] = -1;
                    double ph = s->solidPhi;
                    if (zz >= 0 && zz < nz) {
                        i64 q = zz * pl + wrap(y + CY[i], ny) * nx + wrap(x + CX[i], nx);
                        if (s->dom[q]) { nb[i] = q; ph = s->phi[q]; }
                    }
                    gx += 3. * WT(i) * CX[i] * ph; gy += 3. * WT(i) * CY[i] * ph; gz += 3. * WT(i) * CZ[i] * ph;
                }
                s->Gx[n] = gx; s->Gy[n] = gy; s->Gz[n] = gz;
                double g2 = gx * gx + gy * gy + gz * gz, gn = sqrt(g2);
                double rR = s->rhoR[n], rB = s->rhoB[n], rho = rR + rB, phi = s->phi[n];
                double tau = 0.5 + 1. / ((1. + phi) / (2. * (s->tauR - 0.5)) + (1. - phi) / (2. * (s->tauB - 0.5)));
                double ux = s->vx[n], uy = s->vy[n], uz = s->vz[n], usq = ux * ux + uy * uy + uz * uz;
                double *r = s->fR + Q * n, *b = s->fB + Q * n;
                double dm[Q];
                if (s->mrt) {
                    for (int i = 0; i < Q; ++i) {
                        double eu = CX[i] * ux + CY[i] * uy + CZ[i] * uz;
                        dm[i] = (r[i] + b[i]) - rho * WT(i) * (1. + 3. * eu + 4.5 * eu * eu - 1.5 * usq);
                    }
                    mrt_relax(M, nrm, 1. / tau, dm);
                }
                for (int i = 0; i < Q; ++i) {
                    double eu = CX[i] * ux + CY[i] * uy + CZ[i] * uz;
                    double feq = rho * WT(i) * (1. + 3. * eu + 4.5 * eu * eu - 1.5 * usq);
                    double ft = r[i] + b[i];
                    ft = s->mrt ? ft - dm[i] : ft - (ft - feq) / tau;
                    double eg = CX[i] * gx + CY[i] * gy + CZ[i] * gz;
                    if (g2 != 0.) ft += (s->akR + s->akB) * 0.5 * gn * (WT(i) * (eg * eg) / g2 - BI(i));
                    double en = sqrt((double)(CX[i] * CX[i] + CY[i] * CY[i] + CZ[i] * CZ[i]));
                    double c = (en == 0. || gn == 0.) ? 0. : eg / (en * gn);
                    double a = (s->beta * rR * rB / (rho * rho)) * WT(i) * c;
                    if (s->rcA > 0. || s->rcD > 0.) {
                        double wa = s->rcA > 0. ? s->rcA : 1. / 18., wd = s->rcD > 0. ? s->rcD : 1. / 36. / sqrt(2.);
                        a = (s->beta * rR * rB / (rho * rho)) * (i == 0 ? 0. : (i < 7 ? wa : wd)) * (gn == 0. ? 0. : eg / gn);
                    }
                    double pr = rR / rho * ft + a, pb = rB / rho * ft - a;
                    /* push with half-way bounce-back */
                    if (i == 0) { s->gR[Q * n] = pr; s->gB[Q * n] = pb; }
                    else if (nb[i] >= 0) { s->gR[Q * nb[i] + i] = pr; s->gB[Q * nb[i] + i] = pb; }
                    else { s->gR[Q * n + OPP[i]] = pr; s->gB[Q * n + OPP[i]] = pb; }
                }
            }
    double *t = s->fR; s->fR = s->gR; s->gR = t;
    t = s->fB; s->fB = s->gB; s->gB = t;
}

void rk3d_run(rk3d_sim *s, i64 nsteps)
{
    for (i64 k = 0; k < nsteps; ++k) { rk3d_bc_and_macro(s); rk3d_collide_stream(s); }
}

/* densities / velocity / phi of the current state without touching the populations */
void rk3d_observe(rk3d_sim *s)
{
    i64 N = s->nx * s->ny * s->nz;
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        if (!s->dom[n]) continue;
        s->rhoR[n] = sum19(s->fR + Q * n);
        s->rhoB[n] = sum19(s->fB + Q * n);
    }
}

void rk3d_init(rk3d_sim *s, const double *rhoR0, const double *rhoB0)
{
    i64 N = s->nx * s->ny * s->nz;
    for (i64 n = 0; n < N; ++n)
        for (int i = 0; i < Q; ++i) {
            s->fR[Q * n + i] = s->dom[n] ? WT(i) * rhoR0[n] : 0.;
            s->fB[Q * n + i] = s->dom[n] ? WT(i) * rhoB0[n] : 0.;
            s->gR[Q * n + i] = 0.; s->gB[Q * n + i] = 0.;
        }
}

void rk3d_bc_and_macro_public(rk3d_sim *s) { rk3d_bc_and_macro(s); }
