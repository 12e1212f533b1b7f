/*
 * oracle/rk3d_csf_oracle.c -- CPU statement of the D3Q19 colour-gradient model with continuum-surface-force tension
 * ([SurfaceTension] SurfaceTensionType = 'CSF'): the loop of RKColorGradientLBM.runRKColorGradient2DCSF (RKD2Q9.py:1295-1490) carried to
 * three dimensions kernel by kernel, which is what SURVEY.md 8 a17 asks of the 3-D model ("extend a3-a11 to D3Q19 ... CSF kappa =
 * -div n in 3-D").  The reference ships no 3-D code (main.py:22 imports a module that is not in the tree).
 *
 * TEST INFRASTRUCTURE ONLY (rules in oracle/rk_oracle.c).
 *
 * Parity status: PINNED BY REDUCTION (SRT) -- a lattice that is uniform along y projects onto the reference's D2Q9 loop term by term,
 * and this file then reproduces the capture of the REAL 2-D driver (tests/golden/rk_csf_srt_capillary.npz) and the pinned 2-D oracle
 * on further SRT set-ups (tests/test_oracle_rk3d_csf.py).  What projects and why:
 *   weights            sum over c_y of the D3Q19 w_i = the D2Q9 w_i (1/3 + 2/18 = 4/9, 1/18 + 2/36 = 1/9, 1/36)
 *   phi on solids      weighted mean over fluid neighbours (A:1560-1581): numerator and denominator project
 *   gradient           G = 3 sum w_i e_i phi (A:1584-1634)
 *   wetting rule 2     Akai et al. 2018 (A:2430-2492) is a 3-D rule in vector form already: n = (cos t - c) n_s + c' u
 *   solid normals      the reference's 24-point weights 4/21, 4/45, 1/60, 2/315, 1/5040 (RKD2Q9.py:811-885) are the 2-D isotropic
 *                      E8 set of Sbragaglia et al. 2007; the 3-D E8 set of the same paper (92 points, |c|^2 = 1, 2, 3, 4, 5, 6, 8 with
 *                      4/45, 1/21, 2/105, 5/504, 1/315, 1/630, 1/5040) sums along one axis to exactly those five numbers
 *   curvature          K = -(I - n n) : grad n; A:2512-2551 is its 2-D form n_x n_y (d_y n_x + d_x n_y) - n_y^2 d_x n_x - n_x^2 d_y n_y,
 *                      derivatives 3 sum w_i e_i n(x + e_i) over fluid neighbours
 *   BGK, Guo source    A:1804-1848, A:1743-1798 (linear in the populations / polynomial in e_i)
 *   recolouring        A:1857-1899: beta rhoR rhoB / rho w_i cos(theta_i) |e_i| -- the |e_i| cancels, unlike in the perturbation loop
 *   inlet z = nz-2     non-equilibrium bounce-back on f_tot with the ratioB quirk (A:2348-2412), ghost plane (A:607-650);
 *                      or Zou-He pressure per colour (A:925-962; Hecht & Harting 2010 transverse terms), ghost (A:968-1002)
 *   outlet z = 1       Zou-He pressure on f_tot (A:2560-2590), ghost (A:1045-1081); or the convective copies (A:700-784)
 *   topology           periodic wrap on every edge like fillNeighboringNodes (A:15-53); walls are what the mask says
 * NOT covered by the reduction: the MRT option -- the D3Q19 basis of d'Humieres et al. 2002 with the rates of oracle/rk3d_oracle.c and
 * the Guo source in moment space, f += M^-1 (I - S/2) M src (A:2027-2113) -- held to the BGK limit (all rates 1/tau) and to symmetry
 * tests; wetting rule 1 (Xu 2017, A:1639-1679) is a 2-D rotation and has no 3-D form: not offered.
 *
 * Layout: dense AoS f[z][y][x][19]; phi carries the wetting solids' values in the same dense array.
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

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

typedef struct {
    i64 nx, ny, nz;
    const uint8_t *dom;          /* [nz][ny][nx] 1 = fluid */
    double sigma, cosT, sinT, beta, delta, tauR, tauB;
    double vzIn;                 /* velocityZB + velocityZR (D:1300) */
    double pInB, pInR;           /* densityBH, densityRH */
    double pOutTotal;            /* densityBL + densityRL (D:1344) */
    int tauType, mrt, inletType /* 0 'Neumann' 1 'Dirichlet' */, outletType /* 0 'Dirichlet' 1 'Convective' */, wetting /* 0 | 2 */;
    double *fR, *fB, *gR, *gB, *fT;                       /* [N][19] */
    double *rhoR, *rhoB, *vx, *vy, *vz, *phi, *Gx, *Gy, *Gz, *Fx, *Fy, *Fz, *K;
    double *nsx, *nsy, *nsz;     /* solid normal at the fluid cells next to solid (kind 3), dense */
    uint8_t *kind;               /* 0 solid, 1 fluid, 2 wetting solid (>= 1 fluid among its 18 neighbours), 3 fluid with >= 1 solid among them */
    i64 W;                       /* wetting solids in the lattice (the `if W > 0` of D:1405-1424) */
    double rates[6];             /* MRT: s_e, s_eps, s_q, s_pi, s_m, and the rate of the conserved moments (0 in the model, as S[0] = S[3] = S[5] = 0 in
                                    RKD2Q9.py:338-340; the BGK-limit test sets all six to 1/tau: the velocity carries half the force of the step BEFORE,
                                    so the momentum of f - feq does not cancel against the source's and its rate matters) */
    double crisp;                /* 0: the loop as the reference writes it.  > 0 (the library's rule, include/lbmpm.h bulk_epsilon; its default
                                    2^-51): a colour whose density is below crisp * rho after the collision is absent -- it hands on exact zeros and
                                    the other colour takes f_tot.  Exists so that the HIP kernels can be compared with this file on K as well,
                                    which turns any last-bit difference into 1e-8 where |G| sits at its threshold (tests/test_rk3d_csf_gpu.py); that
                                    the rule itself moves the other fields by 1e-15 is a CPU test (tests/test_oracle_rk3d_csf.py) */
} rk3dcsf_sim;

static i64 wrap(i64 v, i64 n) { v %= n; return v < 0 ? v + n : v; }      /* any offset, any n >= 1 */
static i64 nbr(const rk3dcsf_sim *s, i64 x, i64 y, i64 z, int dx, int dy, int dz)
{
    return (wrap(z + dz, s->nz) * s->ny + wrap(y + dy, s->ny)) * s->nx + wrap(x + dx, s->nx);
}

/* weight of the 3-D E8 stencil (Sbragaglia et al. 2007) by |c|^2; 0 off the stencil */
static double e8w(int c2)
{
    switch (c2) {
    case 1: return 4. / 45.;
    case 2: return 1. / 21.;
    case 3: return 2. / 105.;
    case 4: return 5. / 504.;
    case 5: return 1. / 315.;
    case 6: return 1. / 630.;
    case 8: return 1. / 5040.;
    default: return 0.;
    }
}

/* D:657-690 (wetting solids), D:741-763 (fluid next to solid), D:768-892 (their solid normals) on the D3Q19 neighbourhood / the 3-D E8 set */
void rk3dcsf_setup(rk3dcsf_sim *s)
{
    i64 nx = s->nx, ny = s->ny, nz = s->nz;
    i64 W = 0;
    for (i64 z = 0; z < nz; ++z)
        for (i64 y = 0; y < ny; ++y)
            for (i64 x = 0; x < nx; ++x) {
                i64 n = (z * ny + y) * nx + x;
                int other = 0;
                for (int i = 1; i < Q; ++i)
                    if ((s->dom[nbr(s, x, y, z, CX[i], CY[i], CZ[i])] == 1) != (s->dom[n] == 1)) ++other;
                if (s->dom[n] == 1) s->kind[n] = other ? 3 : 1;
                else { s->kind[n] = other ? 2 : 0; if (other) ++W; }
                s->nsx[n] = s->nsy[n] = s->nsz[n] = 0.;
                if (s->kind[n] != 3) continue;
                double sx = 0., sy = 0., sz = 0.;
                for (int dz = -2; dz <= 2; ++dz)
                    for (int dy = -2; dy <= 2; ++dy)
                        for (int dx = -2; dx <= 2; ++dx) {
                            double w = e8w(dx * dx + dy * dy + dz * dz);
                            if (w == 0.) continue;
                            if (s->dom[nbr(s, x, y, z, dx, dy, dz)] == 1) continue;
                            sx += w * dx; sy += w * dy; sz += w * dz;
                        }
                double nrm = sqrt(sx * sx + sy * sy + sz * sz);
                s->nsx[n] = sx / nrm; s->nsy[n] = sy / nrm; s->nsz[n] = sz / nrm;
            }
    s->W = W;
}

static double sum19(const double *f) { double r = 0.; for (int i = 0; i < Q; ++i) r += f[i]; return r; }

/* A:170-176 calEquilibriumRK2D in three dimensions */
static double feq(double rho, int i, double vx, double vy, double vz)
{
    double eu = CX[i] * vx + CY[i] * vy + CZ[i] * vz;
    return rho * WT(i) * (1 + (3. * eu + 4.5 * eu * eu - 1.5 * (vx * vx + vy * vy + vz * vz)));
}

/* tau(phi): A:1815-1827 */
static double tau_of(const rk3dcsf_sim *s, double Phi, double rR, double rB)
{
    double tau = 1.;
    if (Phi > s->delta) tau = s->tauR;
    else if (Phi < -s->delta) tau = s->tauB;
    else if (fabs(Phi) <= s->delta) {
        if (s->tauType == 1) {
            tau = 0.5 + 1. / ((1. + Phi) / (2. * (s->tauR - 0.5)) + (1. - Phi) / (2. * (s->tauB - 0.5)));
        } else if (s->tauType == 2) {
            double ratioR = rR / (rR + rB), ratioB = rB / (rR + rB);
            double miuR = 3. / (s->tauR - 0.5), miuB = 3. / (s->tauB - 0.5);
            double miu = 1. / (ratioR * miuR + ratioB * miuB);
            tau = 3. * miu + 0.5;
        }
    }
    return tau;
}

/* transverse momentum corrections of the D3Q19 Zou-He closures (Hecht & Harting 2010); their 2-D image is the 1/2 (f_1 - f_3) of A:943-944 */
static void transverse(const double *f, double *Nx, double *Ny)
{
    *Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    *Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
}
static double sum_inplane(const double *f) { return f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10]; }
static const int UP[5] = {5, 11, 14, 15, 18};        /* e_z = +1 */
static const int DN[5] = {6, 12, 13, 16, 17};        /* e_z = -1, DN[k] = OPP[UP[k]] */

/* first half of the loop: boundary planes ... wetting-corrected colour gradient (D:1299-1424) */
void rk3dcsf_step_a(rk3dcsf_sim *s)
{
    i64 nx = s->nx, ny = s->ny, nz = s->nz, pl = nx * ny, N = pl * nz;
    /* inlet plane nz-2 and its ghost plane nz-1 */
    for (i64 k = 0; k < pl; ++k) {
        i64 n = (nz - 2) * pl + k, g = (nz - 1) * pl + k;
        if (s->dom[n] == 1) {
            if (s->inletType == 0) {                       /* A:2348-2412 constantTotalVelocityInlet */
                double *t = s->fT + Q * n, v = s->vzIn;
                double rho = (sum_inplane(t) + 2. * (t[5] + t[11] + t[14] + t[15] + t[18])) / (1. + v);
                for (int a = 0; a < 5; ++a) t[DN[a]] = feq(rho, DN[a], 0., 0., v) + (t[UP[a]] - feq(rho, UP[a], 0., 0., v));
                double ratioR = s->rhoR[n] / (s->rhoR[n] + s->rhoB[n]);
                s->rhoR[n] = ratioR * rho;
                for (int a = 0; a < 5; ++a) s->fR[Q * n + DN[a]] = ratioR * t[DN[a]];
                double ratioB = s->rhoB[n] / (s->rhoR[n] + s->rhoB[n]);      /* with the new rhoR: reference quirk */
                s->rhoB[n] = ratioB * rho;
                for (int a = 0; a < 5; ++a) s->fB[Q * n + DN[a]] = ratioB * t[DN[a]];
            } else {                                       /* A:925-962 calConstPressureInletGPU */
                for (int c = 0; c < 2; ++c) {
                    double *f = (c == 0 ? s->fB : s->fR) + Q * n, p = c == 0 ? s->pInB : s->pInR, Nx, Ny;
                    double v = -1. + (sum_inplane(f) + 2. * (f[5] + f[11] + f[14] + f[15] + f[18])) / p;
                    transverse(f, &Nx, &Ny);
                    f[6] = f[5] - 1. / 3. * p * v;
                    f[12] = f[11] + Nx - 1. / 6. * p * v;
                    f[13] = f[14] - Nx - 1. / 6. * p * v;
                    f[16] = f[15] + Ny - 1. / 6. * p * v;
                    f[17] = f[18] - Ny - 1. / 6. * p * v;
                    if (c == 0) s->rhoB[n] = p; else s->rhoR[n] = p;
                }
            }
        }
        if (s->dom[g] == 1) {
            memcpy(s->fR + Q * g, s->fR + Q * n, sizeof(double) * Q);
            memcpy(s->fB + Q * g, s->fB + Q * n, sizeof(double) * Q);
            if (s->inletType == 0) { s->rhoR[g] = sum19(s->fR + Q * g); s->rhoB[g] = sum19(s->fB + Q * g); }   /* A:607-650 */
            else { s->rhoR[g] = s->rhoR[n]; s->rhoB[g] = s->rhoB[n]; }                                            /* A:968-1002 */
        }
    }
    if (s->outletType == 1) {                              /* A:700-784: planes 2, 1, 0 one after the other */
        for (i64 z = 2; z >= 0; --z)
            for (i64 k = 0; k < pl; ++k) {
                i64 n = z * pl + k, up = (z + 1) * pl + k;
                if (s->dom[n] != 1) continue;
                memcpy(s->fR + Q * n, s->fR + Q * up, sizeof(double) * Q);
                memcpy(s->fB + Q * n, s->fB + Q * up, sizeof(double) * Q);
                s->rhoR[n] = sum19(s->fR + Q * n); s->rhoB[n] = sum19(s->fB + Q * n);
            }
    } else {
        for (i64 k = 0; k < pl; ++k) {
            i64 n = pl + k, g = k;
            if (s->dom[n] == 1) {                          /* A:2560-2590 calConstPressureLowerGPUTotal */
                double *t = s->fT + Q * n, pL = s->pOutTotal, Nx, Ny;
                double v = 1. - 1. / pL * (sum_inplane(t) + 2. * (t[6] + t[12] + t[13] + t[16] + t[17]));
                transverse(t, &Nx, &Ny);
                t[5] = t[6] + 1. / 3. * (pL * v);
                t[11] = t[12] - Nx + 1. / 6. * pL * v;
                t[14] = t[13] + Nx + 1. / 6. * pL * v;
                t[15] = t[16] - Ny + 1. / 6. * pL * v;
                t[18] = t[17] + Ny + 1. / 6. * pL * v;
                double ratioR = s->rhoR[n] / (s->rhoR[n] + s->rhoB[n]);
                for (int a = 0; a < 5; ++a) s->fR[Q * n + UP[a]] = ratioR * t[UP[a]];
                double ratioB = s->rhoB[n] / (s->rhoR[n] + s->rhoB[n]);
                for (int a = 0; a < 5; ++a) s->fB[Q * n + UP[a]] = ratioB * t[UP[a]];
            }
            if (s->dom[g] == 1) {                          /* A:1045-1081 */
                memcpy(s->fR + Q * g, s->fR + Q * n, sizeof(double) * Q);
                memcpy(s->fB + Q * g, s->fB + Q * n, sizeof(double) * Q);
                s->rhoR[g] = s->rhoR[n]; s->rhoB[g] = s->rhoB[n];
            }
        }
    }
    /* A:1414-1424 f_tot, A:2634-2654 velocity with half the force of the step before, A:1348-1357 phase field */
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        if (s->dom[n] != 1) continue;
        double *t = s->fT + Q * n;
        for (int i = 0; i < Q; ++i) t[i] = s->fR[Q * n + i] + s->fB[Q * n + i];
        double rs = s->rhoB[n] + s->rhoR[n];
        double mx = 0., my = 0., mz = 0.;
        for (int i = 1; i < Q; ++i) { mx += CX[i] * t[i]; my += CY[i] * t[i]; mz += CZ[i] * t[i]; }
        s->vx[n] = (mx + 0.5 * s->Fx[n]) / rs;
        s->vy[n] = (my + 0.5 * s->Fy[n]) / rs;
        s->vz[n] = (mz + 0.5 * s->Fz[n]) / rs;
        s->phi[n] = (s->rhoR[n] - s->rhoB[n]) / (s->rhoR[n] + s->rhoB[n]);
    }
    /* A:1560-1581 calColorValueOnSolid */
    if (s->W > 0) {
        PARFOR
        for (i64 z = 0; z < nz; ++z)
            for (i64 y = 0; y < ny; ++y)
                for (i64 x = 0; x < nx; ++x) {
                    i64 n = (z * ny + y) * nx + x;
                    if (s->kind[n] != 2) continue;
                    double sum = 0., sw = 0.;
                    for (int i = 1; i < Q; ++i) {
                        i64 q = nbr(s, x, y, z, CX[i], CY[i], CZ[i]);
                        if (s->dom[q] == 1) { sum += WT(i) * s->phi[q]; sw += WT(i); }
                    }
                    s->phi[n] = sum / sw;
                }
    }
    /* A:1584-1634 calRKInitialGradient; plain solids (no fluid neighbour) are never read, wetting solids carry phi_s */
    PARFOR
    for (i64 z = 0; z < nz; ++z)
        for (i64 y = 0; y < ny; ++y)
            for (i64 x = 0; x < nx; ++x) {
                i64 n = (z * ny + y) * nx + x;
                if (s->dom[n] != 1) continue;
                double gx = 0., gy = 0., gz = 0.;
                for (int i = 1; i < Q; ++i) {
                    double v = s->phi[nbr(s, x, y, z, CX[i], CY[i], CZ[i])];
                    gx += WT(i) * v * CX[i]; gy += WT(i) * v * CY[i]; gz += WT(i) * v * CZ[i];
                }
                s->Gx[n] = 3. * gx; s->Gy[n] = 3. * gy; s->Gz[n] = 3. * gz;
            }
    /* A:2430-2492 updateColorGradientOnWettingNew (Akai et al. 2018) */
    if (s->W > 0 && s->wetting == 2) {
        PARFOR
        for (i64 n = 0; n < N; ++n) {
            if (s->kind[n] != 3) continue;
            double gx = s->Gx[n], gy = s->Gy[n], gz = s->Gz[n];
            double nrm = sqrt(gx * gx + gy * gy + gz * gz);
            double ux = 0., uy = 0., uz = 0.;
            if (nrm > 1.0e-8) { ux = -gx / nrm; uy = -gy / nrm; uz = -gz / nrm; }
            double sx = s->nsx[n], sy = s->nsy[n], sz = s->nsz[n];
            double ang = ux * sx + uy * sy + uz * sz;
            double th = acos(ang);
            double c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
            if (fabs(sin(th)) > 1.0e-9) {
                c1 = s->sinT * cos(th) / sin(th);
                c2 = s->sinT / sin(th);
                c3 = -s->sinT * cos(th) / sin(th);
                c4 = -s->sinT / sin(th);
            }
            double ax = (s->cosT - c1) * sx + c2 * ux, ay = (s->cosT - c1) * sy + c2 * uy, az = (s->cosT - c1) * sz + c2 * uz;
            double bx = (s->cosT - c3) * sx + c4 * ux, by = (s->cosT - c3) * sy + c4 * uy, bz = (s->cosT - c3) * sz + c4 * uz;
            double d1 = sqrt((ax - ux) * (ax - ux) + (ay - uy) * (ay - uy) + (az - uz) * (az - uz));
            double d2 = sqrt((bx - ux) * (bx - ux) + (by - uy) * (by - uy) + (bz - uz) * (bz - uz));
            if (d1 < d2) { s->Gx[n] = -nrm * ax; s->Gy[n] = -nrm * ay; s->Gz[n] = -nrm * az; }
            else if (d1 > d2) { s->Gx[n] = -nrm * bx; s->Gy[n] = -nrm * by; s->Gz[n] = -nrm * bz; }
        }
    }
}

/* D3Q19 moment basis of d'Humieres et al. 2002 (as oracle/rk3d_oracle.c) */
static void mrt_basis(double M[Q][Q], double nrm[Q])
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
    for (int k = 0; k < Q; ++k) {
        double a = 0.;
        for (int i = 0; i < Q; ++i) a += M[k][i] * M[k][i];
        nrm[k] = a;
    }
}
/* the rows, for tests/test_oracle_rk3d.py (the paper's equilibrium moments as a known answer) */
void rk3dcsf_mrt_basis_public(double *M19x19) { double nrm[Q]; mrt_basis((double (*)[Q])M19x19, nrm); }
/* d <- M^-1 diag(S) M d (rows of M mutually orthogonal) */
static void mrt_apply(const double M[Q][Q], const double nrm[Q], const double S[Q], double d[Q])
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
static void mrt_rates(const rk3dcsf_sim *s, double inv_tau, double S[Q])
{
    const double *r = s->rates;
    const double v[Q] = {r[5], r[0], r[1], r[5], r[2], r[5], r[2], r[5], r[2], inv_tau, r[3], inv_tau, r[3], inv_tau, inv_tau, inv_tau, r[4], r[4], r[4]};
    for (int i = 0; i < Q; ++i) S[i] = v[i];
}

/* second half: CSF force ... streaming and densities (D:1425-1490) */
void rk3dcsf_step_b(rk3dcsf_sim *s)
{
    i64 nx = s->nx, ny = s->ny, nz = s->nz, N = nx * ny * nz;
    double M[Q][Q], nrm[Q];
    mrt_basis(M, nrm);
    /* A:2499-2551 calForceTermInColorGradientNew2D: n = -G / |G| (threshold 1e-8), derivatives of n over the fluid neighbours */
    PARFOR
    for (i64 z = 0; z < nz; ++z)
        for (i64 y = 0; y < ny; ++y)
            for (i64 x = 0; x < nx; ++x) {
                i64 n = (z * ny + y) * nx + x;
                if (s->dom[n] != 1) continue;
                double gn = sqrt(s->Gx[n] * s->Gx[n] + s->Gy[n] * s->Gy[n] + s->Gz[n] * s->Gz[n]);
                double ux = 0., uy = 0., uz = 0.;
                if (gn > 1.0e-8) { ux = -s->Gx[n] / gn; uy = -s->Gy[n] / gn; uz = -s->Gz[n] / gn; }
                double d[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};      /* d[a][b] = d_a n_b */
                for (int i = 1; i < Q; ++i) {
                    i64 q = nbr(s, x, y, z, CX[i], CY[i], CZ[i]);
                    if (s->dom[q] != 1) continue;
                    double qn = sqrt(s->Gx[q] * s->Gx[q] + s->Gy[q] * s->Gy[q] + s->Gz[q] * s->Gz[q]);
                    double q3[3] = {0., 0., 0.};
                    if (qn > 1.0e-8) { q3[0] = -s->Gx[q] / qn; q3[1] = -s->Gy[q] / qn; q3[2] = -s->Gz[q] / qn; }
                    const int e[3] = {CX[i], CY[i], CZ[i]};
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) d[a][b] += 3. * WT(i) * q3[b] * e[a];
                }
                double k = ux * uy * (d[1][0] + d[0][1]) + ux * uz * (d[2][0] + d[0][2]) + uy * uz * (d[2][1] + d[1][2])
                           - (uy * uy + uz * uz) * d[0][0] - (ux * ux + uz * uz) * d[1][1] - (ux * ux + uy * uy) * d[2][2];
                s->K[n] = k;
                s->Fx[n] = -0.5 * s->sigma * k * s->Gx[n];
                s->Fy[n] = -0.5 * s->sigma * k * s->Gy[n];
                s->Fz[n] = -0.5 * s->sigma * k * s->Gz[n];
            }
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        if (s->dom[n] != 1) continue;
        double *t = s->fT + Q * n;
        double rR = s->rhoR[n], rB = s->rhoB[n], vx = s->vx[n], vy = s->vy[n], vz = s->vz[n];
        double fx = s->Fx[n], fy = s->Fy[n], fz = s->Fz[n];
        double tau = tau_of(s, s->phi[n], rR, rB);
        if (!s->mrt) {
            for (int i = 0; i < Q; ++i) {                    /* A:1804-1848 */
                double eT = feq(rR, i, vx, vy, vz) + feq(rB, i, vx, vy, vz);
                t[i] = -1. / tau * (t[i] - eT) + t[i];
            }
            for (int i = 0; i < Q; ++i) {                    /* A:1743-1798 */
                double eu = CX[i] * vx + CY[i] * vy + CZ[i] * vz;
                double src = WT(i) * ((3. * (CX[i] - vx) + 9. * CX[i] * eu) * fx + (3. * (CY[i] - vy) + 9. * CY[i] * eu) * fy +
                                      (3. * (CZ[i] - vz) + 9. * CZ[i] * eu) * fz) * (1. - 1. / (2. * tau));
                t[i] = t[i] + src;
            }
        } else {
            double S[Q], d[Q];
            mrt_rates(s, 1. / tau, S);
            for (int i = 0; i < Q; ++i) d[i] = t[i] - (feq(rR, i, vx, vy, vz) + feq(rB, i, vx, vy, vz));
            mrt_apply(M, nrm, S, d);                         /* A:1938-2017 */
            for (int i = 0; i < Q; ++i) t[i] = -d[i] + t[i];
            for (int i = 0; i < Q; ++i) {                    /* A:2027-2113: src_i = w_i (3 e.F + 9 (e e - I/3) : u F) */
                double ef = CX[i] * fx + CY[i] * fy + CZ[i] * fz, eu = CX[i] * vx + CY[i] * vy + CZ[i] * vz;
                double uf = vx * fx + vy * fy + vz * fz;
                d[i] = WT(i) * (3. * ef + 9. * eu * ef - 3. * uf);
                S[i] = 1. - 0.5 * S[i];
            }
            mrt_apply(M, nrm, S, d);
            for (int i = 0; i < Q; ++i) t[i] = t[i] + d[i];
        }
        /* A:1857-1899 calRecoloringProcessM */
        double gx = s->Gx[n], gy = s->Gy[n], gz = s->Gz[n], gn = sqrt(gx * gx + gy * gy + gz * gz), tot = rR + rB;
        for (int i = 0; i < Q; ++i) {
            double un = sqrt((double)(CX[i] * CX[i] + CY[i] * CY[i] + CZ[i] * CZ[i])), c;
            if (gn > 1.0e-8 && un > 1.0e-8) c = (CX[i] * gx + CY[i] * gy + CZ[i] * gz) / (un * gn);
            else c = 0.;
            s->fR[Q * n + i] = rR / tot * t[i] + s->beta * rR * rB / tot * WT(i) * c * un;
            s->fB[Q * n + i] = rB / tot * t[i] - s->beta * rR * rB / tot * WT(i) * c * un;
            if (s->crisp > 0.) {
                const int noB = fabs(rB) <= s->crisp * tot, noR = !noB && fabs(rR) <= s->crisp * tot;
                if (noB) { s->fR[Q * n + i] = t[i]; s->fB[Q * n + i] = 0.; }
                else if (noR) { s->fR[Q * n + i] = 0.; s->fB[Q * n + i] = t[i]; }
            }
        }
    }
    /* A:340-417 streaming: push + half-way bounce-back; direction 0 stays where it is */
    PARFOR
    for (i64 z = 0; z < nz; ++z)
        for (i64 y = 0; y < ny; ++y)
            for (i64 x = 0; x < nx; ++x) {
                i64 n = (z * ny + y) * nx + x;
                if (s->dom[n] != 1) continue;
                /* pull form of the same map: from the cell the population came from, or its own opposite one off a solid */
                for (int i = 0; i < Q; ++i) {
                    i64 q = nbr(s, x, y, z, -CX[i], -CY[i], -CZ[i]);
                    if (i == 0 || s->dom[q] == 1) { s->gR[Q * n + i] = s->fR[Q * q + i]; s->gB[Q * n + i] = s->fB[Q * q + i]; }
                    else { s->gR[Q * n + i] = s->fR[Q * n + OPP[i]]; s->gB[Q * n + i] = s->fB[Q * n + OPP[i]]; }
                }
            }
    double *tp = s->fR; s->fR = s->gR; s->gR = tp;
    tp = s->fB; s->fB = s->gB; s->gB = tp;
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        if (s->dom[n] != 1) continue;
        for (int i = 0; i < Q; ++i) s->fT[Q * n + i] = s->fR[Q * n + i] + s->fB[Q * n + i];
        s->rhoR[n] = sum19(s->fR + Q * n); s->rhoB[n] = sum19(s->fB + Q * n);
    }
}

void rk3dcsf_run(rk3dcsf_sim *s, i64 nsteps)
{
    for (i64 k = 0; k < nsteps; ++k) { rk3dcsf_step_a(s); rk3dcsf_step_b(s); }
}

/* D:577-601 in three dimensions; the loop's arrays as they stand at its top */
void rk3dcsf_init(rk3dcsf_sim *s, const double *rhoR0, const double *rhoB0, const double *vx0, const double *vy0, const double *vz0)
{
    i64 N = s->nx * s->ny * s->nz;
    for (i64 n = 0; n < N; ++n) {
        double vx = vx0 ? vx0[n] : 0., vy = vy0 ? vy0[n] : 0., vz = vz0 ? vz0[n] : 0.;
        for (int i = 0; i < Q; ++i) {
            s->fR[Q * n + i] = s->dom[n] == 1 ? feq(rhoR0[n], i, vx, vy, vz) : 0.;
            s->fB[Q * n + i] = s->dom[n] == 1 ? feq(rhoB0[n], i, vx, vy, vz) : 0.;
            s->gR[Q * n + i] = 0.; s->gB[Q * n + i] = 0.;
            s->fT[Q * n + i] = s->fR[Q * n + i] + s->fB[Q * n + i];
        }
        s->rhoR[n] = s->dom[n] == 1 ? sum19(s->fR + Q * n) : 0.;
        s->rhoB[n] = s->dom[n] == 1 ? sum19(s->fB + Q * n) : 0.;
        s->Fx[n] = s->Fy[n] = s->Fz[n] = s->K[n] = 0.;
        s->vx[n] = s->vy[n] = s->vz[n] = s->phi[n] = s->Gx[n] = s->Gy[n] = s->Gz[n] = 0.;
    }
}

/* restart: the loop's arrays at its top from given streamed populations (and the force of the step before) */
void rk3dcsf_set_populations(rk3dcsf_sim *s, const double *fR, const double *fB)
{
    i64 N = s->nx * s->ny * s->nz;
    for (i64 n = 0; n < N; ++n) {
        for (int i = 0; i < Q; ++i) {
            s->fR[Q * n + i] = s->dom[n] == 1 ? fR[Q * n + i] : 0.;
            s->fB[Q * n + i] = s->dom[n] == 1 ? fB[Q * n + i] : 0.;
            s->fT[Q * n + i] = s->fR[Q * n + i] + s->fB[Q * n + i];
        }
        s->rhoR[n] = s->dom[n] == 1 ? sum19(s->fR + Q * n) : 0.;
        s->rhoB[n] = s->dom[n] == 1 ? sum19(s->fB + Q * n) : 0.;
    }
}
