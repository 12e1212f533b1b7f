/*
 * oracle/rk_oracle.c -- CPU restatement of the reference's colour-gradient D2Q9 path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a plain-C restatement of
 * the algorithm of PorousMediaSimulation/openLBMPM's RKCG2D GPU path, written from
 * reading the reference (no reference source is copied; arithmetic is restated in the
 * reference's evaluation order so that results agree with the reference kernels to
 * round-off).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product path (openlbmpm_amd/, liblbmpm_hip.so) never does.
 *
 * Parity status: PINNED.  Checked against golden vectors captured by running the real
 * reference driver RKColorGradientLBM.runRKColorGradient2DCSF under a numba stand-in
 * (tests/golden/gen/make_golden_rk.py -> tests/golden/rk_*.npz; tests/test_oracle_rk.py).
 *
 * Data layout = the reference's: sparse, fluid nodes only, AoS f[N][9] float64,
 * int64 neighbour table nbr[8N] in the order E,N,W,S,NE,NW,SW,SE, value >=0 compact
 * fluid index, -1 plain solid, <=-2 wetting-solid id (-id-2 indexes phiSolid).
 * Citations "A:" = RKCG2D/AcceleratedRKGPU2D.py, "D:" = RKCG2D/RKD2Q9.py (line numbers
 * of /root/reference at the surveyed revision).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off [-fopenmp]).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* lattice constants, D:300-303 */
static const double EX[9] = {0., 1., 0., -1., 0., 1., -1., -1., 1.};
static const double EY[9] = {0., 0., 1., 0., -1., 1., 1., -1., -1.};
static const double WT[9] = {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9.,
                             1. / 36., 1. / 36., 1. / 36., 1. / 36.};

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

/* ------------------------------------------------------------------ set-up (host side) */

/* D:657-690 optimizeFluidandSolidArray: compaction of fluid nodes (row-major scan),
 * wetting-solid ids -2,-3,... for solid nodes with >=1 fluid node in their 3x3
 * (upper overflow wraps to 0, lower underflow wraps like Python's negative index).
 * Returns W (number of wetting solids); caller sized the outputs for ny*nx. */
i64 rk_compact(i64 nx, i64 ny, const uint8_t *isDomain, i64 *fluidNodes, i64 *newIndex,
               i64 *wettingSolidNodes, i64 *outN)
{
    i64 n = 0, w = 0, id = -2;
    for (i64 i = 0; i < ny; ++i)
        for (i64 j = 0; j < nx; ++j) {
            newIndex[i * nx + j] = -1;
        }
    for (i64 i = 0; i < ny; ++i)
        for (i64 j = 0; j < nx; ++j) {
            if (isDomain[i * nx + j] == 1) {
                fluidNodes[n] = i * nx + j;
                newIndex[i * nx + j] = n;
                ++n;
            } else {
                int cnt = 0;
                for (int m = -1; m < 2; ++m)
                    for (int q = -1; q < 2; ++q) {
                        i64 ty = (i + m < ny) ? i + m : 0;
                        i64 tx = (j + q < nx) ? j + q : 0;
                        if (ty < 0) ty += ny;   /* Python negative index */
                        if (tx < 0) tx += nx;
                        if (isDomain[ty * nx + tx] == 1) ++cnt;
                    }
                if (cnt > 0) {
                    wettingSolidNodes[w++] = i * nx + j;
                    newIndex[i * nx + j] = id--;
                }
            }
        }
    *outN = n;
    return w;
}

/* A:15-53 fillNeighboringNodes / A:58-95 fillNeighboringWettingNodes (same body):
 * periodic wrap on all four edges. */
void rk_fill_neighbors(i64 total, i64 nx, i64 ny, const i64 *nodes, const i64 *newIndex,
                       i64 *nbr)
{
    PARFOR
    for (i64 k = 0; k < total; ++k) {
        i64 loc = nodes[k];
        i64 i = loc / nx, j = loc % nx;
        i64 F = (j < nx - 1) ? j + 1 : 0;
        i64 B = (j > 0) ? j - 1 : nx - 1;
        i64 U = (i < ny - 1) ? i + 1 : 0;
        i64 L = (i > 0) ? i - 1 : ny - 1;
        i64 *o = nbr + 8 * k;
        o[0] = newIndex[i * nx + F];
        o[1] = newIndex[U * nx + j];
        o[2] = newIndex[i * nx + B];
        o[3] = newIndex[L * nx + j];
        o[4] = newIndex[U * nx + F];
        o[5] = newIndex[U * nx + B];
        o[6] = newIndex[L * nx + B];
        o[7] = newIndex[L * nx + F];
    }
}

/* D:741-763 sortOutFluidNodesToSolid: fluid nodes with >=1 solid in their 3x3. */
i64 rk_fluid_near_solid(i64 nx, i64 ny, const uint8_t *isDomain, const i64 *newIndex,
                        i64 *nodesGPU, i64 *nodesOriginal)
{
    i64 c = 0;
    for (i64 i = 0; i < ny; ++i)
        for (i64 j = 0; j < nx; ++j) {
            if (isDomain[i * nx + j] != 1) continue;
            int cnt = 0;
            for (int m = -1; m < 2; ++m)
                for (int q = -1; q < 2; ++q) {
                    i64 ty = (i + m < ny) ? i + m : 0;
                    i64 tx = (j + q < nx) ? j + q : 0;
                    if (ty < 0) ty += ny;
                    if (tx < 0) tx += nx;
                    if (isDomain[ty * nx + tx] == 0) ++cnt;
                }
            if (cnt > 0) {
                nodesGPU[c] = newIndex[i * nx + j];
                nodesOriginal[c] = i * nx + j;
                ++c;
            }
        }
    return c;
}

/* D:768-892 calVectorNormaltoSolid: 24-point iso-8 weighted sum of lattice vectors
 * pointing at solid nodes; accumulation order as in the reference. */
void rk_solid_normals(i64 nx, i64 ny, const uint8_t *isDomain, i64 count,
                      const i64 *nodesOriginal, double *nsX, double *nsY)
{
    for (i64 k = 0; k < count; ++k) {
        i64 loc = nodesOriginal[k];
        i64 y = loc / nx, x = loc % nx;
        i64 E1 = (x < nx - 1) ? x + 1 : 0;
        i64 W1 = (x > 0) ? x - 1 : nx - 1;
        i64 N1 = (y < ny - 1) ? y + 1 : 0;
        i64 S1 = (y > 0) ? y - 1 : ny - 1;
        i64 E2 = (x < nx - 2) ? x + 2 : (x == nx - 2 ? 0 : 1);
        i64 W2 = (x > 1) ? x - 2 : (x == 1 ? nx - 1 : nx - 2);
        i64 N2 = (y < ny - 2) ? y + 2 : (y == ny - 2 ? 0 : 1);
        i64 S2 = (y > 1) ? y - 2 : (y == 1 ? ny - 1 : ny - 2);
        double sx = 0., sy = 0.;
#define SOLID(yy, xx) (isDomain[(yy) * nx + (xx)] == 0)
#define ACC(w, cx, cy) do { sx += (w) * 1. * (cx); sy += (w) * 1. * (cy); } while (0)
        if (SOLID(y, E1)) ACC(4. / 21., 1., 0.);
        if (SOLID(N1, x)) ACC(4. / 21., 0., 1.);
        if (SOLID(y, W1)) ACC(4. / 21., -1., 0.);
        if (SOLID(S1, x)) ACC(4. / 21., 0., -1.);
        if (SOLID(N1, E1)) ACC(4. / 45., 1., 1.);
        if (SOLID(N1, W1)) ACC(4. / 45., -1., 1.);
        if (SOLID(S1, W1)) ACC(4. / 45., -1., -1.);
        if (SOLID(S1, E1)) ACC(4. / 45., 1., -1.);
        if (SOLID(y, E2)) ACC(1. / 60., 2., 0.);
        if (SOLID(N2, x)) ACC(1. / 60., 0., 2.);
        if (SOLID(y, W2)) ACC(1. / 60., -2., 0.);
        if (SOLID(S2, x)) ACC(1. / 60., 0., -2.);
        if (SOLID(N1, E2)) ACC(2. / 315., 2., 1.);
        if (SOLID(N2, E1)) ACC(2. / 315., 1., 2.);
        if (SOLID(N2, W1)) ACC(2. / 315., -1., 2.);
        if (SOLID(N1, W2)) ACC(2. / 315., -2., 1.);
        if (SOLID(S1, W2)) ACC(2. / 315., -2., -1.);
        if (SOLID(S2, W1)) ACC(2. / 315., -1., -2.);
        if (SOLID(S2, E1)) ACC(2. / 315., 1., -2.);
        if (SOLID(S1, E2)) ACC(2. / 315., 2., -1.);
        if (SOLID(N2, E2)) ACC(1. / 5040., 2., 2.);
        if (SOLID(N2, W2)) ACC(1. / 5040., -2., 2.);
        if (SOLID(S2, W2)) ACC(1. / 5040., -2., -2.);
        if (SOLID(S2, E2)) ACC(1. / 5040., 2., -2.);
#undef SOLID
#undef ACC
        double nrm = sqrt(sx * sx + sy * sy);
        nsX[k] = sx / nrm;
        nsY[k] = sy / nrm;
    }
}

/* D:577-601 __initializeFluidPDF: f_i = rho w_i (1 + (3eu + 4.5(eu)^2 - 1.5u^2)) */
void rk_init_pdf(double rhoR, double rhoB, double vx, double vy, double *fR, double *fB)
{
    for (int i = 0; i < 9; ++i) {
        double eu = EX[i] * vx + EY[i] * vy;
        double t = 1 + (3. * eu + 4.5 * eu * eu - 1.5 * (vx * vx + vy * vy));
        fR[i] = rhoR * WT[i] * t;
        fB[i] = rhoB * WT[i] * t;
    }
}

/* ------------------------------------------------------------------ per-step kernels */

/* A:103-120 calMacroDensityRKGPU2D */
void rk_macro_density(i64 N, const double *fR, const double *fB, double *rhoR, double *rhoB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double r = 0., b = 0.;
        for (int i = 0; i < 9; ++i) { r += fR[9 * n + i]; b += fB[9 * n + i]; }
        rhoR[n] = r; rhoB[n] = b;
    }
}

/* A:1414-1424 calTotalFluidPDF */
void rk_total_pdf(i64 N, const double *fR, const double *fB, double *fT)
{
    PARFOR
    for (i64 k = 0; k < 9 * N; ++k) fT[k] = fR[k] + fB[k];
}

/* A:2634-2654 calPhysicalVelocityRKGPU2DNew1: u = (sum e f_tot + F/2)/(rhoR+rhoB) */
void rk_velocity(i64 N, const double *fT, const double *rhoR, const double *rhoB,
                 double *vx, double *vy, const double *Fx, const double *Fy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        const double *f = fT + 9 * n;
        double rs = rhoB[n] + rhoR[n];
        double tx = f[1] - f[3] + f[5] - f[6] - f[7] + f[8] + 0.5 * Fx[n];
        vx[n] = tx / rs;
        double ty = f[2] - f[4] + f[5] + f[6] - f[7] - f[8] + 0.5 * Fy[n];
        vy[n] = ty / rs;
    }
}

/* A:1348-1357 calPhaseFieldPhi */
void rk_phase_field(i64 N, const double *rhoR, const double *rhoB, double *phi)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) phi[n] = (rhoR[n] - rhoB[n]) / (rhoR[n] + rhoB[n]);
}

/* A:1560-1581 calColorValueOnSolid: weighted mean of phi over fluid neighbours */
void rk_color_on_solid(i64 W, const i64 *nbrWet, const double *phi, double *phiS)
{
    PARFOR
    for (i64 s = 0; s < W; ++s) {
        double sum = 0., sw = 0.;
        for (int i = 0; i < 8; ++i) {
            i64 loc = nbrWet[8 * s + i];
            if (loc >= 0) { sum += WT[i + 1] * phi[loc]; sw += WT[i + 1]; }
        }
        phiS[s] = sum / sw;
    }
}

/* A:1584-1634 calRKInitialGradient: G = 3 sum_i w_i e_i phi(x+e_i) */
void rk_gradient(i64 N, const i64 *nbr, const double *phi, const double *phiS,
                 double *Gx, double *Gy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double gx = 0., gy = 0.;
        for (int i = 0; i < 8; ++i) {
            i64 q = nbr[8 * n + i];
            double v = (q >= 0) ? phi[q] : phiS[-q - 2];
            gx += WT[i + 1] * v * EX[i + 1];
            gy += WT[i + 1] * v * EY[i + 1];
        }
        Gx[n] = 3. * gx; Gy[n] = 3. * gy;
    }
}

/* A:1639-1679 updateColorGradientOnWetting (WettingType 1, Xu 2017) */
void rk_wetting1(i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx,
                 const double *nsy, double *Gx, double *Gy)
{
    PARFOR
    for (i64 k = 0; k < Wf; ++k) {
        double n1x = nsx[k] * cosT - nsy[k] * sinT;
        double n1y = nsy[k] * cosT + nsx[k] * sinT;
        double n2x = nsx[k] * cosT + nsy[k] * sinT;
        double n2y = nsy[k] * cosT - nsx[k] * sinT;
        i64 loc = fluidWet[k];
        double nrm = sqrt(Gx[loc] * Gx[loc] + Gy[loc] * Gy[loc]);
        double ux, uy;
        if (nrm > 1.0e-8) { ux = Gx[loc] / nrm; uy = Gy[loc] / nrm; }
        else { ux = 0.; uy = 0.; }
        double dx1 = ux - n1x, dy1 = uy - n1y, dx2 = ux - n2x, dy2 = uy - n2y;
        double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
        double mx = 0., my = 0.;
        if (d1 < d2) { mx = n1x; my = n1y; }
        else if (d1 > d2) { mx = n2x; my = n2y; }
        else if (d1 == d2) { mx = nsx[k]; my = nsy[k]; }
        Gx[loc] = nrm * mx; Gy[loc] = nrm * my;
    }
}

/* A:2430-2492 updateColorGradientOnWettingNew (WettingType 2, Akai 2018) */
void rk_wetting2(i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx,
                 const double *nsy, double *Gx, double *Gy)
{
    PARFOR
    for (i64 k = 0; k < Wf; ++k) {
        i64 loc = fluidWet[k];
        double nrm = sqrt(Gx[loc] * Gx[loc] + Gy[loc] * Gy[loc]);
        double ux, uy;
        if (nrm > 1.0e-8) { ux = -Gx[loc] / nrm; uy = -Gy[loc] / nrm; }
        else { ux = 0.; uy = 0.; }
        double ang = ux * nsx[k] + uy * nsy[k];
        double th = acos(ang);
        double c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
        if (fabs(sin(th)) > 1.0e-9) {
            c1 = sinT * cos(th) / sin(th);
            c2 = sinT / sin(th);
            c3 = -sinT * cos(th) / sin(th);
            c4 = -sinT / sin(th);
        }
        double nx1 = (cosT - c1) * nsx[k] + c2 * ux;
        double ny1 = (cosT - c1) * nsy[k] + c2 * uy;
        double nx2 = (cosT - c3) * nsx[k] + c4 * ux;
        double ny2 = (cosT - c3) * nsy[k] + c4 * uy;
        double dx1 = nx1 - ux, dy1 = ny1 - uy, dx2 = nx2 - ux, dy2 = ny2 - uy;
        double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
        if (d1 < d2) { Gx[loc] = -nrm * nx1; Gy[loc] = -nrm * ny1; }
        else if (d1 > d2) { Gx[loc] = -nrm * nx2; Gy[loc] = -nrm * ny2; }
        /* d1 == d2 (or NaN): gradient left untouched, as in the reference */
    }
}

/* A:1686-1736 (type 1: n=+G/|G|, thresholds ==0, F=+0.5 sigma K G) and
 * A:2499-2551 (type 2: n=-G/|G|, threshold 1e-8, F=-0.5 sigma K G). */
void rk_force(i64 N, int type, double sigma, const i64 *nbr, const double *Gx,
              const double *Gy, double *Fx, double *Fy, double *K)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double nrm = sqrt(Gx[n] * Gx[n] + Gy[n] * Gy[n]);
        double ux = 0., uy = 0.;
        if (type == 2) {
            if (nrm > 1.0e-8) { ux = -Gx[n] / nrm; uy = -Gy[n] / nrm; }
        } else {
            if (nrm > 0.) { ux = Gx[n] / nrm; uy = Gy[n] / nrm; }
        }
        double pyx = 0., pxy = 0., px = 0., py = 0.;
        for (int i = 0; i < 8; ++i) {
            i64 q = nbr[8 * n + i];
            if (q < 0) continue;
            double nn = sqrt(Gx[q] * Gx[q] + Gy[q] * Gy[q]);
            double qx = 0., qy = 0.;
            if (type == 2) {
                if (nn > 1.0e-8) { qx = -Gx[q] / nn; qy = -Gy[q] / nn; }
            } else {
                if (nn > 0.) { qx = Gx[q] / nn; qy = Gy[q] / nn; }
            }
            pyx += 3. * WT[i + 1] * qy * EX[i + 1];
            pxy += 3. * WT[i + 1] * qx * EY[i + 1];
            px += 3. * WT[i + 1] * qx * EX[i + 1];
            py += 3. * WT[i + 1] * qy * EY[i + 1];
        }
        double k = ux * uy * (pyx + pxy) - uy * uy * px - ux * ux * py;
        K[n] = k;
        if (type == 2) {
            Fx[n] = -0.5 * sigma * k * Gx[n];
            Fy[n] = -0.5 * sigma * k * Gy[n];
        } else {
            Fx[n] = 0.5 * sigma * k * Gx[n];
            Fy[n] = 0.5 * sigma * k * Gy[n];
        }
    }
}

/* tau(phi): A:1815-1827 == A:1967-1981 == A:2052-2066 == A:1755-1767 */
static double rk_tau(int option, double tauR, double tauB, double delta, double Phi,
                     double rR, double rB)
{
    double tau = 1.;
    if (Phi > delta) tau = tauR;
    else if (Phi < -delta) tau = tauB;
    else if (fabs(Phi) <= delta) {
        if (option == 1) {
            tau = 0.5 + 1. / ((1. + Phi) / (2. * (tauR - 0.5)) + (1. - Phi) / (2. * (tauB - 0.5)));
        } else if (option == 2) {
            double ratioR = rR / (rR + rB);
            double ratioB = rB / (rR + rB);
            double miuR = 3. / (tauR - 0.5), miuB = 3. / (tauB - 0.5);
            double miu = 1. / (ratioR * miuR + ratioB * miuB);
            tau = 3. * miu + 0.5;
        }
    }
    return tau;
}

/* A:170-176 calEquilibriumRK2D */
static inline double rk_feq(double rho, double w, double ex, double ey, double vx, double vy)
{
    return rho * w * (1 + (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) *
                           (ex * vx + ey * vy) - 1.5 * (vx * vx + vy * vy)));
}

/* A:1804-1848 calRKCollision1TotalGPU2DSRTM */
void rk_collide_srt(i64 N, int option, double tauR, double tauB, double delta,
                    const double *vx, const double *vy, const double *rhoR,
                    const double *rhoB, const double *phi, double *fT)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        for (int i = 0; i < 9; ++i) {
            double eR = rk_feq(rhoR[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            double eB = rk_feq(rhoB[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            double eT = eR + eB;
            fT[9 * n + i] = -1. / tau * (fT[9 * n + i] - eT) + fT[9 * n + i];
        }
    }
}

/* A:1743-1798 calPerturbationFromForce2D (Guo source, SRT) */
void rk_force_srt(i64 N, int option, double tauR, double tauB, double delta,
                  const double *vx, const double *vy, const double *Fx, const double *Fy,
                  const double *phi, double *fT, const double *rhoR, const double *rhoB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        double fx = Fx[n], fy = Fy[n];
        for (int i = 0; i < 9; ++i) {
            double src = WT[i] * ((3. * (EX[i] - vx[n]) + 9. * EX[i] * (EX[i] * vx[n] + EY[i] * vy[n])) * fx +
                                  (3. * (EY[i] - vy[n]) + 9. * EY[i] * (EX[i] * vx[n] + EY[i] * vy[n])) * fy) *
                         (1. - 1. / (2. * tau));
            fT[9 * n + i] = fT[9 * n + i] + src;
        }
    }
}

/* A:1938-2017 calRKCollision1TotalGPU2DMRTM.  S = collisionS with S[7]=S[8]=1/tau
 * evaluated PER NODE (the reference's shared-memory scratch is racy when tau varies in
 * a block; per-node is the evident intent and what the sequential emulation yields). */
void rk_collide_mrt(i64 N, int option, double tauR, double tauB, double delta,
                    const double *vx, const double *vy, const double *rhoR,
                    const double *rhoB, const double *phi, double *fT, const double *M,
                    const double *Minv, const double *S)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double s[9], d[9], m[9];
        for (int i = 0; i < 9; ++i) s[i] = S[i];
        double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        s[7] = 1. / tau; s[8] = 1. / tau;
        for (int i = 0; i < 9; ++i) {
            double eR = rk_feq(rhoR[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            double eB = rk_feq(rhoB[n], WT[i], EX[i], EY[i], vx[n], vy[n]);
            d[i] = fT[9 * n + i] - (eR + eB);
        }
        for (int i = 0; i < 9; ++i) {
            double t = 0.;
            for (int j = 0; j < 9; ++j) t += M[9 * i + j] * d[j];
            m[i] = t;
        }
        for (int i = 0; i < 9; ++i) m[i] = m[i] * s[i];
        for (int i = 0; i < 9; ++i) {
            double t = 0.;
            for (int j = 0; j < 9; ++j) t += Minv[9 * i + j] * m[j];
            fT[9 * n + i] = -t + fT[9 * n + i];
        }
    }
}

/* A:2027-2113 calPerturbationFromForce2DMRT: f += Minv (I - S/2) M src */
void rk_force_mrt(i64 N, int option, double tauR, double tauB, double delta,
                  const double *vx, const double *vy, const double *Fx, const double *Fy,
                  const double *phi, double *fT, const double *M, const double *Minv,
                  const double *S, const double *rhoR, const double *rhoB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double s[9], src[9], m[9];
        for (int i = 0; i < 9; ++i) s[i] = 1. - 0.5 * S[i];
        double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        s[7] = 1. - 0.5 * 1. / tau; s[8] = 1. - 0.5 * 1. / tau;
        for (int i = 0; i < 9; ++i) {
            double t1 = EX[i] * Fx[n] * 3.;
            double t2 = EY[i] * Fy[n] * 3.;
            double t3 = (EX[i] * EX[i] - 1. / 3.) * vx[n] * Fx[n] * 9.;
            double t4 = EX[i] * EY[i] * vy[n] * Fx[n] * 9.;
            double t5 = EY[i] * EX[i] * vx[n] * Fy[n] * 9.;
            double t6 = (EY[i] * EY[i] - 1. / 3.) * vy[n] * Fy[n] * 9.;
            src[i] = WT[i] * (t1 + t2 + t3 + t4 + t5 + t6);
        }
        for (int i = 0; i < 9; ++i) {
            double t = 0.;
            for (int j = 0; j < 9; ++j) t += M[9 * i + j] * src[j];
            m[i] = t;
        }
        for (int i = 0; i < 9; ++i) m[i] = s[i] * m[i];
        for (int i = 0; i < 9; ++i) {
            double t = 0.;
            for (int j = 0; j < 9; ++j) t += Minv[9 * i + j] * m[j];
            fT[9 * n + i] = fT[9 * n + i] + t;
        }
    }
}

/* A:1857-1899 calRecoloringProcessM */
void rk_recolor(i64 N, double beta, const double *rhoR, const double *rhoB,
                const double *Gx, const double *Gy, double *fR, double *fB, const double *fT)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double gn = sqrt(Gx[n] * Gx[n] + Gy[n] * Gy[n]);
        double tot = rhoR[n] + rhoB[n];
        for (int i = 0; i < 9; ++i) {
            double un = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            double c;
            if (gn > 1.0e-8 && un > 1.0e-8) c = (EX[i] * Gx[n] + EY[i] * Gy[n]) / (un * gn);
            else c = 0.;
            double t = fT[9 * n + i];
            fR[9 * n + i] = rhoR[n] / tot * t + beta * rhoR[n] * rhoB[n] / tot * WT[i] * c * un;
            fB[9 * n + i] = rhoB[n] / tot * t - beta * rhoR[n] * rhoB[n] / tot * WT[i] * c * un;
        }
    }
}

/* A:340-403 calStreaming1GPU: push, nbr<0 -> half-way bounce-back into opposite slot */
static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
void rk_stream1(i64 N, const i64 *nbr, const double *f, double *fNew)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int i = 1; i < 9; ++i) {
            i64 q = nbr[8 * n + i - 1];
            if (q >= 0) fNew[9 * q + i] = f[9 * n + i];
            else fNew[9 * n + OPP[i]] = f[9 * n + i];
        }
}

/* A:409-417 calStreaming2GPU: copy back directions 1..8 (0 is never copied) */
void rk_stream2(i64 N, const double *fNew, double *f)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int j = 1; j < 9; ++j) f[9 * n + j] = fNew[9 * n + j];
}

/* ------------------------------------------------------------------ boundary rows */

/* A:2348-2412 constantTotalVelocityInlet (row ny-2): non-equilibrium bounce-back on
 * f_tot, split by colour ratio; ratioB is evaluated AFTER rhoR was overwritten
 * (reference quirk, replicated). */
void rk_inlet_velocity_total(i64 N, i64 nx, i64 ny, double vyIn, const i64 *fluidNodes,
                             double *rhoR, double *rhoB, double *fR, double *fB,
                             double *fT, double *vy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (ny - 1) * nx && loc >= (ny - 2) * nx)) continue;
        double *t = fT + 9 * n;
        double v = vyIn;
        double rho = (t[0] + t[1] + t[3] + 2. * (t[2] + t[5] + t[6])) / (1. + v);
        double eq2 = rho * 1. / 9. * (1. + 3. * (1. * v) + 4.5 * (0. + 1. * v) * (0. + 1. * v) - 1.5 * (v * v));
        double eq4 = rho * 1. / 9. * (1. + 3. * (-1. * v) + 4.5 * (0. + (-1.) * v) * (0. + (-1.) * v) - 1.5 * (v * v));
        t[4] = eq4 + (t[2] - eq2);
        double eq5 = rho * 1. / 36. * (1. + 3. * (1. * v + 1. * 0.) + 4.5 * (1. * v + 1. * 0.) * (1. * v + 1. * 0.) - 1.5 * (v * v));
        double eq7 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (-1.) * 0.) + 4.5 * ((-1.) * v + (-1.) * 0.) * ((-1.) * v + (-1.) * 0.) - 1.5 * (v * v));
        t[7] = eq7 + (t[5] - eq5);
        double eq6 = rho * 1. / 36. * (1. + 3. * ((1.) * v + (-1.) * 0.) + 4.5 * ((1.) * v + (-1.) * 0.) * (1. * v + (-1.) * 0.) - 1.5 * (v * v));
        double eq8 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (1.) * 0.) + 4.5 * ((-1.) * v + 1. * 0.) * ((-1.) * v + 1. * 0.) - 1.5 * (v * v));
        t[8] = eq8 + (t[6] - eq6);
        double ratioR = rhoR[n] / (rhoR[n] + rhoB[n]);
        rhoR[n] = ratioR * rho;
        fR[9 * n + 4] = ratioR * t[4]; fR[9 * n + 7] = ratioR * t[7]; fR[9 * n + 8] = ratioR * t[8];
        double ratioB = rhoB[n] / (rhoR[n] + rhoB[n]);
        rhoB[n] = ratioB * rho;
        fB[9 * n + 4] = ratioB * t[4]; fB[9 * n + 7] = ratioB * t[7]; fB[9 * n + 8] = ratioB * t[8];
        vy[n] = v;
    }
}

/* A:607-650 ghostPointsConstantVelocityRK (row ny-1 <- S neighbour, rho = sum) */
void rk_ghost_inlet_velocity(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr,
                             double *rhoR, double *rhoB, double *fR, double *fB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < ny * nx && loc >= (ny - 1) * nx)) continue;
        i64 L = nbr[8 * n + 3];
        double *r = fR + 9 * n, *b = fB + 9 * n;
        for (int i = 0; i < 9; ++i) r[i] = fR[9 * L + i];
        rhoR[n] = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8];
        for (int i = 0; i < 9; ++i) b[i] = fB[9 * L + i];
        rhoB[n] = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + b[7] + b[8];
    }
}

/* A:925-962 calConstPressureInletGPU (row ny-2, Zou-He pressure per colour) */
void rk_inlet_pressure(i64 N, i64 nx, i64 ny, double pB, double pR, const i64 *fluidNodes,
                       double *rhoB, double *rhoR, double *fB, double *fR)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= (ny - 2) * nx && loc < (ny - 1) * nx)) continue;
        double *b = fB + 9 * n, *r = fR + 9 * n;
        double vB = -1. + (b[0] + b[1] + b[3] + 2. * (b[2] + b[5] + b[6])) / pB;
        b[4] = b[2] - 2. / 3. * pB * vB;
        b[7] = b[5] + 1. / 2. * (b[1] - b[3]) - 1. / 6. * pB * vB;
        b[8] = b[6] - 1. / 2. * (b[1] - b[3]) - 1. / 6. * pB * vB;
        rhoB[n] = pB;
        double vR = -1. + (r[0] + r[1] + r[3] + 2. * (r[2] + r[5] + r[6])) / pR;
        r[4] = r[2] - 2. / 3. * pR * vR;
        r[7] = r[5] + 1. / 2. * (r[1] - r[3]) - 1. / 6. * pR * vR;
        r[8] = r[6] - 1. / 2. * (r[1] - r[3]) - 1. / 6. * pR * vR;
        rhoR[n] = pR;
    }
}

/* A:968-1002 ghostPointsConstPressureInletRK (row ny-1 <- S neighbour incl. rho) */
void rk_ghost_inlet_pressure(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr,
                             double *rhoR, double *rhoB, double *fR, double *fB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= (ny - 1) * nx && loc < ny * nx)) continue;
        i64 H = nbr[8 * n + 3];
        for (int i = 0; i < 9; ++i) { fR[9 * n + i] = fR[9 * H + i]; fB[9 * n + i] = fB[9 * H + i]; }
        rhoR[n] = rhoR[H]; rhoB[n] = rhoB[H];
    }
}

/* A:2560-2590 calConstPressureLowerGPUTotal (row 1, Zou-He pressure on f_tot) */
void rk_outlet_pressure_total(i64 N, i64 nx, double pL, const i64 *fluidNodes, double *fT,
                              double *vy, const double *rhoR, const double *rhoB,
                              double *fR, double *fB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= nx && loc < 2 * nx)) continue;
        double *t = fT + 9 * n;
        double v = 1. - 1. / pL * (t[0] + t[1] + t[3] + 2. * (t[4] + t[7] + t[8]));
        t[2] = t[4] + 2. / 3. * (pL * v);
        t[5] = t[7] + 0.5 * (t[3] - t[1]) + 1. / 6. * pL * v;
        t[6] = t[8] + 0.5 * (t[1] - t[3]) + 1. / 6. * pL * v;
        vy[n] = v;
        double ratioR = rhoR[n] / (rhoR[n] + rhoB[n]);
        fR[9 * n + 2] = ratioR * t[2]; fR[9 * n + 5] = ratioR * t[5]; fR[9 * n + 6] = ratioR * t[6];
        double ratioB = rhoB[n] / (rhoR[n] + rhoB[n]);
        fB[9 * n + 2] = ratioB * t[2]; fB[9 * n + 5] = ratioB * t[5]; fB[9 * n + 6] = ratioB * t[6];
    }
}

/* A:1045-1081 ghostPointsConstPressureLowerRK: acts on COMPACT indices < nx
 * (reference quirk; equals grid row 0 when row 0 is all fluid). */
void rk_ghost_outlet_pressure(i64 N, i64 nx, const i64 *nbr, double *rhoR, double *rhoB,
                              double *fR, double *fB)
{
    i64 lim = nx < N ? nx : N;
    PARFOR
    for (i64 n = 0; n < lim; ++n) {
        i64 L = nbr[8 * n + 1];
        for (int i = 0; i < 9; ++i) { fR[9 * n + i] = fR[9 * L + i]; fB[9 * n + i] = fB[9 * L + i]; }
        rhoR[n] = rhoR[L]; rhoB[n] = rhoB[L];
    }
}

/* A:700-784 convectiveOutletGPU / Ghost2GPU / Ghost3GPU: row r <- N neighbour,
 * rho re-summed; three sequential launches for rows 2, 1, 0. */
void rk_outlet_convective_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr,
                              double *fR, double *fB, double *rhoR, double *rhoB)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        i64 q = nbr[8 * n + 1];
        double r = 0., b = 0.;
        for (int j = 0; j < 9; ++j) {
            fR[9 * n + j] = fR[9 * q + j];
            fB[9 * n + j] = fB[9 * q + j];
            r += fR[9 * n + j]; b += fB[9 * n + j];
        }
        rhoR[n] = r; rhoB[n] = b;
    }
}

/* ------------------------------------------------------------------ the time loop */

typedef struct {
    i64 N, nx, ny, W, Wf;
    const i64 *fluidNodes, *nbr, *nbrWet, *fluidWet;
    const double *nsx, *nsy;
    /* parameters */
    double sigma, cosT, sinT, beta, delta, tauR, tauB;
    double vyIn;             /* velocityYB + velocityYR, D:1300 */
    double pInB, pInR;       /* densityBH, densityRH */
    double pOutTotal;        /* densityBL + densityRL, D:1344 */
    int wettingType, tauType, mrt, inletType /*0 Neumann 1 Dirichlet*/,
        outletType /*0 Dirichlet 1 Convective*/;
    const double *M, *Minv, *S;
    /* state */
    double *fR, *fB, *fRn, *fBn, *fT, *rhoR, *rhoB, *vx, *vy, *phi, *phiS, *Gx, *Gy, *Fx, *Fy, *K;
} rk_sim;

/* One pass of the while-loop body of runRKColorGradient2DCSF, D:1295-1490 */
/* first half: boundary kernels ... wetting-corrected colour gradient (D:1299-1424) */
void rk_csf_step_a(rk_sim *s)
{
    i64 N = s->N;
    if (s->inletType == 0) {
        rk_inlet_velocity_total(N, s->nx, s->ny, s->vyIn, s->fluidNodes, s->rhoR, s->rhoB,
                                s->fR, s->fB, s->fT, s->vy);
        rk_ghost_inlet_velocity(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    } else {
        rk_inlet_pressure(N, s->nx, s->ny, s->pInB, s->pInR, s->fluidNodes, s->rhoB, s->rhoR, s->fB, s->fR);
        rk_ghost_inlet_pressure(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    }
    if (s->outletType == 1) {
        rk_outlet_convective_row(N, s->nx, 2, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
        rk_outlet_convective_row(N, s->nx, 1, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
        rk_outlet_convective_row(N, s->nx, 0, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
    } else {
        rk_outlet_pressure_total(N, s->nx, s->pOutTotal, s->fluidNodes, s->fT, s->vy, s->rhoR,
                                 s->rhoB, s->fR, s->fB);
        rk_ghost_outlet_pressure(N, s->nx, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    }
    rk_total_pdf(N, s->fR, s->fB, s->fT);
    rk_velocity(N, s->fT, s->rhoR, s->rhoB, s->vx, s->vy, s->Fx, s->Fy);
    rk_phase_field(N, s->rhoR, s->rhoB, s->phi);
    if (s->W > 0) rk_color_on_solid(s->W, s->nbrWet, s->phi, s->phiS);
    rk_gradient(N, s->nbr, s->phi, s->phiS, s->Gx, s->Gy);
    if (s->W > 0) {
        if (s->wettingType == 1) rk_wetting1(s->Wf, s->cosT, s->sinT, s->fluidWet, s->nsx, s->nsy, s->Gx, s->Gy);
        else if (s->wettingType == 2) rk_wetting2(s->Wf, s->cosT, s->sinT, s->fluidWet, s->nsx, s->nsy, s->Gx, s->Gy);
    }
}

/* The same first half in the order of the transport driver's loop (Transport2DRK.py:1177-1418): there the boundary rows come
 * FIRST (outlet :1199-1238, inlet :1245-1279) and the densities are summed from the populations AFTERWARDS (:1281-1287), so a
 * pressure row carries sum_i f_i instead of the prescribed density -- equal up to the last bit, but the wetting kernels' branch
 * switches (|G| thresholds, closer-candidate rule) turn that bit into 1e-6 at single nodes.  Config 4 is pinned in this order. */
void rk_csf_step_a_transport(rk_sim *s)
{
    i64 N = s->N;
    if (s->outletType == 1) {
        rk_outlet_convective_row(N, s->nx, 2, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
        rk_outlet_convective_row(N, s->nx, 1, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
        rk_outlet_convective_row(N, s->nx, 0, s->fluidNodes, s->nbr, s->fR, s->fB, s->rhoR, s->rhoB);
    } else {
        rk_outlet_pressure_total(N, s->nx, s->pOutTotal, s->fluidNodes, s->fT, s->vy, s->rhoR,
                                 s->rhoB, s->fR, s->fB);
        rk_ghost_outlet_pressure(N, s->nx, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    }
    if (s->inletType == 0) {
        rk_inlet_velocity_total(N, s->nx, s->ny, s->vyIn, s->fluidNodes, s->rhoR, s->rhoB,
                                s->fR, s->fB, s->fT, s->vy);
        rk_ghost_inlet_velocity(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    } else {
        rk_inlet_pressure(N, s->nx, s->ny, s->pInB, s->pInR, s->fluidNodes, s->rhoB, s->rhoR, s->fB, s->fR);
        rk_ghost_inlet_pressure(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rhoR, s->rhoB, s->fR, s->fB);
    }
    rk_total_pdf(N, s->fR, s->fB, s->fT);
    rk_macro_density(N, s->fR, s->fB, s->rhoR, s->rhoB);
    rk_velocity(N, s->fT, s->rhoR, s->rhoB, s->vx, s->vy, s->Fx, s->Fy);
    rk_phase_field(N, s->rhoR, s->rhoB, s->phi);
    if (s->W > 0) rk_color_on_solid(s->W, s->nbrWet, s->phi, s->phiS);
    rk_gradient(N, s->nbr, s->phi, s->phiS, s->Gx, s->Gy);
    if (s->W > 0) {
        if (s->wettingType == 1) rk_wetting1(s->Wf, s->cosT, s->sinT, s->fluidWet, s->nsx, s->nsy, s->Gx, s->Gy);
        else if (s->wettingType == 2) rk_wetting2(s->Wf, s->cosT, s->sinT, s->fluidWet, s->nsx, s->nsy, s->Gx, s->Gy);
    }
}

/* second half: CSF force ... streaming and densities (D:1425-1490) */
void rk_csf_step_b(rk_sim *s)
{
    i64 N = s->N;
    rk_force(N, s->wettingType, s->sigma, s->nbr, s->Gx, s->Gy, s->Fx, s->Fy, s->K);
    if (!s->mrt) {
        rk_collide_srt(N, s->tauType, s->tauR, s->tauB, s->delta, s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fT);
        rk_force_srt(N, s->tauType, s->tauR, s->tauB, s->delta, s->vx, s->vy, s->Fx, s->Fy, s->phi, s->fT, s->rhoR, s->rhoB);
    } else {
        rk_collide_mrt(N, s->tauType, s->tauR, s->tauB, s->delta, s->vx, s->vy, s->rhoR, s->rhoB, s->phi, s->fT, s->M, s->Minv, s->S);
        rk_force_mrt(N, s->tauType, s->tauR, s->tauB, s->delta, s->vx, s->vy, s->Fx, s->Fy, s->phi, s->fT, s->M, s->Minv, s->S, s->rhoR, s->rhoB);
    }
    rk_recolor(N, s->beta, s->rhoR, s->rhoB, s->Gx, s->Gy, s->fR, s->fB, s->fT);
    rk_stream1(N, s->nbr, s->fR, s->fRn);
    rk_stream1(N, s->nbr, s->fB, s->fBn);
    rk_stream2(N, s->fRn, s->fR);
    rk_stream2(N, s->fBn, s->fB);
    rk_total_pdf(N, s->fR, s->fB, s->fT);
    rk_macro_density(N, s->fR, s->fB, s->rhoR, s->rhoB);
}

void rk_csf_step(rk_sim *s) { rk_csf_step_a(s); rk_csf_step_b(s); }

void rk_csf_run(rk_sim *s, i64 nsteps)
{
    for (i64 k = 0; k < nsteps; ++k) rk_csf_step(s);
}

void rk_oracle_set_threads(int n)
{
#if defined(_OPENMP)
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int rk_oracle_threads(void)
{
#if defined(_OPENMP)
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
