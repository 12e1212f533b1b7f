/*
 * oracle/sc_oracle.c -- CPU restatement of the reference's Shan-Chen D2Q9 paths
 * (original Shan-Chen and the explicit-forcing scheme "EFS", Porter et al. 2012).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/rk_oracle.c for the rules).  Restated from
 * reading ShanChen2D/OptimizedD2Q9GPU.py ("O:"), ShanChen2D/ExplicitD2Q9GPU.py ("E:")
 * and the driver loops in ShanChen2D/ShanChenD2Q9.py ("D:"); arithmetic follows the
 * reference's evaluation order.
 *
 * Parity status: PINNED against golden vectors captured from the real reference drivers
 * runOptimizedEFLBM / runOptimizedLBM (tests/golden/gen/make_golden_sc.py ->
 * tests/golden/sc_*.npz; tests/test_oracle_sc.py).
 *
 * Layout = the reference's: f[nF][N][9], rho[nF][N], forces [nF][N]; nbr[8N] with -1 for
 * every non-fluid neighbour (D:587-659).  nF is fixed to 2 like every shipped ini (and
 * like the reference's outlet kernel, O:560-561).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
#define NF 2

static const double EX[9] = {0., 1., 0., -1., 0., 1., -1., -1., 1.};
static const double EY[9] = {0., 0., 1., 0., -1., 1., 1., -1., -1.};
static const double WT[9] = {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9.,
                             1. / 36., 1. / 36., 1. / 36., 1. / 36.};
static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};

#if defined(_OPENMP)
#define PARFOR _Pragma("omp parallel for schedule(static)")
#else
#define PARFOR
#endif

#define F(f, k, n, j) (f)[((size_t)(k) * N + (n)) * 9 + (j)]
#define R(r, k, n) (r)[(size_t)(k) * N + (n)]

/* D:587-606 optimizeFluidArray: compaction, every non-fluid node -> -1 */
i64 sc_compact(i64 nx, i64 ny, const uint8_t *isDomain, i64 *fluidNodes, i64 *newIndex)
{
    i64 n = 0;
    for (i64 k = 0; k < nx * ny; ++k) {
        newIndex[k] = -1;
        if (isDomain[k] == 1) { fluidNodes[n] = k; newIndex[k] = n; ++n; }
    }
    return n;
}

/* O:84-94 calFluidRhoGPU */
void sc_rho(i64 N, double *rho, const double *f)
{
    for (int k = 0; k < NF; ++k) {
        PARFOR
        for (i64 n = 0; n < N; ++n) {
            double r = 0.;
            for (int j = 0; j < 9; ++j) r += F(f, k, n, j);
            R(rho, k, n) = r;
        }
    }
}

/* O:334-357 calMacroWholeVelocity: common velocity sum_k (momentum_k / tau_k) / sum_k (rho_k / tau_k).  The
 * original Shan-Chen loop launches it every step (D:1574); interactionCollisionProcess recomputes the same
 * quantity itself, so the output arrays feed nothing further. */
void sc_macro_whole_velocity(i64 N, const double *tau, const double *rho, const double *f, double *pvx, double *pvy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double vxt = 0., vyt = 0., rt = 0.;
        for (int k = 0; k < NF; ++k) {
            vxt += (F(f, k, n, 1) - F(f, k, n, 3) + F(f, k, n, 5) - F(f, k, n, 6) - F(f, k, n, 7) + F(f, k, n, 8)) / tau[k];
            vyt += (F(f, k, n, 2) - F(f, k, n, 4) + F(f, k, n, 5) + F(f, k, n, 6) - F(f, k, n, 7) - F(f, k, n, 8)) / tau[k];
            rt += R(rho, k, n) / tau[k];
        }
        pvx[n] = vxt / rt; pvy[n] = vyt / rt;
    }
}

/* O:156-180 calPhysicalVelocity */
void sc_physical_velocity(i64 N, const double *f, const double *rho, const double *Fx,
                          const double *Fy, double *vx, double *vy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double tx = 0., ty = 0., tr = 0.;
        for (int k = 0; k < NF; ++k) {
            tx += (F(f, k, n, 1) - F(f, k, n, 3) + F(f, k, n, 5) - F(f, k, n, 6) - F(f, k, n, 7) +
                   F(f, k, n, 8) + 1. / 2. * R(Fx, k, n));
            ty += (F(f, k, n, 2) - F(f, k, n, 4) + F(f, k, n, 5) + F(f, k, n, 6) - F(f, k, n, 7) -
                   F(f, k, n, 8) + 1. / 2. * R(Fy, k, n));
            tr += R(rho, k, n);
        }
        vx[n] = tx / tr;
        vy[n] = ty / tr;
    }
}

/* O:452-534 calStreaming1GPU (tests "!= -1") and O:539-550 calStreaming2GPU */
void sc_stream(i64 N, const i64 *nbr, double *f, double *fNew)
{
    for (int k = 0; k < NF; ++k) {
        PARFOR
        for (i64 n = 0; n < N; ++n)
            for (int i = 1; i < 9; ++i) {
                i64 q = nbr[8 * n + i - 1];
                if (q != -1) F(fNew, k, q, i) = F(f, k, n, i);
                else F(fNew, k, n, OPP[i]) = F(f, k, n, i);
            }
    }
    for (int k = 0; k < NF; ++k) {
        PARFOR
        for (i64 n = 0; n < N; ++n)
            for (int j = 1; j < 9; ++j) F(f, k, n, j) = F(fNew, k, n, j);
    }
}

/* O:839-863 constantVelocityZouHeBoundaryHigher (row ny-2, per component); O:868-895 ...Higher8 is
 * the same on row ny-3 */
void sc_inlet_velocity_row(i64 N, i64 nx, i64 row, const double *vyIn, const i64 *fluidNodes,
                           double *rho, double *f)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        for (int k = 0; k < NF; ++k) {
            double *g = &F(f, k, n, 0);
            R(rho, k, n) = (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / (1. + vyIn[k]);
            g[4] = g[2] - 2. / 3. * R(rho, k, n) * vyIn[k];
            g[7] = g[5] + (g[1] - g[3]) / 2. - 1. / 6. * R(rho, k, n) * vyIn[k];
            g[8] = g[6] - (g[1] - g[3]) / 2. - 1. / 6. * R(rho, k, n) * vyIn[k];
        }
    }
}

/* O:710-738 ghostPointsConstantVelocityInlet (row ny-1 <- S neighbour, rho summed); O:897-956
 * ghostPointsConstantVelocity8 / 82 are the same on rows ny-2, then ny-1 */
void sc_ghost_inlet_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *rho,
                        double *f)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        i64 L = nbr[8 * n + 3];
        for (int k = 0; k < NF; ++k) {
            double *g = &F(f, k, n, 0);
            for (int j = 0; j < 9; ++j) g[j] = F(f, k, L, j);
            R(rho, k, n) = g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] + g[8];
        }
    }
}

/* O:555-585 constantPressureZouHeBoundaryLower: densities HARD-CODED to (1.0, 0.02);
 * the densityL argument is ignored (reference quirk, replicated). */
void sc_outlet_pressure_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, double *rho, double *f)
{   /* row 1: ...Lower; row 2: O:590-620 ...Lower8 */
    static const double dens[2] = {1.0, 0.02};
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= row * nx && loc < (row + 1) * nx)) continue;
        for (int k = 0; k < NF; ++k) {
            double *g = &F(f, k, n, 0);
            double d = dens[k];
            double v = 1. - (g[0] + g[1] + g[3] + 2. * (g[4] + g[7] + g[8])) / d;
            g[2] = g[4] + 2. / 3. * v * d;
            g[5] = g[7] + 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
            g[6] = g[8] - 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
            R(rho, k, n) = d;
        }
    }
}

/* O:743-770 ghostPointsConstantPressureOutlet (row 0 <- N neighbour, rho summed); O:775-836
 * ghostPointsConstantPressureOutlet8 / 82 are the same on rows 1, then 0 */
void sc_ghost_outlet_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc >= row * nx && loc < (row + 1) * nx)) continue;
        i64 H = nbr[8 * n + 1];
        for (int k = 0; k < NF; ++k) {
            double *g = &F(f, k, n, 0);
            for (int j = 0; j < 9; ++j) g[j] = F(f, k, H, j);
            R(rho, k, n) = g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] + g[8];
        }
    }
}

void sc_inlet_velocity(i64 N, i64 nx, i64 ny, const double *vyIn, const i64 *fluidNodes, double *rho, double *f)
{ sc_inlet_velocity_row(N, nx, ny - 2, vyIn, fluidNodes, rho, f); }
void sc_ghost_inlet(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{ sc_ghost_inlet_row(N, nx, ny - 1, fluidNodes, nbr, rho, f); }
void sc_outlet_pressure(i64 N, i64 nx, const i64 *fluidNodes, double *rho, double *f)
{ sc_outlet_pressure_row(N, nx, 1, fluidNodes, rho, f); }
void sc_ghost_outlet(i64 N, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{ sc_ghost_outlet_row(N, nx, 0, fluidNodes, nbr, rho, f); }

/* O:960-1038 convectiveOutletGPU / Ghost2GPU / Ghost3GPU (original SC): row <- N neighbour */
void sc_outlet_copy_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f,
                        double *rho)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        i64 q = nbr[8 * n + 1];
        for (int k = 0; k < NF; ++k) {
            double r = 0.;
            for (int j = 0; j < 9; ++j) { F(f, k, n, j) = F(f, k, q, j); r += F(f, k, q, j); }
            R(rho, k, n) = r;
        }
    }
}

/* O:1044-1120 convectiveOutletEachGPU / Each2GPU / Each3GPU (EFS):
 * f = (f_old + |vy(row 3)| f(N neighbour)) / (1 + |vy(row 3)|), rows 2, 1, 0 in sequence */
void sc_outlet_convective_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr,
                              double *f, const double *fOld, double *rho, const double *vy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        i64 q1 = nbr[8 * n + 1];
        i64 q = q1;                         /* node on row 3 above this column */
        for (i64 h = row; h < 2; ++h) q = nbr[8 * q + 1];
        double v = fabs(vy[q]);
        for (int k = 0; k < NF; ++k) {
            double r = 0.;
            for (int j = 0; j < 9; ++j) {
                F(f, k, n, j) = (F(fOld, k, n, j) + v * F(f, k, q1, j)) / (1. + v);
                r += F(f, k, n, j);
            }
            R(rho, k, n) = r;
        }
    }
}

/* O:1274-1449 interactionCollisionProcess (original Shan-Chen, fused force + BGK).
 * weightInter = 1/9 x4, 1/36 x4 (D:1476-1478); solid adhesion weights hard-coded 1/9, 1/36. */
void sc_interaction_collision(i64 N, const double *tau, const double *G /*[2][2]*/,
                              const double *Gs /*[2]*/, const double *rho, const double *psi,
                              double *f, const i64 *nbr, double *Fx, double *Fy)
{
    static const double wi[8] = {1. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 36., 1. / 36., 1. / 36., 1. / 36.};
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double vxt = 0., vyt = 0., rt = 0.;
        for (int k = 0; k < NF; ++k) {
            vxt += (F(f, k, n, 1) - F(f, k, n, 3) + F(f, k, n, 5) - F(f, k, n, 6) - F(f, k, n, 7) + F(f, k, n, 8)) / tau[k];
            vyt += (F(f, k, n, 2) - F(f, k, n, 4) + F(f, k, n, 5) + F(f, k, n, 6) - F(f, k, n, 7) - F(f, k, n, 8)) / tau[k];
            rt += R(rho, k, n) / tau[k];
        }
        double pvx = vxt / rt, pvy = vyt / rt;
        for (int k = 0; k < NF; ++k) {
            double fx = 0., fy = 0.;
            for (int i = 0; i < 8; ++i) {
                i64 q = nbr[8 * n + i];
                double ex = EX[i + 1], ey = EY[i + 1];
                if (q != -1) {
                    for (int j = 0; j < NF; ++j) {
                        if (ex != 0.) fx += -wi[i] * G[k * NF + j] * R(psi, k, n) * R(psi, j, q) * (ex);
                        if (ey != 0.) fy += -wi[i] * G[k * NF + j] * R(psi, k, n) * R(psi, j, q) * (ey);
                    }
                } else {
                    double ws = (i < 4) ? 1. / 9. : 1. / 36.;
                    if (ex != 0.) fx += -ws * Gs[k] * R(psi, k, n) * (ex);
                    if (ey != 0.) fy += -ws * Gs[k] * R(psi, k, n) * (ey);
                }
            }
            R(Fx, k, n) = fx; R(Fy, k, n) = fy;
            double ux = pvx + tau[k] * fx / R(rho, k, n);
            double uy = pvy + tau[k] * fy / R(rho, k, n);
            double usq = ux * ux + uy * uy;
            double *g = &F(f, k, n, 0);
            double rk = R(rho, k, n), tk = tau[k];
            g[0] = (1 - 1. / tk) * g[0] + WT[0] * rk / tk * (1. - 1.5 * usq);
            g[1] = (1 - 1. / tk) * g[1] + WT[1] * rk / tk * (1. + 3. * ux + 4.5 * (ux * ux) - 1.5 * usq);
            g[2] = (1. - 1. / tk) * g[2] + WT[2] * rk / tk * (1. + 3. * uy + 4.5 * (uy * uy) - 1.5 * usq);
            g[3] = (1. - 1. / tk) * g[3] + WT[3] * rk / tk * (1. + 3. * (-ux) + 4.5 * ((-ux) * (-ux)) - 1.5 * usq);
            g[4] = (1. - 1. / tk) * g[4] + WT[4] * rk / tk * (1. + 3. * (-uy) + 4.5 * ((-uy) * (-uy)) - 1.5 * usq);
            double sq = (ux + uy) * (ux + uy);
            g[5] = (1. - 1. / tk) * g[5] + WT[5] * rk / tk * (1. + 3. * (ux + uy) + 4.5 * sq - 1.5 * usq);
            sq = (-ux + uy) * (-ux + uy);
            g[6] = (1. - 1. / tk) * g[6] + WT[6] * rk / tk * (1. + 3. * (-ux + uy) + 4.5 * sq - 1.5 * usq);
            sq = (-ux - uy) * (-ux - uy);
            g[7] = (1. - 1. / tk) * g[7] + WT[7] * rk / tk * (1. + 3. * (-ux - uy) + 4.5 * sq - 1.5 * usq);
            sq = (ux - uy) * (ux - uy);
            g[8] = (1. - 1. / tk) * g[8] + WT[8] * rk / tk * (1. + 3. * (ux - uy) + 4.5 * sq - 1.5 * usq);
        }
    }
}

/* E:51-216 calExplicit4thOrderScheme; weightInter4 = 1/3 x4, 1/12 x4 (D:1675) */
void sc_efs_force4(i64 N, const i64 *nbr, const double *G, const double *Gs, const double *psi,
                   double *Fx, double *Fy)
{
    static const double wi[8] = {1. / 3., 1. / 3., 1. / 3., 1. / 3., 1. / 12., 1. / 12., 1. / 12., 1. / 12.};
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        for (int k = 0; k < NF; ++k) {
            double gxs = 0., gys = 0., sx = 0., sy = 0.;
            for (int i = 0; i < 8; ++i) {
                i64 q = nbr[8 * n + i];
                double ex = EX[i + 1], ey = EY[i + 1];
                if (q != -1) {
                    for (int j = 0; j < NF; ++j) {
                        if (ex != 0.) gxs += wi[i] * (R(psi, j, q) - R(psi, j, n)) * (ex) * G[k * NF + j];
                        if (ey != 0.) gys += wi[i] * (R(psi, j, q) - R(psi, j, n)) * (ey) * G[k * NF + j];
                    }
                } else {
                    if (ex != 0.) sx += -wi[i] * Gs[k] * R(psi, k, n) * (ex);
                    if (ey != 0.) sy += -wi[i] * Gs[k] * R(psi, k, n) * (ey);
                }
            }
            double fx = 0., fy = 0.;
            fx += -6.0 * R(psi, k, n) * gxs;
            fy += -6.0 * R(psi, k, n) * gys;
            fx += sx;
            fy += sy;
            R(Fx, k, n) = fx; R(Fy, k, n) = fy;
        }
    }
}


/* ---- higher-isotropy force stencils, [ForceScheme] ExplicitScheme = 8 | 10
 * neighbour order of fillNeighboringNodesISO8 / ISO10 (E:392-486, :488-625): the 8 D2Q9 neighbours,
 * distance-2 axis, (2,2) diagonals, the 8 knight moves, then (ISO10) distance-3 axis and the 8 (3,1) moves */
static const int ISO_DX[36] = {1, 0, -1, 0, 1, -1, -1, 1, 2, 0, -2, 0, 2, -2, -2, 2, 2, 1, -1, -2, -2, -1, 1, 2,
                               3, 0, -3, 0, 3, 1, -1, -3, -3, -1, 1, 3};
static const int ISO_DY[36] = {0, 1, 0, -1, 1, 1, -1, -1, 0, 2, 0, -2, 2, 2, -2, -2, 1, 2, 2, 1, -1, -2, -2, -1,
                               0, 3, 0, -3, 1, 3, 3, 1, -1, -3, -3, -1};

void sc_fill_neighbors_iso(i64 N, i64 nx, i64 ny, int nn, const i64 *fluidNodes, const i64 *newidx, i64 *out)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        i64 i = fluidNodes[n] / nx, j = fluidNodes[n] % nx;
        for (int m = 0; m < nn; ++m) {
            i64 ii = (i + ISO_DY[m] + ny) % ny, jj = (j + ISO_DX[m] + nx) % nx;
            out[nn * n + m] = newidx[ii * nx + jj];
        }
    }
}

/* weightInter8 / weightInter10, D:1677-1689 */
void sc_iso_weights(int scheme, double *w)
{
    if (scheme == 8) {
        for (int m = 0; m < 4; ++m) { w[m] = 4. / 21.; w[4 + m] = 4. / 45.; w[8 + m] = 1. / 60.; w[12 + m] = 1. / 5040.; }
        for (int m = 16; m < 24; ++m) w[m] = 2. / 315.;
    } else {
        for (int m = 0; m < 4; ++m) { w[m] = 262. / 1785.; w[4 + m] = 93. / 1190.; w[8 + m] = 7. / 340.; w[12 + m] = 9. / 9520.; w[24 + m] = 2. / 5355.; }
        for (int m = 16; m < 24; ++m) w[m] = 6. / 595.;
        for (int m = 28; m < 36; ++m) w[m] = 1. / 7140.;
    }
}

/* line-of-sight rule of the far neighbours (the `if` in front of every block of E:627-955 / :957-1377): a
 * neighbour at distance 2 counts only if the nearest neighbour on the way to it is fluid, a knight-move
 * neighbour if one of the two nearest neighbours on the way is, distance 3 if the whole path is */
static int iso_gate(const i64 *nb, int m)
{
    static const int K2[8][2] = {{0, 4}, {1, 4}, {1, 5}, {2, 5}, {2, 6}, {3, 6}, {3, 7}, {0, 7}};
    static const int K3[8][4] = {{4, 16, 0, 8}, {1, 9, 4, 17}, {1, 9, 5, 18}, {2, 10, 5, 19},
                                 {2, 10, 6, 20}, {3, 11, 6, 21}, {3, 11, 7, 22}, {0, 8, 7, 23}};
#define FL(k) (nb[k] != -1)
    if (!FL(m)) return 0;
    if (m < 8) return 1;
    if (m < 16) return FL(m - 8);
    if (m < 24) return FL(K2[m - 16][0]) || FL(K2[m - 16][1]);
    if (m < 28) return FL(m - 24) && FL(m - 16);
    return (FL(K3[m - 28][0]) && FL(K3[m - 28][1])) || (FL(K3[m - 28][2]) && FL(K3[m - 28][3]));
#undef FL
}

/* E:627-955 calExplicit8thOrderScheme, E:957-1377 calExplicit10thOrderScheme: fluid neighbour n adds
 * -|e| 6 w_n G_ij psi_i (psi_j(n) - psi_j) sign(e) per axis (scheme 10: psi_j(n) alone); a missing neighbour adds the solid term
 * -1/9 (axis) / -1/36 (diagonal) Gs_i psi_i sign(e) for the 8 nearest neighbours only */
void sc_efs_force_iso(i64 N, int nn, const i64 *nbrX, const double *w, const double *G, const double *Gs,
                      const double *psi, double *Fx, double *Fy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double fx[NF], fy[NF];
        for (int i = 0; i < NF; ++i) { fx[i] = 0.0; fy[i] = 0.0; }
        for (int m = 0; m < nn; ++m) {
            i64 q = nbrX[nn * n + m];
            int dx = ISO_DX[m], dy = ISO_DY[m];
            if (iso_gate(nbrX + nn * n, m)) {
                for (int i = 0; i < NF; ++i)
                    for (int j = 0; j < NF; ++j) {
                        /* scheme 8 differences psi_j(n) - psi_j, scheme 10 uses the plain product (E:1009 ff.) */
                        double d = nn == 36 ? R(psi, j, q) : R(psi, j, q) - R(psi, j, n), sx = dx > 0 ? 1. : -1., sy = dy > 0 ? 1. : -1.;
                        if (dx == 1 || dx == -1) fx[i] += -6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sx);
                        if (dx == 2 || dx == -2) fx[i] += -2. * 6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sx);
                        if (dx == 3 || dx == -3) fx[i] += -3. * 6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sx);
                        if (dy == 1 || dy == -1) fy[i] += -6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sy);
                        if (dy == 2 || dy == -2) fy[i] += -2. * 6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sy);
                        if (dy == 3 || dy == -3) fy[i] += -3. * 6.0 * w[m] * G[i * NF + j] * R(psi, i, n) * (d) * (sy);
                    }
            } else if (m < 8 && q == -1) {
                for (int i = 0; i < NF; ++i) {
                    double c = m < 4 ? -1. / 9. : -1. / 36.;
                    if (dx != 0) fx[i] += c * Gs[i] * R(psi, i, n) * (dx > 0 ? 1. : -1.);
                    if (dy != 0) fy[i] += c * Gs[i] * R(psi, i, n) * (dy > 0 ? 1. : -1.);
                }
            }
        }
        for (int i = 0; i < NF; ++i) { R(Fx, i, n) = fx[i]; R(Fy, i, n) = fy[i]; }
    }
}


/* E:340-363 calEquilibriumVEFGPU (SRT: weights 1/tau_k) and E:1426-1449
 * transformEquilibriumVelocity (MRT: weights conserveS_k) */
void sc_efs_ueq(i64 N, const double *wk /* 1/tau_k or conserveS_k */, int divide, const double *rho,
                const double *Fx, const double *Fy, const double *f, double *ux, double *uy)
{
    PARFOR
    for (i64 n = 0; n < N; ++n) {
        double mx = 0., my = 0., rt = 0.;
        for (int k = 0; k < NF; ++k) {
            double ex = 0., ey = 0.;
            for (int j = 0; j < 9; ++j) { ex += F(f, k, n, j) * EX[j]; ey += F(f, k, n, j) * EY[j]; }
            ex += 1. / 2. * R(Fx, k, n);
            ey += 1. / 2. * R(Fy, k, n);
            if (divide) { mx += ex / wk[k]; my += ey / wk[k]; rt = rt + R(rho, k, n) / wk[k]; }
            else { mx += ex * wk[k]; my += ey * wk[k]; rt += R(rho, k, n) * wk[k]; }
        }
        ux[n] = mx / rt; uy[n] = my / rt;
    }
}

/* E:227-247 calEquilibriumFuncEFGPU */
void sc_efs_feq(i64 N, const double *rho, const double *ux, const double *uy, double *feq)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int k = 0; k < NF; ++k)
            for (int j = 0; j < 9; ++j)
                F(feq, k, n, j) = WT[j] * R(rho, k, n) *
                    (1. + 3. * (EX[j] * ux[n] + EY[j] * uy[n]) +
                     9. / 2. * ((EX[j] * ux[n] + EY[j] * uy[n]) * (EX[j] * ux[n] + EY[j] * uy[n])) -
                     3. / 2. * (ux[n] * ux[n] + uy[n] * uy[n]));
}

/* E:255-271 calForceDistrGPU */
void sc_efs_fforce(i64 N, const double *ux, const double *uy, const double *rho, const double *Fx,
                   const double *Fy, const double *feq, double *ff)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int k = 0; k < NF; ++k)
            for (int j = 0; j < 9; ++j)
                F(ff, k, n, j) = ((R(Fx, k, n) * (EX[j] - ux[n])) + (R(Fy, k, n) * (EY[j] - uy[n]))) *
                                 F(feq, k, n, j) / (1. / 3. * R(rho, k, n));
}

/* E:278-288 transformPDFGPU: f-bar = f - F_i/2 */
void sc_efs_transform(i64 N, double *f, const double *ff)
{
    PARFOR
    for (i64 q = 0; q < (i64)NF * N * 9; ++q) f[q] = f[q] - 1. / 2. * ff[q];
}

/* E:294-304 calCollisionEXGPU (SRT) */
void sc_efs_collide_srt(i64 N, const double *tau, double *f, const double *feq, const double *ff)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int k = 0; k < NF; ++k)
            for (int j = 0; j < 9; ++j)
                F(f, k, n, j) = F(f, k, n, j) + 1. / tau[k] * (F(feq, k, n, j) - F(f, k, n, j) -
                                 1. / 2. * F(ff, k, n, j)) + 1. * F(ff, k, n, j);
}

/* E:1404-1420 transfromForceTerm, E:1379-1399 transformPDFandEquil (f_eq overwritten by
 * Lambda f_eq), E:1457-1469 calAfterCollisionMRT.  Lambda_k = M^-1 S_k M, [2][9][9]. */
void sc_efs_collide_mrt(i64 N, const double *Lam, double *f, double *feq, const double *ff,
                        double *fM, double *ffM)
{
    PARFOR
    for (i64 n = 0; n < N; ++n)
        for (int k = 0; k < NF; ++k) {
            const double *L = Lam + (size_t)k * 81;
            double tF[9], tP[9], tE[9];
            for (int j = 0; j < 9; ++j) {
                double a = 0.;
                for (int m = 0; m < 9; ++m) a += L[9 * j + m] * F(ff, k, n, m);
                tF[j] = a;
            }
            for (int j = 0; j < 9; ++j) F(ffM, k, n, j) = tF[j];
            for (int j = 0; j < 9; ++j) {
                double a = 0., b = 0.;
                for (int m = 0; m < 9; ++m) {
                    a += L[9 * j + m] * F(f, k, n, m);
                    b += L[9 * j + m] * F(feq, k, n, m);
                }
                tP[j] = a; tE[j] = b;
            }
            for (int j = 0; j < 9; ++j) { F(fM, k, n, j) = tP[j]; F(feq, k, n, j) = tE[j]; }
            for (int j = 0; j < 9; ++j) {
                double c = (F(feq, k, n, j) - F(fM, k, n, j) - 1. / 2. * F(ffM, k, n, j));
                F(f, k, n, j) = F(f, k, n, j) + c + 1. * F(ff, k, n, j);
            }
        }
}

/* E:1476-1563 convectiveOutletGPUEFS / ...Ghost2GPUEFS / ...Ghost3GPUEFS (the 'Freeflow' outlet of the explicit forcing
 * loop, D:1865-1884): row `row` takes f-bar, F_i, f_eq of its N neighbour; rho re-summed from the source */
void sc_freeflow_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, double *rho, double *ff, double *feq)
{
    for (i64 n = 0; n < N; ++n) {
        const i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        const i64 q = nbr[8 * n + 1];
        for (int k = 0; k < NF; ++k) {
            R(rho, k, n) = 0.;
            for (int j = 0; j < 9; ++j) {
                F(f, k, n, j) = F(f, k, q, j);
                F(ff, k, n, j) = F(ff, k, q, j);
                F(feq, k, n, j) = F(feq, k, q, j);
                R(rho, k, n) += F(f, k, q, j);
            }
        }
    }
}

/* O:1127-1161 calVelocityBoundaryHigherChangGPU (row ny-2; scheme 4 only, D:1803 / :1999) */
void sc_inlet_chang_row(i64 N, i64 nx, i64 row, const double *vyIn, const i64 *fluidNodes, double *rho, const double *fOld, double *f)
{
    for (i64 n = 0; n < N; ++n) {
        const i64 loc = fluidNodes[n];
        if (!(loc < (row + 1) * nx && loc >= row * nx)) continue;
        for (int k = 0; k < NF; ++k) {
            const double v = vyIn[k];
            R(rho, k, n) = (F(f, k, n, 0) + F(f, k, n, 1) + F(f, k, n, 3) + 2. * (F(f, k, n, 2) + F(f, k, n, 5) + F(f, k, n, 6))) / (1. + v);
            F(f, k, n, 4) = F(fOld, k, n, 4) - 2. / 3. * (R(rho, k, n) * v + F(fOld, k, n, 4) + F(fOld, k, n, 7) + F(fOld, k, n, 8)) +
                            2. / 3. * (F(f, k, n, 2) + F(f, k, n, 5) + F(f, k, n, 6));
            F(f, k, n, 7) = F(fOld, k, n, 7) + 1. / 2. * (F(f, k, n, 1) - F(f, k, n, 3)) + 1. / 6. * (F(f, k, n, 2) - F(fOld, k, n, 4)) +
                            2. / 3. * (F(f, k, n, 5) - F(fOld, k, n, 7)) - 1. / 3. * (F(f, k, n, 6) - F(fOld, k, n, 8)) - 1. / 6. * R(rho, k, n) * v;
            F(f, k, n, 8) = F(fOld, k, n, 8) - 1. / 6. * R(rho, k, n) * v - 1. / 2. * (F(f, k, n, 1) - F(f, k, n, 3)) +
                            1. / 6. * (F(f, k, n, 2) - F(fOld, k, n, 4)) - 1. / 3. * (F(f, k, n, 5) - F(fOld, k, n, 7)) +
                            2. / 3. * (F(f, k, n, 6) - F(fOld, k, n, 8));
        }
    }
}

/* ------------------------------------------------------------------ drivers */
typedef struct {
    i64 N, nx, ny;
    const i64 *fluidNodes, *nbr;
    double tau[NF], G[NF * NF], Gs[NF], vyIn[NF];
    int mrt, outletType /*0 Dirichlet 1 Convective 2 none (periodic box) 3 Freeflow (EFS, SRT)*/;
    const double *Lam;   /* [2][9][9], MRT only */
    double *f, *fOld, *fNew, *rho, *psi, *Fx, *Fy, *ux, *uy, *feq, *ff, *fM, *ffM, *vx, *vy;
    int scheme;          /* 4, 8 or 10 ([ForceScheme] ExplicitScheme) */
    const i64 *nbrX;     /* [N][24] or [N][36] for scheme 8 / 10 */
    double wX[36];
    int inletMethod;     /* 0 Zou-He, 1 Chang (scheme 4) */
    int fOldValid;       /* original Shan-Chen loop: fOld has been filled */
} sc_sim;

static void sc_efs_force_chain(sc_sim *s)
{   /* D:2039-2087 == D:1714-1768: psi, F, u_eq, f_eq, F_i */
    i64 N = s->N;
    memcpy(s->psi, s->rho, sizeof(double) * NF * N);            /* O:99-106 psi = rho */
    if (s->scheme == 8) sc_efs_force_iso(N, 24, s->nbrX, s->wX, s->G, s->Gs, s->psi, s->Fx, s->Fy);
    else if (s->scheme == 10) sc_efs_force_iso(N, 36, s->nbrX, s->wX, s->G, s->Gs, s->psi, s->Fx, s->Fy);
    else sc_efs_force4(N, s->nbr, s->G, s->Gs, s->psi, s->Fx, s->Fy);
    if (!s->mrt) sc_efs_ueq(N, s->tau, 1, s->rho, s->Fx, s->Fy, s->f, s->ux, s->uy);
    else { double ones[NF] = {1., 1.}; sc_efs_ueq(N, ones, 0, s->rho, s->Fx, s->Fy, s->f, s->ux, s->uy); }
    sc_efs_feq(N, s->rho, s->ux, s->uy, s->feq);
    sc_efs_fforce(N, s->ux, s->uy, s->rho, s->Fx, s->Fy, s->feq, s->ff);
}

static void sc_efs_bcs(sc_sim *s, int in_loop)
{
    i64 N = s->N, ny = s->ny;
    /* scheme 8 applies the same rules one row further inside and refreshes two ghost rows
     * (D:1798-1808, :1837-1849, :1942-1953, :1994-2020); scheme 10 has no boundary kernel in either place */
    const int sh = s->scheme == 8 ? 1 : 0, on = s->scheme != 10 && s->outletType != 2;   /* 2: periodic box, no boundary kernels */
    if (in_loop && s->outletType == 1) {          /* D:1913-1930 (not scheme dependent) */
        sc_outlet_convective_row(N, s->nx, 2, s->fluidNodes, s->nbr, s->f, s->fOld, s->rho, s->vy);
        sc_outlet_convective_row(N, s->nx, 1, s->fluidNodes, s->nbr, s->f, s->fOld, s->rho, s->vy);
        sc_outlet_convective_row(N, s->nx, 0, s->fluidNodes, s->nbr, s->f, s->fOld, s->rho, s->vy);
    }
    if (in_loop && s->outletType == 0 && on) {    /* D:1931-1953 */
        sc_outlet_pressure_row(N, s->nx, 1 + sh, s->fluidNodes, s->rho, s->f);
        for (int r = sh; r >= 0; --r) sc_ghost_outlet_row(N, s->nx, r, s->fluidNodes, s->nbr, s->rho, s->f);
    }
    if (on) {
        if (s->inletMethod == 1) sc_inlet_chang_row(N, s->nx, ny - 2, s->vyIn, s->fluidNodes, s->rho, s->fOld, s->f);   /* D:1803 / :1999 */
        else sc_inlet_velocity_row(N, s->nx, ny - 2 - sh, s->vyIn, s->fluidNodes, s->rho, s->f);       /* D:1990 / :1811 */
        for (int r = ny - 1 - sh; r <= ny - 1; ++r) sc_ghost_inlet_row(N, s->nx, r, s->fluidNodes, s->nbr, s->rho, s->f);
    }
    if (!in_loop && s->outletType == 0 && on) {   /* pre-loop order: inlet first, D:1827-1849 */
        sc_outlet_pressure_row(N, s->nx, 1 + sh, s->fluidNodes, s->rho, s->f);
        for (int r = sh; r >= 0; --r) sc_ghost_outlet_row(N, s->nx, r, s->fluidNodes, s->nbr, s->rho, s->f);
    }
}

/* pre-loop part of runOptimizedEFLBM, D:1714-1849 */
void sc_efs_prepare(sc_sim *s)
{
    memcpy(s->fOld, s->f, sizeof(double) * NF * s->N * 9);      /* D:1642: deviceFluidPDFold starts as a copy of the initial f */
    sc_efs_force_chain(s);
    sc_efs_transform(s->N, s->f, s->ff);
    sc_efs_bcs(s, 0);
}

/* one pass of the for-loop body, D:1852-2087 */
void sc_efs_iter(sc_sim *s)
{
    i64 N = s->N;
    memcpy(s->fOld, s->f, sizeof(double) * NF * N * 9);                               /* savePDFLastStep */
    if (s->outletType == 3)                                                             /* D:1865-1884 (SRT; see oracle/sc.py) */
        for (int r = 2; r >= 0; --r) sc_freeflow_row(N, s->nx, r, s->fluidNodes, s->nbr, s->f, s->rho, s->ff, s->feq);
    if (!s->mrt) sc_efs_collide_srt(N, s->tau, s->f, s->feq, s->ff);
    else sc_efs_collide_mrt(N, s->Lam, s->f, s->feq, s->ff, s->fM, s->ffM);
    sc_stream(N, s->nbr, s->f, s->fNew);
    sc_rho(N, s->rho, s->f);
    sc_physical_velocity(N, s->f, s->rho, s->Fx, s->Fy, s->vx, s->vy);
    sc_efs_bcs(s, 1);
    sc_rho(N, s->rho, s->f);
    sc_physical_velocity(N, s->f, s->rho, s->Fx, s->Fy, s->vx, s->vy);
    sc_efs_force_chain(s);
}

void sc_efs_run(sc_sim *s, i64 n) { for (i64 k = 0; k < n; ++k) sc_efs_iter(s); }

/* one pass of the while-loop body of runOptimizedLBM, D:1492-1629 (Neumann/ZouHe inlet,
 * Convective outlet) */
void sc_sc_iter(sc_sim *s)
{
    i64 N = s->N;
    if (s->outletType != 2) {        /* 2: periodic box (static-droplet case), boundary kernels skipped */
        if (s->inletMethod == 1) {   /* D:1529-1534: fOld = the initial f on the first pass (D:1446), then what D:1540 saved */
            if (!s->fOldValid) { memcpy(s->fOld, s->f, sizeof(double) * NF * N * 9); s->fOldValid = 1; }
            sc_inlet_chang_row(N, s->nx, s->ny - 2, s->vyIn, s->fluidNodes, s->rho, s->fOld, s->f);
        } else sc_inlet_velocity(N, s->nx, s->ny, s->vyIn, s->fluidNodes, s->rho, s->f);
        sc_ghost_inlet(N, s->nx, s->ny, s->fluidNodes, s->nbr, s->rho, s->f);
        if (s->inletMethod == 1) memcpy(s->fOld, s->f, sizeof(double) * NF * N * 9);          /* savePDFLastStep D:1540 */
    }
    sc_rho(N, s->rho, s->f);
    memcpy(s->psi, s->rho, sizeof(double) * NF * N);
    sc_interaction_collision(N, s->tau, s->G, s->Gs, s->rho, s->psi, s->f, s->nbr, s->Fx, s->Fy);
    sc_stream(N, s->nbr, s->f, s->fNew);
    if (s->outletType == 1) {
        sc_outlet_copy_row(N, s->nx, 2, s->fluidNodes, s->nbr, s->f, s->rho);
        sc_outlet_copy_row(N, s->nx, 1, s->fluidNodes, s->nbr, s->f, s->rho);
        sc_outlet_copy_row(N, s->nx, 0, s->fluidNodes, s->nbr, s->f, s->rho);
    }
    sc_rho(N, s->rho, s->f);
    sc_physical_velocity(N, s->f, s->rho, s->Fx, s->Fy, s->vx, s->vy);
}

void sc_sc_run(sc_sim *s, i64 n) { for (i64 k = 0; k < n; ++k) sc_sc_iter(s); }
