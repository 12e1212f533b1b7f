// sparse_kernels.hip -- kernel-level (drop-in) path: the reference's @cuda.jit kernels one by one on the
// reference's own sparse arrays (AoS f[N][9] / f[nF][N][9] / g[nT][N][5] float64, int64 neighbour tables),
// behind include/lbmpm_kernels.h.  This is the compatibility path that lets an unmodified reference-style
// driver loop run on MI355X (numba.cuda-shaped Python shim in openlbmpm_amd/dropin); the performance path is
// the fused solvers (rk2d.hip, sc2d.hip, rk3d.hip).  Hand-written for the GPU within what the reference's
// array layout allows:
//   * population arrays cross HBM in whole tiles: a block owns 256 consecutive compact nodes, whose 256 x Q
//     doubles are ONE contiguous run; lanes move it with unit stride through LDS (row stride Q is odd:
//     conflict-free 8-byte reads) and every thread then holds its node's Q values in registers.  (The
//     reference reads f[n][i] with one thread per node: a 72-byte stride per lane.)
//   * streaming is a pull: the coalesced side is the store (fNew in whole tiles), the gather side the load.
//     Equivalent to the reference's push + in-place bounce-back (A:340-417, O:452-550) for the neighbour
//     tables its fill kernels build (periodic, hence symmetric: q = nbr[n][i] <=> n = nbr[q][opposite i]).
//   * purely elementwise kernels run flat over all N x Q entries.
//   * boundary-row kernels launch nx threads, not N: the compact index of a grid node comes from a binary
//     search in fluidNodes (ascending by construction: RKD2Q9.py:603-655 scans row-major).  The reference
//     launches the whole lattice for one row.
// Arithmetic keeps the reference's statement order (file built with -ffp-contract=off): the drop-in loops
// reproduce the golden captures of the real drivers to 1e-11 (tests/test_dropin_gpu.py).
#include "lbmpm_common.h"
#include "../../include/lbmpm_kernels.h"

#include <cmath>

namespace {

using lbmpm::set_error;
typedef int64_t i64;

constexpr int NB = 256;            // nodes (= threads) per block of the tile kernels

// lattice constants (RKD2Q9.py:300-303, SimpleD2Q9.py:226; D2Q5: Transport2DRK.py:60-61, :314)
__device__ const double EX[9] = {0., 1., 0., -1., 0., 1., -1., -1., 1.};
__device__ const double EY[9] = {0., 0., 1., 0., -1., 1., 1., -1., -1.};
__device__ const double WT[9] = {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 36., 1. / 36., 1. / 36., 1. / 36.};
__device__ const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
__device__ const double VX[5] = {0., 1., -1., 0., 0.};
__device__ const double VY[5] = {0., 0., 0., 1., -1.};
__device__ const double WT5[5] = {1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.};
__device__ const int OPP5[5] = {0, 2, 1, 4, 3};
// neighbour order of fillNeighboringNodesISO8 / ISO10 (ExplicitD2Q9GPU.py:392, :488)
__device__ const int ISO_DX[36] = {1, 0, -1, 0, 1, -1, -1, 1, 2, 0, -2, 0, 2, -2, -2, 2, 2, 1, -1, -2, -2, -1, 1, 2,
                                   3, 0, -3, 0, 3, 1, -1, -3, -3, -1, 1, 3};
__device__ const int ISO_DY[36] = {0, 1, 0, -1, 1, 1, -1, -1, 0, 2, 0, -2, 2, 2, -2, -2, 1, 2, 2, 1, -1, -2, -2, -1,
                                   0, 3, 0, -3, 1, 3, 3, 1, -1, -3, -3, -1};

#define NF 2

// ------------------------------------------------------------------------------------------ tile access
// g[N][Q] <-> this thread's node in registers, through LDS.  Every thread of the block must call these.
template <int Q, typename T>
__device__ __forceinline__ void tile_in(T *lds, const T *g, i64 n0, i64 N, T r[Q])
{
    const int t = threadIdx.x;
    const i64 base = n0 * Q, lim = N * Q;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const i64 idx = base + k * NB + t;
        if (idx < lim) lds[k * NB + t] = g[idx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = lds[t * Q + i];
    __syncthreads();
}
// SKIP0: entry 0 of every node stays as it is in memory (the reference's streaming never writes direction 0)
template <int Q, bool SKIP0 = false>
__device__ __forceinline__ void tile_out(double *lds, double *g, i64 n0, i64 N, const double r[Q])
{
    const int t = threadIdx.x;
    const i64 base = n0 * Q, lim = N * Q;
#pragma unroll
    for (int i = 0; i < Q; ++i) lds[t * Q + i] = r[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int e = k * NB + t;
        const i64 idx = base + e;
        if (idx < lim && !(SKIP0 && e % Q == 0)) g[idx] = lds[e];
    }
    __syncthreads();
}

// compact index of grid node `loc`, or -1: fluidNodes is ascending (row-major scan of the lattice)
__device__ __forceinline__ i64 find_node(const i64 *fluidNodes, i64 N, i64 loc)
{
    i64 lo = 0, hi = N;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        if (fluidNodes[mid] < loc) lo = mid + 1;
        else hi = mid;
    }
    return (lo < N && fluidNodes[lo] == loc) ? lo : -1;
}
// the thread's node on grid row `row` (one thread per column), or -1
// The reference's boundary-row kernels index populations and densities with a neighbour id taken from the table unlooked-at; for a
// solid neighbour the id is negative (-1, or -2 - w for a wetting solid) and numba indexes like Python: a[-1] is the LAST node.  The
// same here (an image geometry without all-fluid rows next to the inlet / outlet rows gets there); an id that stays negative after
// the wrap is out of bounds in the reference as well: node 0.
__device__ __forceinline__ i64 nbr_node(i64 id, i64 N)
{
    if (id < 0) id += N;
    return id < 0 ? 0 : id;
}
__device__ __forceinline__ i64 row_node(const i64 *fluidNodes, i64 N, i64 nx, i64 row)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    return j < nx ? find_node(fluidNodes, N, row * nx + j) : -1;
}

#define GRID_NODES(N) dim3((unsigned)(((N) + NB - 1) / NB)), dim3(NB), 0, st
#define GRID_FLAT(cnt) dim3((unsigned)(((cnt) + 255) / 256)), dim3(256), 0, st
#define GRID_ROW(nx) dim3((unsigned)(((nx) + 63) / 64)), dim3(64), 0, st
#define THIS_NODE const i64 n0 = (i64)blockIdx.x * NB, n = n0 + threadIdx.x; const bool on = n < N

// tau(phi): AcceleratedRKGPU2D.py:1967-1981
__device__ double rk_tau(int option, double tauR, double tauB, double delta, double Phi, double rR, double rB)
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
// AcceleratedRKGPU2D.py:170-176 calEquilibriumRK2D
__device__ __forceinline__ double rk_feq(double rho, double w, double ex, double ey, double vx, double vy)
{
    return rho * w * (1 + (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) * (ex * vx + ey * vy) - 1.5 * (vx * vx + vy * vy)));
}
// d <- A d for a 9 x 9 matrix in global/constant memory (row-major), in the reference's accumulation order
__device__ __forceinline__ void mat9(const double *A, const double in[9], double out[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double t = 0.;
#pragma unroll
        for (int j = 0; j < 9; ++j) t += A[9 * i + j] * in[j];
        out[i] = t;
    }
}

// =========================================================================== colour gradient (RKCG2D)
// A:15-53 / A:58-95 fillNeighboringNodes / ...WettingNodes: periodic wrap on all four edges
__global__ void k_rk_fill_neighbors(i64 total, i64 nx, i64 ny, const i64 *nodes, const i64 *newIndex, i64 *nbr)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per table entry: unit-stride stores
    if (e >= 8 * total) return;
    const i64 k = e >> 3;
    const int d = (int)(e & 7);
    const i64 loc = nodes[k], i = loc / nx, j = loc % nx;
    const int dx = (int)EX[d + 1], dy = (int)EY[d + 1];
    const i64 jj = dx > 0 ? (j < nx - 1 ? j + 1 : 0) : (dx < 0 ? (j > 0 ? j - 1 : nx - 1) : j);
    const i64 ii = dy > 0 ? (i < ny - 1 ? i + 1 : 0) : (dy < 0 ? (i > 0 ? i - 1 : ny - 1) : i);
    nbr[e] = newIndex[ii * nx + jj];
}
static inline void launch_rk_fill_neighbors(hipStream_t st, i64 total, i64 nx, i64 ny, const i64 *nodes, const i64 *newIndex, i64 *nbr)
{
    if (total > 0) k_rk_fill_neighbors<<<GRID_FLAT(8 * total)>>>(total, nx, ny, nodes, newIndex, nbr);
}

// A:103-120 calMacroDensityRKGPU2D
__global__ __launch_bounds__(NB) void k_rk_macro_density(i64 N, const double *fR, const double *fB, double *rhoR, double *rhoB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double a[9], b[9];
    tile_in<9>(lds, fR, n0, N, a);
    tile_in<9>(lds, fB, n0, N, b);
    if (!on) return;
    double r = 0., s = 0.;
#pragma unroll
    for (int i = 0; i < 9; ++i) { r += a[i]; s += b[i]; }
    rhoR[n] = r; rhoB[n] = s;
}
static inline void launch_rk_macro_density(hipStream_t st, i64 N, const double *fR, const double *fB, double *rhoR, double *rhoB)
{
    if (N > 0) k_rk_macro_density<<<GRID_NODES(N)>>>(N, fR, fB, rhoR, rhoB);
}

// A:1414-1424 calTotalFluidPDF (flat)
__global__ void k_add_flat(i64 cnt, const double *a, const double *b, double *c)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) c[k] = a[k] + b[k];
}
static inline void launch_rk_total_pdf(hipStream_t st, i64 N, const double *fR, const double *fB, double *fT)
{
    if (N > 0) k_add_flat<<<GRID_FLAT(9 * N)>>>(9 * N, fR, fB, fT);
}

// A:2634-2654 calPhysicalVelocityRKGPU2DNew1: u = (sum e f_tot + F/2)/(rhoR+rhoB)
__global__ __launch_bounds__(NB) void k_rk_velocity(i64 N, const double *fT, const double *rhoR, const double *rhoB, double *vx, double *vy,
                                                    const double *Fx, const double *Fy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (!on) return;
    const double rs = rhoB[n] + rhoR[n];
    const double tx = f[1] - f[3] + f[5] - f[6] - f[7] + f[8] + 0.5 * Fx[n];
    vx[n] = tx / rs;
    const double ty = f[2] - f[4] + f[5] + f[6] - f[7] - f[8] + 0.5 * Fy[n];
    vy[n] = ty / rs;
}
static inline void launch_rk_velocity(hipStream_t st, i64 N, const double *fT, const double *rhoR, const double *rhoB, double *vx, double *vy,
                                      const double *Fx, const double *Fy)
{
    if (N > 0) k_rk_velocity<<<GRID_NODES(N)>>>(N, fT, rhoR, rhoB, vx, vy, Fx, Fy);
}
// A:125-147 calPhysicalVelocityRKGPU2D (perturbation loop): both colours, no force term
__global__ __launch_bounds__(NB) void k_rk_pert_velocity(i64 N, const double *fR, const double *fB, const double *rhoR, const double *rhoB,
                                                         double *vx, double *vy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double r[9], b[9];
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (!on) return;
    const double rho = rhoB[n] + rhoR[n];
    const double tx = r[1] - r[3] + r[5] - r[6] - r[7] + r[8] + b[1] - b[3] + b[5] - b[6] - b[7] + b[8];
    vx[n] = tx / rho;
    const double ty = r[2] - r[4] + r[5] + r[6] - r[7] - r[8] + b[2] - b[4] + b[5] + b[6] - b[7] - b[8];
    vy[n] = ty / rho;
}
static inline void launch_rk_pert_velocity(hipStream_t st, i64 N, const double *fR, const double *fB, const double *rhoR, const double *rhoB,
                                           double *vx, double *vy)
{
    if (N > 0) k_rk_pert_velocity<<<GRID_NODES(N)>>>(N, fR, fB, rhoR, rhoB, vx, vy);
}

// A:1348-1357 calPhaseFieldPhi
__global__ void k_rk_phase_field(i64 N, const double *rhoR, const double *rhoB, double *phi)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) phi[n] = (rhoR[n] - rhoB[n]) / (rhoR[n] + rhoB[n]);
}
static inline void launch_rk_phase_field(hipStream_t st, i64 N, const double *rhoR, const double *rhoB, double *phi)
{
    if (N > 0) k_rk_phase_field<<<GRID_FLAT(N)>>>(N, rhoR, rhoB, phi);
}

// A:1560-1581 calColorValueOnSolid: weighted mean of phi over the fluid neighbours of a wetting solid node
__global__ __launch_bounds__(NB) void k_rk_color_on_solid(i64 N /* wetting solids */, const i64 *nbrWet, const double *phi, double *phiS)
{
    __shared__ i64 lds[NB * 8];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(lds, nbrWet, n0, N, q);
    if (!on) return;
    double sum = 0., sw = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (q[i] >= 0) { sum += WT[i + 1] * phi[q[i]]; sw += WT[i + 1]; }
    phiS[n] = sum / sw;
}
static inline void launch_rk_color_on_solid(hipStream_t st, i64 W, const i64 *nbrWet, const double *phi, double *phiS)
{
    if (W > 0) k_rk_color_on_solid<<<GRID_NODES(W)>>>(W, nbrWet, phi, phiS);
}

// A:1584-1634 calRKInitialGradient: G = 3 sum_i w_i e_i phi(x+e_i); wetting solids carry phiS[-q-2]
__global__ __launch_bounds__(NB) void k_rk_gradient(i64 N, const i64 *nbr, const double *phi, const double *phiS, double *Gx, double *Gy)
{
    __shared__ i64 lds[NB * 8];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(lds, nbr, n0, N, q);
    if (!on) return;
    double gx = 0., gy = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double v = (q[i] >= 0) ? phi[q[i]] : phiS[-q[i] - 2];
        gx += WT[i + 1] * v * EX[i + 1];
        gy += WT[i + 1] * v * EY[i + 1];
    }
    Gx[n] = 3. * gx; Gy[n] = 3. * gy;
}
static inline void launch_rk_gradient(hipStream_t st, i64 N, const i64 *nbr, const double *phi, const double *phiS, double *Gx, double *Gy)
{
    if (N > 0) k_rk_gradient<<<GRID_NODES(N)>>>(N, nbr, phi, phiS, Gx, Gy);
}

// A:1639-1679 updateColorGradientOnWetting (WettingType 1, Xu 2017): one thread per wall-adjacent fluid node
__global__ void k_rk_wetting1(i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx, const double *nsy, double *Gx, double *Gy)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Wf) return;
    const double sx = nsx[k], sy = nsy[k];
    const double n1x = sx * cosT - sy * sinT, n1y = sy * cosT + sx * sinT;
    const double n2x = sx * cosT + sy * sinT, n2y = sy * cosT - sx * sinT;
    const i64 loc = fluidWet[k];
    const double gx = Gx[loc], gy = Gy[loc];
    const double nrm = sqrt(gx * gx + gy * gy);
    double ux = 0., uy = 0.;
    if (nrm > 1.0e-8) { ux = gx / nrm; uy = gy / nrm; }
    const double dx1 = ux - n1x, dy1 = uy - n1y, dx2 = ux - n2x, dy2 = uy - n2y;
    const double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
    double mx = 0., my = 0.;
    if (d1 < d2) { mx = n1x; my = n1y; }
    else if (d1 > d2) { mx = n2x; my = n2y; }
    else if (d1 == d2) { mx = sx; my = sy; }
    Gx[loc] = nrm * mx; Gy[loc] = nrm * my;
}
static inline void launch_rk_wetting1(hipStream_t st, i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx, const double *nsy,
                                      double *Gx, double *Gy)
{
    if (Wf > 0) k_rk_wetting1<<<GRID_FLAT(Wf)>>>(Wf, cosT, sinT, fluidWet, nsx, nsy, Gx, Gy);
}
// A:2430-2492 updateColorGradientOnWettingNew (WettingType 2, Akai 2018)
__global__ void k_rk_wetting2(i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx, const double *nsy, double *Gx, double *Gy)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Wf) return;
    const i64 loc = fluidWet[k];
    const double sx = nsx[k], sy = nsy[k], gx = Gx[loc], gy = Gy[loc];
    const double nrm = sqrt(gx * gx + gy * gy);
    double ux = 0., uy = 0.;
    if (nrm > 1.0e-8) { ux = -gx / nrm; uy = -gy / nrm; }
    const double ang = ux * sx + uy * sy;
    const double th = acos(ang);
    double c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
    if (fabs(sin(th)) > 1.0e-9) {
        c1 = sinT * cos(th) / sin(th);
        c2 = sinT / sin(th);
        c3 = -sinT * cos(th) / sin(th);
        c4 = -sinT / sin(th);
    }
    const double nx1 = (cosT - c1) * sx + c2 * ux, ny1 = (cosT - c1) * sy + c2 * uy;
    const double nx2 = (cosT - c3) * sx + c4 * ux, ny2 = (cosT - c3) * sy + c4 * uy;
    const double dx1 = nx1 - ux, dy1 = ny1 - uy, dx2 = nx2 - ux, dy2 = ny2 - uy;
    const double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
    if (d1 < d2) { Gx[loc] = -nrm * nx1; Gy[loc] = -nrm * ny1; }
    else if (d1 > d2) { Gx[loc] = -nrm * nx2; Gy[loc] = -nrm * ny2; }
    // d1 == d2 (or NaN): the gradient stays as it is, as in the reference
}
static inline void launch_rk_wetting2(hipStream_t st, i64 Wf, double cosT, double sinT, const i64 *fluidWet, const double *nsx, const double *nsy,
                                      double *Gx, double *Gy)
{
    if (Wf > 0) k_rk_wetting2<<<GRID_FLAT(Wf)>>>(Wf, cosT, sinT, fluidWet, nsx, nsy, Gx, Gy);
}

// A:1686-1736 (type 1: n = +G/|G|, thresholds == 0, F = +sigma K G / 2) and A:2499-2551 (type 2: n = -G/|G|,
// threshold 1e-8, F = -sigma K G / 2): curvature from the isotropic derivatives of n over the fluid neighbours
template <int TYPE>
__device__ __forceinline__ void unit_normal(double gx, double gy, double &ux, double &uy)
{
    const double nrm = sqrt(gx * gx + gy * gy);
    ux = 0.; uy = 0.;
    if (TYPE == 2) { if (nrm > 1.0e-8) { ux = -gx / nrm; uy = -gy / nrm; } }
    else if (nrm > 0.) { ux = gx / nrm; uy = gy / nrm; }
}
template <int TYPE>
__global__ __launch_bounds__(NB) void k_rk_force(i64 N, double sigma, const i64 *nbr, const double *Gx, const double *Gy, double *Fx, double *Fy, double *K)
{
    __shared__ i64 lds[NB * 8];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(lds, nbr, n0, N, q);
    if (!on) return;
    double ux, uy;
    unit_normal<TYPE>(Gx[n], Gy[n], ux, uy);
    double pyx = 0., pxy = 0., px = 0., py = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (q[i] < 0) continue;
        double qx, qy;
        unit_normal<TYPE>(Gx[q[i]], Gy[q[i]], qx, qy);
        pyx += 3. * WT[i + 1] * qy * EX[i + 1];
        pxy += 3. * WT[i + 1] * qx * EY[i + 1];
        px += 3. * WT[i + 1] * qx * EX[i + 1];
        py += 3. * WT[i + 1] * qy * EY[i + 1];
    }
    const double k = ux * uy * (pyx + pxy) - uy * uy * px - ux * ux * py;
    K[n] = k;
    const double sgn = TYPE == 2 ? -0.5 : 0.5;
    Fx[n] = sgn * sigma * k * Gx[n];
    Fy[n] = sgn * sigma * k * Gy[n];
}
static inline void launch_rk_force(hipStream_t st, i64 N, int type, double sigma, const i64 *nbr, const double *Gx, const double *Gy, double *Fx,
                                   double *Fy, double *K)
{
    if (N <= 0) return;
    if (type == 2) k_rk_force<2><<<GRID_NODES(N)>>>(N, sigma, nbr, Gx, Gy, Fx, Fy, K);
    else k_rk_force<1><<<GRID_NODES(N)>>>(N, sigma, nbr, Gx, Gy, Fx, Fy, K);
}

// A:1804-1848 calRKCollision1TotalGPU2DSRTM: BGK on f_tot
__global__ __launch_bounds__(NB) void k_rk_collide_srt(i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                                       const double *rhoR, const double *rhoB, const double *phi, double *fT)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (on) {
        const double rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
        const double tau = rk_tau(option, tauR, tauB, delta, phi[n], rR, rB);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double eT = rk_feq(rR, WT[i], EX[i], EY[i], ux, uy) + rk_feq(rB, WT[i], EX[i], EY[i], ux, uy);
            f[i] = -1. / tau * (f[i] - eT) + f[i];
        }
    }
    tile_out<9>(lds, fT, n0, N, f);
}
static inline void launch_rk_collide_srt(hipStream_t st, i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                         const double *rhoR, const double *rhoB, const double *phi, double *fT)
{
    if (N > 0) k_rk_collide_srt<<<GRID_NODES(N)>>>(N, option, tauR, tauB, delta, vx, vy, rhoR, rhoB, phi, fT);
}
// A:1743-1798 calPerturbationFromForce2D (Guo source, SRT)
__global__ __launch_bounds__(NB) void k_rk_force_srt(i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                                     const double *Fx, const double *Fy, const double *phi, double *fT, const double *rhoR,
                                                     const double *rhoB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (on) {
        const double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        const double fx = Fx[n], fy = Fy[n], ux = vx[n], uy = vy[n];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double src = WT[i] * ((3. * (EX[i] - ux) + 9. * EX[i] * (EX[i] * ux + EY[i] * uy)) * fx +
                                        (3. * (EY[i] - uy) + 9. * EY[i] * (EX[i] * ux + EY[i] * uy)) * fy) *
                               (1. - 1. / (2. * tau));
            f[i] = f[i] + src;
        }
    }
    tile_out<9>(lds, fT, n0, N, f);
}
static inline void launch_rk_force_srt(hipStream_t st, i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                       const double *Fx, const double *Fy, const double *phi, double *fT, const double *rhoR, const double *rhoB)
{
    if (N > 0) k_rk_force_srt<<<GRID_NODES(N)>>>(N, option, tauR, tauB, delta, vx, vy, Fx, Fy, phi, fT, rhoR, rhoB);
}
// A:1938-2017 calRKCollision1TotalGPU2DMRTM: f -= Minv S M (f - feq), S[7] = S[8] = 1/tau(phi)
__global__ __launch_bounds__(NB) void k_rk_collide_mrt(i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                                       const double *rhoR, const double *rhoB, const double *phi, double *fT, const double *M,
                                                       const double *Minv, const double *S)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (on) {
        const double rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
        const double tau = rk_tau(option, tauR, tauB, delta, phi[n], rR, rB);
        double d[9], m[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            d[i] = f[i] - (rk_feq(rR, WT[i], EX[i], EY[i], ux, uy) + rk_feq(rB, WT[i], EX[i], EY[i], ux, uy));
        mat9(M, d, m);
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = m[i] * (i >= 7 ? 1. / tau : S[i]);
        mat9(Minv, m, d);
#pragma unroll
        for (int i = 0; i < 9; ++i) f[i] = -d[i] + f[i];
    }
    tile_out<9>(lds, fT, n0, N, f);
}
static inline void launch_rk_collide_mrt(hipStream_t st, i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                         const double *rhoR, const double *rhoB, const double *phi, double *fT, const double *M, const double *Minv,
                                         const double *S)
{
    if (N > 0) k_rk_collide_mrt<<<GRID_NODES(N)>>>(N, option, tauR, tauB, delta, vx, vy, rhoR, rhoB, phi, fT, M, Minv, S);
}
// A:2027-2113 calPerturbationFromForce2DMRT: f += Minv (I - S/2) M src
__global__ __launch_bounds__(NB) void k_rk_force_mrt(i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                                     const double *Fx, const double *Fy, const double *phi, double *fT, const double *M,
                                                     const double *Minv, const double *S, const double *rhoR, const double *rhoB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (on) {
        const double tau = rk_tau(option, tauR, tauB, delta, phi[n], rhoR[n], rhoB[n]);
        const double fx = Fx[n], fy = Fy[n], ux = vx[n], uy = vy[n];
        double src[9], m[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double t1 = EX[i] * fx * 3.;
            const double t2 = EY[i] * fy * 3.;
            const double t3 = (EX[i] * EX[i] - 1. / 3.) * ux * fx * 9.;
            const double t4 = EX[i] * EY[i] * uy * fx * 9.;
            const double t5 = EY[i] * EX[i] * ux * fy * 9.;
            const double t6 = (EY[i] * EY[i] - 1. / 3.) * uy * fy * 9.;
            src[i] = WT[i] * (t1 + t2 + t3 + t4 + t5 + t6);
        }
        mat9(M, src, m);
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = (i >= 7 ? 1. - 0.5 * 1. / tau : 1. - 0.5 * S[i]) * m[i];
        mat9(Minv, m, src);
#pragma unroll
        for (int i = 0; i < 9; ++i) f[i] = f[i] + src[i];
    }
    tile_out<9>(lds, fT, n0, N, f);
}
static inline void launch_rk_force_mrt(hipStream_t st, i64 N, int option, double tauR, double tauB, double delta, const double *vx, const double *vy,
                                       const double *Fx, const double *Fy, const double *phi, double *fT, const double *M, const double *Minv,
                                       const double *S, const double *rhoR, const double *rhoB)
{
    if (N > 0) k_rk_force_mrt<<<GRID_NODES(N)>>>(N, option, tauR, tauB, delta, vx, vy, Fx, Fy, phi, fT, M, Minv, S, rhoR, rhoB);
}
// A:1857-1899 calRecoloringProcessM
__global__ __launch_bounds__(NB) void k_rk_recolor(i64 N, double beta, const double *rhoR, const double *rhoB, const double *Gx, const double *Gy,
                                                   double *fR, double *fB, const double *fT)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double t[9], r[9], b[9];
    tile_in<9>(lds, fT, n0, N, t);
    if (on) {
        const double gx = Gx[n], gy = Gy[n], rR = rhoR[n], rB = rhoB[n];
        const double gn = sqrt(gx * gx + gy * gy), tot = rR + rB;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double un = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            double c = 0.;
            if (gn > 1.0e-8 && un > 1.0e-8) c = (EX[i] * gx + EY[i] * gy) / (un * gn);
            r[i] = rR / tot * t[i] + beta * rR * rB / tot * WT[i] * c * un;
            b[i] = rB / tot * t[i] - beta * rR * rB / tot * WT[i] * c * un;
        }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_recolor(hipStream_t st, i64 N, double beta, const double *rhoR, const double *rhoB, const double *Gx, const double *Gy,
                                     double *fR, double *fB, const double *fT)
{
    if (N > 0) k_rk_recolor<<<GRID_NODES(N)>>>(N, beta, rhoR, rhoB, Gx, Gy, fR, fB, fT);
}

// ---- the perturbation loop's own collision kernels (RKD2Q9.py:1164-1211; the 2-D loop the D3Q19 model extends)
__device__ __forceinline__ double tau_harmonic(double phi, double tauR, double tauB)      // A:1144-1145, A:1307-1308
{
    return 0.5 + 1. / ((1. + phi) / (2. * (tauR - 0.5)) + (1. - phi) / (2. * (tauB - 0.5)));
}
// A:1125-1163 calRKCollision1GPU2DSRTNew: BGK on each colour, in place
__global__ __launch_bounds__(NB) void k_rk_pert_collide1_srt(i64 N, double tauR, double tauB, const double *vx, const double *vy, const double *rhoR,
                                                             const double *rhoB, const double *phi, double *fR, double *fB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double r[9], b[9];
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (on) {
        const double tau = tau_harmonic(phi[n], tauR, tauB), rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double cR = -1. / tau * (r[i] - rk_feq(rR, WT[i], EX[i], EY[i], ux, uy));
            const double cB = -1. / tau * (b[i] - rk_feq(rB, WT[i], EX[i], EY[i], ux, uy));
            r[i] = r[i] + cR;
            b[i] = b[i] + cB;
        }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_pert_collide1_srt(hipStream_t st, i64 N, double tauR, double tauB, const double *vx, const double *vy, const double *rhoR,
                                               const double *rhoB, const double *phi, double *fR, double *fB)
{
    if (N > 0) k_rk_pert_collide1_srt<<<GRID_NODES(N)>>>(N, tauR, tauB, vx, vy, rhoR, rhoB, phi, fR, fB);
}
// A:1272-1343 calRKCollision1GPU2DMRTNew: fT += -Minv S M (fT - feq) + wF_i (e_i . bodyF), S[7] = S[8] = 1/tau
__global__ __launch_bounds__(NB) void k_rk_pert_collide1_mrt(i64 N, double tauR, double tauB, double bodyFX, double bodyFY, const double *vx,
                                                             const double *vy, const double *rhoR, const double *rhoB, const double *phi, double *fT,
                                                             const double *M, const double *Minv, const double *S)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (on) {
        const double tau = tau_harmonic(phi[n], tauR, tauB), rho = rhoB[n] + rhoR[n], ux = vx[n], uy = vy[n];
        double eq[9], m[9], a[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) eq[i] = rk_feq(rho, WT[i], EX[i], EY[i], ux, uy);
#pragma unroll
        for (int i = 0; i < 9; ++i) {            // the reference transforms f and feq separately, then subtracts
            double s = 0., e = 0.;
#pragma unroll
            for (int j = 0; j < 9; ++j) { s += M[9 * i + j] * f[j]; e += M[9 * i + j] * eq[j]; }
            m[i] = (s - e) * (i >= 7 ? 1. / tau : S[i]);
        }
        mat9(Minv, m, a);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double wf = i == 0 ? 0. : (i < 5 ? 1. / 3. : 1. / 12.);
            f[i] = -a[i] + wf * (EX[i] * bodyFX + EY[i] * bodyFY) + f[i];
        }
    }
    tile_out<9>(lds, fT, n0, N, f);
}
static inline void launch_rk_pert_collide1_mrt(hipStream_t st, i64 N, double tauR, double tauB, double bodyFX, double bodyFY, const double *vx,
                                               const double *vy, const double *rhoR, const double *rhoB, const double *phi, double *fT, const double *M,
                                               const double *Minv, const double *S)
{
    if (N > 0) k_rk_pert_collide1_mrt<<<GRID_NODES(N)>>>(N, tauR, tauB, bodyFX, bodyFY, vx, vy, rhoR, rhoB, phi, fT, M, Minv, S);
}
// A:1169-1267 calRKCollision23GPUNew: gradient from the neighbours' densities (every non-fluid neighbour carries
// solidPhi), perturbation (AkR + AkB)/2 |G| (w (e.G)^2/|G|^2 - B_i) added to fT, recolouring of fT into fR, fB
__global__ __launch_bounds__(NB) void k_rk_pert_collide23(i64 N, double beta, double AkR, double AkB, double solidPhi, const i64 *nbr, const double *Bc,
                                                          const double *w, const double *rhoR, const double *rhoB, double *fR, double *fB, double *fT)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double t[9], r[9], b[9];
    tile_in<9>(lds, fT, n0, N, t);
    if (on) {
        double gx = 0., gy = 0.;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double ph = (q[i] != -1) ? (rhoR[q[i]] - rhoB[q[i]]) / (rhoR[q[i]] + rhoB[q[i]]) : solidPhi;
            gx += 3. * w[i + 1] * EX[i + 1] * ph;
            gy += 3. * w[i + 1] * EY[i + 1] * ph;
        }
        const double g2 = gx * gx + gy * gy, gn = sqrt(g2);
        const double rR = rhoR[n], rB = rhoB[n], rs = rR + rB, rm = rR * rB, rs2 = rs * rs;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double c2 = 0.;
            if (g2 != 0.) {
                const double part = w[i] * ((EX[i] * gx + EY[i] * gy) * (EX[i] * gx + EY[i] * gy)) / g2;
                c2 = (AkR + AkB) * 0.5 * gn * (part - Bc[i]);
            }
            t[i] += c2;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double en = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            double c = 0.;
            if (!(en == 0. || gn == 0.)) c = (EX[i] * gx + EY[i] * gy) / (sqrt(EX[i] * EX[i] + EY[i] * EY[i]) * gn);
            r[i] = rR / rs * t[i] + (beta * rm / rs2) * w[i] * c;
            b[i] = rB / rs * t[i] - (beta * rm / rs2) * w[i] * c;
        }
    }
    tile_out<9>(lds, fT, n0, N, t);
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_pert_collide23(hipStream_t st, i64 N, double beta, double AkR, double AkB, double solidPhi, const i64 *nbr, const double *Bc,
                                            const double *w, const double *rhoR, const double *rhoB, double *fR, double *fB, double *fT)
{
    if (N > 0) k_rk_pert_collide23<<<GRID_NODES(N)>>>(N, beta, AkR, AkB, solidPhi, nbr, Bc, w, rhoR, rhoB, fR, fB, fT);
}

// A:340-403 calStreaming1GPU as a pull (header of this file): fNew[n][i] = f[upstream(n, i)][i], or f[n][opp i]
// where the upstream node is not fluid; direction 0 of fNew is never written (as in the reference)
template <int Q, int QN>
__device__ __forceinline__ void pull_tile(double *lds, i64 N, const i64 *nbr, const double *f, double *fNew, const int *opp)
{
    THIS_NODE;
    i64 q[QN];
    tile_in<QN>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double r[Q];
    r[0] = 0.;
    if (on) {
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            const i64 up = q[opp[i] - 1];
            r[i] = up >= 0 ? f[Q * up + i] : f[Q * n + opp[i]];
        }
    }
    tile_out<Q, true>(lds, fNew, n0, N, r);
}
__global__ __launch_bounds__(NB) void k_rk_stream1(i64 N, const i64 *nbr, const double *f, double *fNew)
{
    __shared__ double lds[NB * 9];
    pull_tile<9, 8>(lds, N, nbr, f, fNew, OPP);
}
static inline void launch_rk_stream1(hipStream_t st, i64 N, const i64 *nbr, const double *f, double *fNew)
{
    if (N > 0) k_rk_stream1<<<GRID_NODES(N)>>>(N, nbr, f, fNew);
}
// A:409-417 calStreaming2GPU: copy back directions 1..8 (0 is never copied); flat
template <int Q>
__global__ void k_copy_skip0(i64 cnt, const double *src, double *dst)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt && k % Q != 0) dst[k] = src[k];
}
static inline void launch_rk_stream2(hipStream_t st, i64 N, const double *fNew, double *f)
{
    if (N > 0) k_copy_skip0<9><<<GRID_FLAT(9 * N)>>>(9 * N, fNew, f);
}

// ------------------------------------------------------------------ boundary rows: nx threads, not N
// A:2348-2412 constantTotalVelocityInlet (row ny-2): non-equilibrium bounce-back on f_tot, split by colour ratio;
// ratioB is evaluated AFTER rhoR was overwritten (reference quirk, replicated)
__global__ void k_rk_inlet_velocity_total(i64 N, i64 nx, i64 ny, double vyIn, const i64 *fluidNodes, double *rhoR, double *rhoB, double *fR, double *fB,
                                          double *fT, double *vy)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    double *t = fT + 9 * n;
    const double v = vyIn;
    const double rho = (t[0] + t[1] + t[3] + 2. * (t[2] + t[5] + t[6])) / (1. + v);
    const double eq2 = rho * 1. / 9. * (1. + 3. * (1. * v) + 4.5 * (0. + 1. * v) * (0. + 1. * v) - 1.5 * (v * v));
    const double eq4 = rho * 1. / 9. * (1. + 3. * (-1. * v) + 4.5 * (0. + (-1.) * v) * (0. + (-1.) * v) - 1.5 * (v * v));
    t[4] = eq4 + (t[2] - eq2);
    const double eq5 = rho * 1. / 36. * (1. + 3. * (1. * v + 1. * 0.) + 4.5 * (1. * v + 1. * 0.) * (1. * v + 1. * 0.) - 1.5 * (v * v));
    const double eq7 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (-1.) * 0.) + 4.5 * ((-1.) * v + (-1.) * 0.) * ((-1.) * v + (-1.) * 0.) - 1.5 * (v * v));
    t[7] = eq7 + (t[5] - eq5);
    const double eq6 = rho * 1. / 36. * (1. + 3. * ((1.) * v + (-1.) * 0.) + 4.5 * ((1.) * v + (-1.) * 0.) * (1. * v + (-1.) * 0.) - 1.5 * (v * v));
    const double eq8 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (1.) * 0.) + 4.5 * ((-1.) * v + 1. * 0.) * ((-1.) * v + 1. * 0.) - 1.5 * (v * v));
    t[8] = eq8 + (t[6] - eq6);
    const double ratioR = rhoR[n] / (rhoR[n] + rhoB[n]);
    rhoR[n] = ratioR * rho;
    fR[9 * n + 4] = ratioR * t[4]; fR[9 * n + 7] = ratioR * t[7]; fR[9 * n + 8] = ratioR * t[8];
    const double ratioB = rhoB[n] / (rhoR[n] + rhoB[n]);
    rhoB[n] = ratioB * rho;
    fB[9 * n + 4] = ratioB * t[4]; fB[9 * n + 7] = ratioB * t[7]; fB[9 * n + 8] = ratioB * t[8];
    vy[n] = v;
}
static inline void launch_rk_inlet_velocity_total(hipStream_t st, i64 N, i64 nx, i64 ny, double vyIn, const i64 *fluidNodes, double *rhoR, double *rhoB,
                                                  double *fR, double *fB, double *fT, double *vy)
{
    if (N > 0) k_rk_inlet_velocity_total<<<GRID_ROW(nx)>>>(N, nx, ny, vyIn, fluidNodes, rhoR, rhoB, fR, fB, fT, vy);
}
// A:657-695 constantVelocityZHBoundaryHigherRK: Zou-He velocity inlet per colour, row ny-2
__device__ __forceinline__ double zouhe_velocity_top(double *f, double v)
{
    const double rho = (f[0] + f[1] + f[3] + 2. * (f[2] + f[5] + f[6])) / (1. + v);
    f[4] = f[2] - 2. / 3. * rho * v;
    f[7] = f[5] + (f[1] - f[3]) / 2. - 1. / 6. * rho * v;
    f[8] = f[6] - (f[1] - f[3]) / 2. - 1. / 6. * rho * v;
    return rho;
}
__global__ void k_rk_pert_inlet_velocity(i64 N, i64 nx, i64 ny, double vyR, double vyB, const i64 *fluidNodes, double *rhoR, double *rhoB, double *fR,
                                         double *fB)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    rhoR[n] = zouhe_velocity_top(fR + 9 * n, vyR);
    rhoB[n] = zouhe_velocity_top(fB + 9 * n, vyB);
}
static inline void launch_rk_pert_inlet_velocity(hipStream_t st, i64 N, i64 nx, i64 ny, double vyR, double vyB, const i64 *fluidNodes, double *rhoR,
                                                 double *rhoB, double *fR, double *fB)
{
    if (N > 0) k_rk_pert_inlet_velocity<<<GRID_ROW(nx)>>>(N, nx, ny, vyR, vyB, fluidNodes, rhoR, rhoB, fR, fB);
}
// copy of a node's nine populations (both colours) from node `src`
__device__ __forceinline__ void copy_node2(double *fR, double *fB, i64 n, i64 src, double &sR, double &sB)
{
    sR = 0.; sB = 0.;
    for (int j = 0; j < 9; ++j) {
        fR[9 * n + j] = fR[9 * src + j];
        fB[9 * n + j] = fB[9 * src + j];
        sR += fR[9 * n + j]; sB += fB[9 * n + j];
    }
}
// A:607-650 ghostPointsConstantVelocityRK (row ny-1 <- its S neighbour, rho = sum)
__global__ void k_rk_ghost_inlet_velocity(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB, double *fR, double *fB)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 1);
    if (n < 0) return;
    const i64 L = nbr_node(nbr[8 * n + 3], N);
    double *r = fR + 9 * n, *b = fB + 9 * n;
    for (int i = 0; i < 9; ++i) r[i] = fR[9 * L + i];
    rhoR[n] = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8];
    for (int i = 0; i < 9; ++i) b[i] = fB[9 * L + i];
    rhoB[n] = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + b[7] + b[8];
}
static inline void launch_rk_ghost_inlet_velocity(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB,
                                                  double *fR, double *fB)
{
    if (N > 0) k_rk_ghost_inlet_velocity<<<GRID_ROW(nx)>>>(N, nx, ny, fluidNodes, nbr, rhoR, rhoB, fR, fB);
}
// A:925-962 calConstPressureInletGPU (row ny-2, Zou-He pressure per colour)
__device__ __forceinline__ void zouhe_pressure_top(double *f, double p)
{
    const double v = -1. + (f[0] + f[1] + f[3] + 2. * (f[2] + f[5] + f[6])) / p;
    f[4] = f[2] - 2. / 3. * p * v;
    f[7] = f[5] + 1. / 2. * (f[1] - f[3]) - 1. / 6. * p * v;
    f[8] = f[6] - 1. / 2. * (f[1] - f[3]) - 1. / 6. * p * v;
}
__global__ void k_rk_inlet_pressure(i64 N, i64 nx, i64 ny, double pB, double pR, const i64 *fluidNodes, double *rhoB, double *rhoR, double *fB, double *fR)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    zouhe_pressure_top(fB + 9 * n, pB); rhoB[n] = pB;
    zouhe_pressure_top(fR + 9 * n, pR); rhoR[n] = pR;
}
static inline void launch_rk_inlet_pressure(hipStream_t st, i64 N, i64 nx, i64 ny, double pB, double pR, const i64 *fluidNodes, double *rhoB, double *rhoR,
                                            double *fB, double *fR)
{
    if (N > 0) k_rk_inlet_pressure<<<GRID_ROW(nx)>>>(N, nx, ny, pB, pR, fluidNodes, rhoB, rhoR, fB, fR);
}
// A:968-1002 ghostPointsConstPressureInletRK (row ny-1 <- its S neighbour incl. rho)
__global__ void k_rk_ghost_inlet_pressure(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB, double *fR, double *fB)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 1);
    if (n < 0) return;
    const i64 H = nbr_node(nbr[8 * n + 3], N);
    for (int i = 0; i < 9; ++i) { fR[9 * n + i] = fR[9 * H + i]; fB[9 * n + i] = fB[9 * H + i]; }
    rhoR[n] = rhoR[H]; rhoB[n] = rhoB[H];
}
static inline void launch_rk_ghost_inlet_pressure(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB,
                                                  double *fR, double *fB)
{
    if (N > 0) k_rk_ghost_inlet_pressure<<<GRID_ROW(nx)>>>(N, nx, ny, fluidNodes, nbr, rhoR, rhoB, fR, fB);
}
// A:2560-2590 calConstPressureLowerGPUTotal (row 1, Zou-He pressure on f_tot)
__global__ void k_rk_outlet_pressure_total(i64 N, i64 nx, double pL, const i64 *fluidNodes, double *fT, double *vy, const double *rhoR, const double *rhoB,
                                           double *fR, double *fB)
{
    const i64 n = row_node(fluidNodes, N, nx, 1);
    if (n < 0) return;
    double *t = fT + 9 * n;
    const double v = 1. - 1. / pL * (t[0] + t[1] + t[3] + 2. * (t[4] + t[7] + t[8]));
    t[2] = t[4] + 2. / 3. * (pL * v);
    t[5] = t[7] + 0.5 * (t[3] - t[1]) + 1. / 6. * pL * v;
    t[6] = t[8] + 0.5 * (t[1] - t[3]) + 1. / 6. * pL * v;
    vy[n] = v;
    const double ratioR = rhoR[n] / (rhoR[n] + rhoB[n]);
    fR[9 * n + 2] = ratioR * t[2]; fR[9 * n + 5] = ratioR * t[5]; fR[9 * n + 6] = ratioR * t[6];
    const double ratioB = rhoB[n] / (rhoR[n] + rhoB[n]);
    fB[9 * n + 2] = ratioB * t[2]; fB[9 * n + 5] = ratioB * t[5]; fB[9 * n + 6] = ratioB * t[6];
}
static inline void launch_rk_outlet_pressure_total(hipStream_t st, i64 N, i64 nx, double pL, const i64 *fluidNodes, double *fT, double *vy, const double *rhoR,
                                                   const double *rhoB, double *fR, double *fB)
{
    if (N > 0) k_rk_outlet_pressure_total<<<GRID_ROW(nx)>>>(N, nx, pL, fluidNodes, fT, vy, rhoR, rhoB, fR, fB);
}
// Zou-He pressure outlet per colour, unknown populations 2, 5, 6
__device__ __forceinline__ void zouhe_pressure_bottom(double *f, double p, bool grouped)
{
    const double v = 1. - 1. / p * (f[0] + f[1] + f[3] + 2. * (f[4] + f[7] + f[8]));
    f[2] = grouped ? f[4] + 2. / 3. * (p * v) : f[4] + 2. / 3. * p * v;       // (the reference brackets the blue line only)
    f[5] = f[7] + 0.5 * (f[3] - f[1]) + 1. / 6. * p * v;
    f[6] = f[8] + 0.5 * (f[1] - f[3]) + 1. / 6. * p * v;
}
// A:1008-1039 calConstPressureLowerGPU: the row test is on the COMPACT index (nx <= n < 2 nx; grid row 1 when rows 0
// and 1 are all fluid); RKGPU2DBoundary.py:414-446 tests the GRID index instead (by_grid)
__global__ void k_rk_pert_outlet_pressure(i64 N, i64 nx, double pLB, double pLR, const i64 *fluidNodes, double *rhoB, double *rhoR, double *fB, double *fR,
                                          int by_grid)
{
    i64 n;
    if (by_grid) n = row_node(fluidNodes, N, nx, 1);
    else { n = nx + (i64)blockIdx.x * blockDim.x + threadIdx.x; if (n >= 2 * nx || n >= N) n = -1; }
    if (n < 0) return;
    zouhe_pressure_bottom(fB + 9 * n, pLB, true); rhoB[n] = pLB;
    zouhe_pressure_bottom(fR + 9 * n, pLR, false); rhoR[n] = pLR;
}
static inline void launch_rk_pert_outlet_pressure(hipStream_t st, i64 N, i64 nx, double pLB, double pLR, const i64 *fluidNodes, double *rhoB, double *rhoR,
                                                  double *fB, double *fR, int by_grid)
{
    if (N > 0) k_rk_pert_outlet_pressure<<<GRID_ROW(nx)>>>(N, nx, pLB, pLR, fluidNodes, rhoB, rhoR, fB, fR, by_grid);
}
// A:1045-1081 ghostPointsConstPressureLowerRK: acts on COMPACT indices < nx (grid row 0 when row 0 is all fluid);
// RKGPU2DBoundary.py:452-490 on grid row 0 (by_grid)
__global__ void k_rk_ghost_outlet_pressure(i64 N, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB, double *fR, double *fB, int by_grid)
{
    i64 n;
    if (by_grid) n = row_node(fluidNodes, N, nx, 0);
    else { n = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (n >= nx || n >= N) n = -1; }
    if (n < 0) return;
    const i64 L = nbr_node(nbr[8 * n + 1], N);
    for (int i = 0; i < 9; ++i) { fR[9 * n + i] = fR[9 * L + i]; fB[9 * n + i] = fB[9 * L + i]; }
    rhoR[n] = rhoR[L]; rhoB[n] = rhoB[L];
}
static inline void launch_rk_ghost_outlet_pressure(hipStream_t st, i64 N, i64 nx, const i64 *nbr, double *rhoR, double *rhoB, double *fR, double *fB)
{
    if (N > 0) k_rk_ghost_outlet_pressure<<<GRID_ROW(nx)>>>(N, nx, nullptr, nbr, rhoR, rhoB, fR, fB, 0);
}
static inline void launch_rk_ghost_outlet_pressure_grid(hipStream_t st, i64 N, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *rhoB,
                                                        double *fR, double *fB)
{
    if (N > 0) k_rk_ghost_outlet_pressure<<<GRID_ROW(nx)>>>(N, nx, fluidNodes, nbr, rhoR, rhoB, fR, fB, 1);
}
// A:700-784 convectiveOutletGPU / Ghost2GPU / Ghost3GPU: row r <- its N neighbour, rho re-summed
__global__ void k_rk_outlet_convective_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *fR, double *fB, double *rhoR, double *rhoB)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    double r, b;
    copy_node2(fR, fB, n, nbr_node(nbr[8 * n + 1], N), r, b);
    rhoR[n] = r; rhoB[n] = b;
}
static inline void launch_rk_outlet_convective_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *fR, double *fB,
                                                   double *rhoR, double *rhoB)
{
    if (N > 0) k_rk_outlet_convective_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, nbr, fR, fB, rhoR, rhoB);
}

// ---- kernels of RKCG2D/RKGPU2DBoundary.py ("B:") that its twin AcceleratedRKGPU2D.py does not use in a working loop
// B:222-320 (= A:791-880) convectiveAverageBoundaryGPU / 2 / 3: rows 2, 1, 0 relax towards their N neighbour with the
// |normal velocity| of the row-3 node of the column; the row's own normalVelocity entry takes that value
__global__ void k_rk_outlet_average_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *vn, double *fR, double *fB,
                                        const double *fROld, const double *fBOld)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q1 = nbr_node(nbr[8 * n + 1], N);
    i64 q = q1;                                  // the node on row 3 of this column
    for (i64 h = row; h < 2; ++h) q = nbr_node(nbr[8 * q + 1], N);
    const double v = fabs(vn[q]);
    for (int j = 0; j < 9; ++j) {
        fR[9 * n + j] = (fROld[9 * n + j] + v * fR[9 * q1 + j]) / (1. + v);
        fB[9 * n + j] = (fBOld[9 * n + j] + v * fB[9 * q1 + j]) / (1. + v);
    }
    vn[n] = vn[q];
}
static inline void launch_rk_outlet_average_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *vn, double *fR, double *fB,
                                                const double *fROld, const double *fBOld)
{
    if (N > 0) k_rk_outlet_average_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, nbr, vn, fR, fB, fROld, fBOld);
}
// B:496-530 (= A:1087-1120) calConstPressureHighGPU: Zou-He pressure on the TOP row ny-1, densities not updated;
// the tangential correction carries the sign the reference wrote (-(f3 - f1)/2 for 7, -(f1 - f3)/2 for 8)
__device__ __forceinline__ void zouhe_pressure_high(double *f, double p)
{
    const double v = -1. + 1. / p * (f[0] + f[1] + f[3] + 2. * (f[2] + f[5] + f[6]));
    f[4] = f[2] - 2. / 3. * p * v;
    f[7] = f[5] - 0.5 * (f[3] - f[1]) - 1. / 6. * p * v;
    f[8] = f[6] - 0.5 * (f[1] - f[3]) - 1. / 6. * p * v;
}
__global__ void k_rk_pressure_high(i64 N, i64 nx, i64 ny, double pB, double pR, const i64 *fluidNodes, double *fB, double *fR)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 1);
    if (n < 0) return;
    zouhe_pressure_high(fB + 9 * n, pB);
    zouhe_pressure_high(fR + 9 * n, pR);
}
static inline void launch_rk_pressure_high(hipStream_t st, i64 N, i64 nx, i64 ny, double pB, double pR, const i64 *fluidNodes, double *fB, double *fR)
{
    if (N > 0) k_rk_pressure_high<<<GRID_ROW(nx)>>>(N, nx, ny, pB, pR, fluidNodes, fB, fR);
}
// B:535-575 constantVelocityZHBoundaryHigherNewRK: Zou-He velocity for red on row ny-2; the retreating blue fluid takes
// its three unknown populations from the row above (4 from the N neighbour's 2; 7 / 8 from that node's E / W
// neighbours' 5 / 6 where those are fluid).  (A:2307-2345 has the blue part commented out: `blue` = 0.)
__global__ void k_rk_inlet_velocity_red(i64 N, i64 nx, i64 ny, double vyR, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *fR, double *fB, int blue)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    rhoR[n] = zouhe_velocity_top(fR + 9 * n, vyR);
    if (!blue) return;
    const i64 up = nbr_node(nbr[8 * n + 1], N);
    fB[9 * n + 4] = fB[9 * up + 2];
    const i64 q7 = nbr[8 * up], q8 = nbr[8 * up + 2];
    if (q7 >= 0) fB[9 * n + 7] = fB[9 * q7 + 5];
    if (q8 >= 0) fB[9 * n + 8] = fB[9 * q8 + 6];
}
static inline void launch_rk_inlet_velocity_red(hipStream_t st, i64 N, i64 nx, i64 ny, double vyR, const i64 *fluidNodes, const i64 *nbr, double *rhoR, double *fR,
                                                double *fB, int blue)
{
    if (N > 0) k_rk_inlet_velocity_red<<<GRID_ROW(nx)>>>(N, nx, ny, vyR, fluidNodes, nbr, rhoR, fR, fB, blue);
}

#include "sparse_sc_tr.h"
#include "sparse_rest_rk.h"
#include "sparse_rest_sc.h"
#include "sparse_rest_tr.h"
#include "dense_ef.h"

#define sc_check_nf(nf) do { if ((nf) != 2) { set_error("numFluids must be 2 (got %lld)", (long long)(nf)); return LBMPM_ERR_UNSUPPORTED; } } while (0)
#define tr_check_q5(q) do { if ((q) != 5) { set_error("numSchemes must be 5 (D2Q5); got %lld", (long long)(q)); return LBMPM_ERR_UNSUPPORTED; } } while (0)

}  // namespace

#include "sparse_entry_gen.h"

// ---- device-memory facade for the numba.cuda-shaped shim
extern "C" int lbmpm_device_malloc(int64_t bytes, void **out)
{
    LBMPM_REQUIRE(out && bytes >= 0, "lbmpm_device_malloc: bad argument");
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 8));
    if (e != hipSuccess) { set_error("hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e)); return LBMPM_ERR_NOMEM; }
    *out = p;
    return LBMPM_OK;
}
extern "C" int lbmpm_device_free(void *ptr)
{
    lbmpm::forget_lattice_constant(ptr);          // the address may come back holding something else
    if (ptr) LBMPM_HIP_TRY(hipFree(ptr));
    return LBMPM_OK;
}
extern "C" int lbmpm_memcpy_h2d(void *dst, const void *src, int64_t bytes)
{
    lbmpm::forget_lattice_constant(dst);
    LBMPM_HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return LBMPM_OK;
}
extern "C" int lbmpm_memcpy_d2h(void *dst, const void *src, int64_t bytes)
{
    LBMPM_HIP_TRY(hipDeviceSynchronize());
    LBMPM_HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return LBMPM_OK;
}
extern "C" int lbmpm_device_synchronize(void) { LBMPM_HIP_TRY(hipDeviceSynchronize()); return LBMPM_OK; }
