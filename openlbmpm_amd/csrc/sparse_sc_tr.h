// sparse_sc_tr.h -- second half of sparse_kernels.hip (included inside its anonymous namespace): the Shan-Chen /
// explicit-forcing kernels (ShanChen2D/OptimizedD2Q9GPU.py "O:", ExplicitD2Q9GPU.py "E:") on f[2][N][9], and the
// D2Q5 tracer kernels (RKCG2D/AccelerateTransport2DRK.py "T:") on g[nT][N][5].  Same tile / pull / flat / row
// patterns as the first half.
#pragma once

// component k of a [2][N][...] array
#define COMP(a, k, stride) ((a) + (size_t)(k) * N * (stride))

// =========================================================================== Shan-Chen / EFS
// O:84-94 calFluidRhoGPU
__global__ __launch_bounds__(NB) void k_sc_rho(i64 N, double *rho, const double *f)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    for (int k = 0; k < NF; ++k) {
        double g[9];
        tile_in<9>(lds, COMP(f, k, 9), n0, N, g);
        if (on) {
            double r = 0.;
#pragma unroll
            for (int j = 0; j < 9; ++j) r += g[j];
            COMP(rho, k, 1)[n] = r;
        }
    }
}
static inline void launch_sc_rho(hipStream_t st, i64 N, double *rho, const double *f)
{
    if (N > 0) k_sc_rho<<<GRID_NODES(N)>>>(N, rho, f);
}

// O:336-360 calMacroWholeVelocity: u' = sum_k (sum e f_k)/tau_k / sum_k rho_k/tau_k
__device__ __forceinline__ void whole_velocity(i64 N, i64 n, const double g[NF][9], const double *tau, const double *rho, double &pvx, double &pvy)
{
    double vxt = 0., vyt = 0., rt = 0.;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        vxt += (g[k][1] - g[k][3] + g[k][5] - g[k][6] - g[k][7] + g[k][8]) / tau[k];
        vyt += (g[k][2] - g[k][4] + g[k][5] + g[k][6] - g[k][7] - g[k][8]) / tau[k];
        rt += COMP(rho, k, 1)[n] / tau[k];
    }
    pvx = vxt / rt; pvy = vyt / rt;
}
__global__ __launch_bounds__(NB) void k_sc_macro_whole_velocity(i64 N, const double *tau, const double *rho, const double *f, double *pvx, double *pvy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (!on) return;
    double a, b;
    whole_velocity(N, n, g, tau, rho, a, b);
    pvx[n] = a; pvy[n] = b;
}
static inline void launch_sc_macro_whole_velocity(hipStream_t st, i64 N, const double *tau, const double *rho, const double *f, double *pvx, double *pvy)
{
    if (N > 0) k_sc_macro_whole_velocity<<<GRID_NODES(N)>>>(N, tau, rho, f, pvx, pvy);
}
// O:156-180 calPhysicalVelocity: u = sum_k (sum e f_k + F_k/2) / sum_k rho_k
__global__ __launch_bounds__(NB) void k_sc_physical_velocity(i64 N, const double *f, const double *rho, const double *Fx, const double *Fy, double *vx, double *vy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (!on) return;
    double tx = 0., ty = 0., tr = 0.;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        tx += (g[k][1] - g[k][3] + g[k][5] - g[k][6] - g[k][7] + g[k][8] + 1. / 2. * COMP(Fx, k, 1)[n]);
        ty += (g[k][2] - g[k][4] + g[k][5] + g[k][6] - g[k][7] - g[k][8] + 1. / 2. * COMP(Fy, k, 1)[n]);
        tr += COMP(rho, k, 1)[n];
    }
    vx[n] = tx / tr;
    vy[n] = ty / tr;
}
static inline void launch_sc_physical_velocity(hipStream_t st, i64 N, const double *f, const double *rho, const double *Fx, const double *Fy, double *vx, double *vy)
{
    if (N > 0) k_sc_physical_velocity<<<GRID_NODES(N)>>>(N, f, rho, Fx, Fy, vx, vy);
}

// O:452-534 / O:539-550 streaming of both components (pull + flat copy, see the header of sparse_kernels.hip; the
// Shan-Chen tables mark every non-fluid neighbour -1)
__global__ __launch_bounds__(NB) void k_sc_stream1(i64 N, const i64 *nbr, const double *f, double *fNew)
{
    __shared__ double lds[NB * 9];
    pull_tile<9, 8>(lds, N, nbr, COMP(f, blockIdx.y, 9), COMP(fNew, blockIdx.y, 9), OPP);
}
static inline void launch_sc_stream1(hipStream_t st, i64 N, const i64 *nbr, const double *f, double *fNew)
{
    if (N > 0) k_sc_stream1<<<dim3((unsigned)((N + NB - 1) / NB), NF), dim3(NB), 0, st>>>(N, nbr, f, fNew);
}
static inline void launch_sc_stream2(hipStream_t st, i64 N, const double *fNew, double *f)
{
    if (N > 0) k_copy_skip0<9><<<GRID_FLAT((i64)NF * 9 * N)>>>((i64)NF * 9 * N, fNew, f);
}

// ---- boundary rows (one thread per column of the row)
// O:839-866 / O:868-895 constantVelocityZouHeBoundaryHigher / ...Higher8: Zou-He velocity, rows ny-2 / ny-3
__global__ void k_sc_inlet_velocity_row(i64 N, i64 nx, i64 row, const double *vyIn, const i64 *fluidNodes, double *rho, double *f)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    for (int k = 0; k < NF; ++k) COMP(rho, k, 1)[n] = zouhe_velocity_top(COMP(f, k, 9) + 9 * n, vyIn[k]);
}
static inline void launch_sc_inlet_velocity_row(hipStream_t st, i64 N, i64 nx, i64 row, const double *vyIn, const i64 *fluidNodes, double *rho, double *f)
{
    if (N > 0) k_sc_inlet_velocity_row<<<GRID_ROW(nx)>>>(N, nx, row, vyIn, fluidNodes, rho, f);
}
// row <- the node `which` (3: S neighbour, 1: N neighbour) of the table, rho re-summed in the order 0..8
__global__ void k_sc_ghost_row(i64 N, i64 nx, i64 row, int which, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[8 * n + which], N);
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(f, k, 9) + 9 * n;
        const double *s = COMP(f, k, 9) + 9 * q;
        for (int j = 0; j < 9; ++j) g[j] = s[j];
        COMP(rho, k, 1)[n] = g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] + g[8];
    }
}
// O:710-738, O:897-955 ghostPointsConstantVelocityInlet / ...Velocity8 / ...82
static inline void launch_sc_ghost_inlet_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{
    if (N > 0) k_sc_ghost_row<<<GRID_ROW(nx)>>>(N, nx, row, 3, fluidNodes, nbr, rho, f);
}
// O:743-770, O:775-836 ghostPointsConstantPressureOutlet / ...Outlet8 / ...82
static inline void launch_sc_ghost_outlet_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{
    if (N > 0) k_sc_ghost_row<<<GRID_ROW(nx)>>>(N, nx, row, 1, fluidNodes, nbr, rho, f);
}
// O:555-585 / O:590-620 constantPressureZouHeBoundaryLower / ...Lower8: the outlet densities are the reference's
// hard-coded 1.0 / 0.02 (O:560-561), whatever densityL says
__global__ void k_sc_outlet_pressure_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, double *rho, double *f)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(f, k, 9) + 9 * n;
        const double d = k == 0 ? 1.0 : 0.02;
        const double v = 1. - (g[0] + g[1] + g[3] + 2. * (g[4] + g[7] + g[8])) / d;
        g[2] = g[4] + 2. / 3. * v * d;
        g[5] = g[7] + 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
        g[6] = g[8] - 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
        COMP(rho, k, 1)[n] = d;
    }
}
static inline void launch_sc_outlet_pressure_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, double *rho, double *f)
{
    if (N > 0) k_sc_outlet_pressure_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, rho, f);
}
// O:960-1041 convectiveOutletGPU / Ghost2GPU / Ghost3GPU: row <- its N neighbour, rho = sum of the SOURCE values
__global__ void k_sc_outlet_copy_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, double *rho)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[8 * n + 1], N);
    for (int k = 0; k < NF; ++k) {
        double r = 0.;
        for (int j = 0; j < 9; ++j) { const double v = COMP(f, k, 9)[9 * q + j]; COMP(f, k, 9)[9 * n + j] = v; r += v; }
        COMP(rho, k, 1)[n] = r;
    }
}
static inline void launch_sc_outlet_copy_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, double *rho)
{
    if (N > 0) k_sc_outlet_copy_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, nbr, f, rho);
}
// O:1044-1125 convectiveOutletEach{,2,3}GPU: f <- (f_old + |v_y(row 3)| f(N neighbour)) / (1 + |v_y(row 3)|)
__global__ void k_sc_outlet_convective_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, const double *fOld, double *rho, const double *vy)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q1 = nbr_node(nbr[8 * n + 1], N);
    i64 q = q1;                                  // the node on row 3 of this column
    for (i64 h = row; h < 2; ++h) q = nbr_node(nbr[8 * q + 1], N);
    const double v = fabs(vy[q]);
    for (int k = 0; k < NF; ++k) {
        double r = 0.;
        for (int j = 0; j < 9; ++j) {
            const double t = (COMP(fOld, k, 9)[9 * n + j] + v * COMP(f, k, 9)[9 * q1 + j]) / (1. + v);
            COMP(f, k, 9)[9 * n + j] = t;
            r += t;
        }
        COMP(rho, k, 1)[n] = r;
    }
}
static inline void launch_sc_outlet_convective_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, const double *fOld,
                                                   double *rho, const double *vy)
{
    if (N > 0) k_sc_outlet_convective_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, nbr, f, fOld, rho, vy);
}

// O:1274-1449 interactionCollisionProcess (original Shan-Chen: force from psi PRODUCTS, u_eq shift, BGK), fused in
// the reference too
__global__ __launch_bounds__(NB) void k_sc_interaction_collision(i64 N, const double *tau, const double *G, const double *Gs, const double *rho,
                                                                 const double *psi, double *f, const i64 *nbr, double *Fx, double *Fy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (on) {
        double pvx, pvy;
        whole_velocity(N, n, g, tau, rho, pvx, pvy);
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const double pk = COMP(psi, k, 1)[n];
            double fx = 0., fy = 0.;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double ex = EX[i + 1], ey = EY[i + 1], wi = i < 4 ? 1. / 9. : 1. / 36.;
                if (q[i] != -1) {
                    for (int j = 0; j < NF; ++j) {
                        const double pj = COMP(psi, j, 1)[q[i]];
                        if (ex != 0.) fx += -wi * G[k * NF + j] * pk * pj * (ex);
                        if (ey != 0.) fy += -wi * G[k * NF + j] * pk * pj * (ey);
                    }
                } else {
                    if (ex != 0.) fx += -wi * Gs[k] * pk * (ex);
                    if (ey != 0.) fy += -wi * Gs[k] * pk * (ey);
                }
            }
            COMP(Fx, k, 1)[n] = fx; COMP(Fy, k, 1)[n] = fy;
            const double rk = COMP(rho, k, 1)[n], tk = tau[k];
            const double ux = pvx + tk * fx / rk, uy = pvy + tk * fy / rk, usq = ux * ux + uy * uy;
            double *c = g[k];
            c[0] = (1 - 1. / tk) * c[0] + WT[0] * rk / tk * (1. - 1.5 * usq);
            c[1] = (1 - 1. / tk) * c[1] + WT[1] * rk / tk * (1. + 3. * ux + 4.5 * (ux * ux) - 1.5 * usq);
            c[2] = (1. - 1. / tk) * c[2] + WT[2] * rk / tk * (1. + 3. * uy + 4.5 * (uy * uy) - 1.5 * usq);
            c[3] = (1. - 1. / tk) * c[3] + WT[3] * rk / tk * (1. + 3. * (-ux) + 4.5 * ((-ux) * (-ux)) - 1.5 * usq);
            c[4] = (1. - 1. / tk) * c[4] + WT[4] * rk / tk * (1. + 3. * (-uy) + 4.5 * ((-uy) * (-uy)) - 1.5 * usq);
            double sq = (ux + uy) * (ux + uy);
            c[5] = (1. - 1. / tk) * c[5] + WT[5] * rk / tk * (1. + 3. * (ux + uy) + 4.5 * sq - 1.5 * usq);
            sq = (-ux + uy) * (-ux + uy);
            c[6] = (1. - 1. / tk) * c[6] + WT[6] * rk / tk * (1. + 3. * (-ux + uy) + 4.5 * sq - 1.5 * usq);
            sq = (-ux - uy) * (-ux - uy);
            c[7] = (1. - 1. / tk) * c[7] + WT[7] * rk / tk * (1. + 3. * (-ux - uy) + 4.5 * sq - 1.5 * usq);
            sq = (ux - uy) * (ux - uy);
            c[8] = (1. - 1. / tk) * c[8] + WT[8] * rk / tk * (1. + 3. * (ux - uy) + 4.5 * sq - 1.5 * usq);
        }
    }
    for (int k = 0; k < NF; ++k) tile_out<9>(lds, COMP(f, k, 9), n0, N, g[k]);
}
static inline void launch_sc_interaction_collision(hipStream_t st, i64 N, const double *tau, const double *G, const double *Gs, const double *rho,
                                                   const double *psi, double *f, const i64 *nbr, double *Fx, double *Fy)
{
    if (N > 0) k_sc_interaction_collision<<<GRID_NODES(N)>>>(N, tau, G, Gs, rho, psi, f, nbr, Fx, Fy);
}

// E:51-220 calExplicit4thOrderScheme: F_k = -6 psi_k sum_j G_kj sum_i w_i (psi_j(x+e_i) - psi_j(x)) e_i + wall term
__global__ __launch_bounds__(NB) void k_sc_efs_force4(i64 N, const i64 *nbr, const double *G, const double *Gs, const double *psi, double *Fx, double *Fy)
{
    __shared__ i64 lds[NB * 8];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(lds, nbr, n0, N, q);
    if (!on) return;
    double p0[NF];
    for (int j = 0; j < NF; ++j) p0[j] = COMP(psi, j, 1)[n];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        double gxs = 0., gys = 0., sx = 0., sy = 0.;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double ex = EX[i + 1], ey = EY[i + 1], wi = i < 4 ? 1. / 3. : 1. / 12.;
            if (q[i] != -1) {
                for (int j = 0; j < NF; ++j) {
                    const double d = COMP(psi, j, 1)[q[i]] - p0[j];
                    if (ex != 0.) gxs += wi * (d) * (ex) * G[k * NF + j];
                    if (ey != 0.) gys += wi * (d) * (ey) * G[k * NF + j];
                }
            } else {
                if (ex != 0.) sx += -wi * Gs[k] * p0[k] * (ex);
                if (ey != 0.) sy += -wi * Gs[k] * p0[k] * (ey);
            }
        }
        double fx = 0., fy = 0.;
        fx += -6.0 * p0[k] * gxs;
        fy += -6.0 * p0[k] * gys;
        fx += sx;
        fy += sy;
        COMP(Fx, k, 1)[n] = fx; COMP(Fy, k, 1)[n] = fy;
    }
}
static inline void launch_sc_efs_force4(hipStream_t st, i64 N, const i64 *nbr, const double *G, const double *Gs, const double *psi, double *Fx, double *Fy)
{
    if (N > 0) k_sc_efs_force4<<<GRID_NODES(N)>>>(N, nbr, G, Gs, psi, Fx, Fy);
}

// E:392-486 / E:488-625 fillNeighboringNodesISO8 / ISO10: 24 / 36 neighbours, periodic; one thread per table entry
__global__ void k_sc_fill_neighbors_iso(i64 N, i64 nx, i64 ny, int nn, const i64 *fluidNodes, const i64 *newidx, i64 *out)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nn * N) return;
    const i64 n = e / nn;
    const int m = (int)(e % nn);
    const i64 i = fluidNodes[n] / nx, j = fluidNodes[n] % nx;
    const i64 ii = (i + ISO_DY[m] + ny) % ny, jj = (j + ISO_DX[m] + nx) % nx;
    out[e] = newidx[ii * nx + jj];
}
static inline void launch_sc_fill_neighbors_iso(hipStream_t st, i64 N, i64 nx, i64 ny, int nn, const i64 *fluidNodes, const i64 *newidx, i64 *out)
{
    if (N > 0) k_sc_fill_neighbors_iso<<<GRID_FLAT((i64)nn * N)>>>(N, nx, ny, nn, fluidNodes, newidx, out);
}
// line-of-sight rule of the far neighbours (the `if` in front of every block of E:627-955 / :957-1377)
__device__ int iso_gate(const i64 *nb, int m)
{
    const int K2[8][2] = {{0, 4}, {1, 4}, {1, 5}, {2, 5}, {2, 6}, {3, 6}, {3, 7}, {0, 7}};
    const int K3[8][4] = {{4, 16, 0, 8}, {1, 9, 4, 17}, {1, 9, 5, 18}, {2, 10, 5, 19}, {2, 10, 6, 20}, {3, 11, 6, 21}, {3, 11, 7, 22}, {0, 8, 7, 23}};
    if (nb[m] == -1) return 0;
    if (m < 8) return 1;
    if (m < 16) return nb[m - 8] != -1;
    if (m < 24) return nb[K2[m - 16][0]] != -1 || nb[K2[m - 16][1]] != -1;
    if (m < 28) return nb[m - 24] != -1 && nb[m - 16] != -1;
    return (nb[K3[m - 28][0]] != -1 && nb[K3[m - 28][1]] != -1) || (nb[K3[m - 28][2]] != -1 && nb[K3[m - 28][3]] != -1);
}
// E:627-955 / E:957-1377 calExplicit8thOrderScheme / 10thOrderScheme (scheme 8 differences psi_j - psi_j(x), scheme 10
// the plain value, E:1009 ff.); the node's table row is staged in LDS (the gate rule reads it many times)
template <int NN>
__global__ __launch_bounds__(NB) void k_sc_efs_force_iso(i64 N, const i64 *nbrX, const double *w, const double *G, const double *Gs, const double *psi,
                                                         double *Fx, double *Fy)
{
    extern __shared__ i64 rowbuf[];                 // [NB][NN + 1]: odd row stride
    THIS_NODE;
    {
        const i64 base = n0 * NN, lim = N * NN;
        for (int e = threadIdx.x; e < NB * NN; e += NB)
            if (base + e < lim) rowbuf[(e / NN) * (NN + 1) + e % NN] = nbrX[base + e];
        __syncthreads();
    }
    if (!on) return;
    const i64 *nb = rowbuf + (size_t)threadIdx.x * (NN + 1);
    double fx[NF], fy[NF], p0[NF];
    for (int i = 0; i < NF; ++i) { fx[i] = 0.0; fy[i] = 0.0; p0[i] = COMP(psi, i, 1)[n]; }
    for (int m = 0; m < NN; ++m) {
        const i64 q = nb[m];
        const int dx = ISO_DX[m], dy = ISO_DY[m];
        if (iso_gate(nb, m)) {
            for (int i = 0; i < NF; ++i)
                for (int j = 0; j < NF; ++j) {
                    const double pq = COMP(psi, j, 1)[q];
                    const double d = NN == 36 ? pq : pq - p0[j], sx = dx > 0 ? 1. : -1., sy = dy > 0 ? 1. : -1.;
                    if (dx == 1 || dx == -1) fx[i] += -6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sx);
                    if (dx == 2 || dx == -2) fx[i] += -2. * 6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sx);
                    if (dx == 3 || dx == -3) fx[i] += -3. * 6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sx);
                    if (dy == 1 || dy == -1) fy[i] += -6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sy);
                    if (dy == 2 || dy == -2) fy[i] += -2. * 6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sy);
                    if (dy == 3 || dy == -3) fy[i] += -3. * 6.0 * w[m] * G[i * NF + j] * p0[i] * (d) * (sy);
                }
        } else if (m < 8 && q == -1) {
            for (int i = 0; i < NF; ++i) {
                const double c = m < 4 ? -1. / 9. : -1. / 36.;
                if (dx != 0) fx[i] += c * Gs[i] * p0[i] * (dx > 0 ? 1. : -1.);
                if (dy != 0) fy[i] += c * Gs[i] * p0[i] * (dy > 0 ? 1. : -1.);
            }
        }
    }
    for (int i = 0; i < NF; ++i) { COMP(Fx, i, 1)[n] = fx[i]; COMP(Fy, i, 1)[n] = fy[i]; }
}
static inline void launch_sc_efs_force_iso(hipStream_t st, i64 N, int nn, const i64 *nbrX, const double *w, const double *G, const double *Gs,
                                           const double *psi, double *Fx, double *Fy)
{
    if (N <= 0) return;
    const dim3 grid((unsigned)((N + NB - 1) / NB));
    if (nn == 36) k_sc_efs_force_iso<36><<<grid, dim3(NB), sizeof(i64) * NB * 37, st>>>(N, nbrX, w, G, Gs, psi, Fx, Fy);
    else k_sc_efs_force_iso<24><<<grid, dim3(NB), sizeof(i64) * NB * 25, st>>>(N, nbrX, w, G, Gs, psi, Fx, Fy);
}

// E:340-365 calEquilibriumVEFGPU (divide by tau_k) / E:1426-1452 transformEquilibriumVelocity (multiply by s_k)
__global__ __launch_bounds__(NB) void k_sc_efs_ueq(i64 N, const double *wk, int divide, const double *rho, const double *Fx, const double *Fy, const double *f,
                                                   double *ux, double *uy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (!on) return;
    double mx = 0., my = 0., rt = 0.;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        double ex = 0., ey = 0.;
#pragma unroll
        for (int j = 0; j < 9; ++j) { ex += g[k][j] * EX[j]; ey += g[k][j] * EY[j]; }
        ex += 1. / 2. * COMP(Fx, k, 1)[n];
        ey += 1. / 2. * COMP(Fy, k, 1)[n];
        if (divide) { mx += ex / wk[k]; my += ey / wk[k]; rt = rt + COMP(rho, k, 1)[n] / wk[k]; }
        else { mx += ex * wk[k]; my += ey * wk[k]; rt += COMP(rho, k, 1)[n] * wk[k]; }
    }
    ux[n] = mx / rt; uy[n] = my / rt;
}
static inline void launch_sc_efs_ueq(hipStream_t st, i64 N, const double *wk, int divide, const double *rho, const double *Fx, const double *Fy,
                                     const double *f, double *ux, double *uy)
{
    if (N > 0) k_sc_efs_ueq<<<GRID_NODES(N)>>>(N, wk, divide, rho, Fx, Fy, f, ux, uy);
}
// E:227-250 calEquilibriumFuncEFGPU: one thread per (component, node, direction) entry, unit-stride stores
__global__ void k_sc_efs_feq(i64 N, const double *rho, const double *ux, const double *uy, double *feq)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * 9 * N) return;
    const int j = (int)(e % 9);
    const i64 kn = e / 9, n = kn % N;
    const double vx = ux[n], vy = uy[n];
    feq[e] = WT[j] * rho[kn] * (1. + 3. * (EX[j] * vx + EY[j] * vy) + 9. / 2. * ((EX[j] * vx + EY[j] * vy) * (EX[j] * vx + EY[j] * vy)) -
                                 3. / 2. * (vx * vx + vy * vy));
}
static inline void launch_sc_efs_feq(hipStream_t st, i64 N, const double *rho, const double *ux, const double *uy, double *feq)
{
    if (N > 0) k_sc_efs_feq<<<GRID_FLAT((i64)NF * 9 * N)>>>(N, rho, ux, uy, feq);
}
// E:255-273 calForceDistrGPU: F_i = ((F.(e_i - u)) f_eq,i) / (rho / 3); flat
__global__ void k_sc_efs_fforce(i64 N, const double *ux, const double *uy, const double *rho, const double *Fx, const double *Fy, const double *feq, double *ff)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * 9 * N) return;
    const int j = (int)(e % 9);
    const i64 kn = e / 9, n = kn % N;
    ff[e] = ((Fx[kn] * (EX[j] - ux[n])) + (Fy[kn] * (EY[j] - uy[n]))) * feq[e] / (1. / 3. * rho[kn]);
}
static inline void launch_sc_efs_fforce(hipStream_t st, i64 N, const double *ux, const double *uy, const double *rho, const double *Fx, const double *Fy,
                                        const double *feq, double *ff)
{
    if (N > 0) k_sc_efs_fforce<<<GRID_FLAT((i64)NF * 9 * N)>>>(N, ux, uy, rho, Fx, Fy, feq, ff);
}
// E:278-289 transformPDFGPU: f <- f - F_i / 2; flat
__global__ void k_sc_efs_transform(i64 cnt, double *f, const double *ff)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < cnt) f[e] = f[e] - 1. / 2. * ff[e];
}
static inline void launch_sc_efs_transform(hipStream_t st, i64 N, double *f, const double *ff)
{
    if (N > 0) k_sc_efs_transform<<<GRID_FLAT((i64)NF * 9 * N)>>>((i64)NF * 9 * N, f, ff);
}
// E:294-306 calCollisionEXGPU; flat
__global__ void k_sc_efs_collide_srt(i64 N, const double *tau, double *f, const double *feq, const double *ff)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * 9 * N) return;
    const int k = (int)(e / (9 * N));
    f[e] = f[e] + 1. / tau[k] * (feq[e] - f[e] - 1. / 2. * ff[e]) + 1. * ff[e];
}
static inline void launch_sc_efs_collide_srt(hipStream_t st, i64 N, const double *tau, double *f, const double *feq, const double *ff)
{
    if (N > 0) k_sc_efs_collide_srt<<<GRID_FLAT((i64)NF * 9 * N)>>>(N, tau, f, feq, ff);
}

// E:1379-1399 transformPDFandEquil (f_eq overwritten by Lambda f_eq), E:1404-1420 transfromForceTerm,
// E:1457-1469 calAfterCollisionMRT: Lambda_k = M^-1 S_k M per component
__global__ __launch_bounds__(NB) void k_sc_mrt_transform_pdf_eq(i64 N, const double *f, double *feq, const double *Lam, double *fM)
{
    __shared__ double lds[NB * 9];
    THIS_NODE; (void)on; (void)n;
    const int k = blockIdx.y;
    double a[9], b[9], ta[9], tb[9];
    tile_in<9>(lds, COMP(f, k, 9), n0, N, a);
    tile_in<9>(lds, COMP(feq, k, 9), n0, N, b);
    mat9(Lam + (size_t)k * 81, a, ta);
    mat9(Lam + (size_t)k * 81, b, tb);
    tile_out<9>(lds, COMP(fM, k, 9), n0, N, ta);
    tile_out<9>(lds, COMP(feq, k, 9), n0, N, tb);
}
__global__ __launch_bounds__(NB) void k_sc_mrt_transform_force(i64 N, const double *ff, const double *Lam, double *ffM)
{
    __shared__ double lds[NB * 9];
    THIS_NODE; (void)on; (void)n;
    const int k = blockIdx.y;
    double a[9], ta[9];
    tile_in<9>(lds, COMP(ff, k, 9), n0, N, a);
    mat9(Lam + (size_t)k * 81, a, ta);
    tile_out<9>(lds, COMP(ffM, k, 9), n0, N, ta);
}
__global__ void k_sc_mrt_after_collision(i64 cnt, double *f, const double *ff, const double *feq, const double *fM, const double *ffM)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt) return;
    const double c = (feq[e] - fM[e] - 1. / 2. * ffM[e]);
    f[e] = f[e] + c + 1. * ff[e];
}
#define GRID_NODES_NF(N) dim3((unsigned)(((N) + NB - 1) / NB), NF), dim3(NB), 0, st
static inline void launch_sc_mrt_transform_pdf_eq(hipStream_t st, i64 N, const double *f, double *feq, const double *Lam, double *fM)
{
    if (N > 0) k_sc_mrt_transform_pdf_eq<<<GRID_NODES_NF(N)>>>(N, f, feq, Lam, fM);
}
static inline void launch_sc_mrt_transform_force(hipStream_t st, i64 N, const double *ff, const double *Lam, double *ffM)
{
    if (N > 0) k_sc_mrt_transform_force<<<GRID_NODES_NF(N)>>>(N, ff, Lam, ffM);
}
static inline void launch_sc_mrt_after_collision(hipStream_t st, i64 N, double *f, const double *ff, const double *feq, const double *fM, const double *ffM)
{
    if (N > 0) k_sc_mrt_after_collision<<<GRID_FLAT((i64)NF * 9 * N)>>>((i64)NF * 9 * N, f, ff, feq, fM, ffM);
}
// =========================================================================== D2Q5 tracers
// T:51-76 fillNeighboringNodesTransport: E, W, N, S
__global__ void k_tr_fill_neighbors(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *newIndex, i64 *nbr)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 4 * N) return;
    const i64 k = e >> 2;
    const int d = (int)(e & 3);
    const i64 loc = fluidNodes[k], i = loc / nx, j = loc % nx;
    const i64 jj = d == 0 ? (j < nx - 1 ? j + 1 : 0) : (d == 1 ? (j > 0 ? j - 1 : nx - 1) : j);
    const i64 ii = d == 2 ? (i < ny - 1 ? i + 1 : 0) : (d == 3 ? (i > 0 ? i - 1 : ny - 1) : i);
    nbr[e] = newIndex[ii * nx + jj];
}
static inline void launch_tr_fill_neighbors(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *newIndex, i64 *nbr)
{
    if (N > 0) k_tr_fill_neighbors<<<GRID_FLAT(4 * N)>>>(N, nx, ny, fluidNodes, newIndex, nbr);
}
#define GRID_NODES_T(N, nT) dim3((unsigned)(((N) + NB - 1) / NB), (unsigned)(nT)), dim3(NB), 0, st
// T:78-90 calConcentrationGPU
__global__ __launch_bounds__(NB) void k_tr_concentration(i64 N, double *C, const double *g)
{
    __shared__ double lds[NB * 5];
    THIS_NODE;
    const int t = blockIdx.y;
    double r[5];
    tile_in<5>(lds, COMP(g, t, 5), n0, N, r);
    if (!on) return;
    double c = 0.;
#pragma unroll
    for (int j = 0; j < 5; ++j) c += r[j];
    COMP(C, t, 1)[n] = c;
}
static inline void launch_tr_concentration(hipStream_t st, i64 N, int nT, double *C, const double *g)
{
    if (N > 0 && nT > 0) k_tr_concentration<<<GRID_NODES_T(N, nT)>>>(N, C, g);
}
// T:535-600 calCollisionTransportLinearEqlMRTGPU: g += A_t (M g - M g_eq), g_eq = C w (1 + 3 e.u)
__global__ __launch_bounds__(NB) void k_tr_collide_mrt(i64 N, const double *vx, const double *vy, const double *C, double *g, const double *M, const double *A)
{
    __shared__ double lds[NB * 5];
    THIS_NODE;
    const int t = blockIdx.y;
    double r[5];
    tile_in<5>(lds, COMP(g, t, 5), n0, N, r);
    if (on) {
        const double c = COMP(C, t, 1)[n], ux = vx[n], uy = vy[n];
        double eq[5], diff[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) eq[j] = c * WT5[j] * (1. + 3. * (VX[j] * ux + VY[j] * uy));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            double ve = 0., vp = 0.;
#pragma unroll
            for (int k = 0; k < 5; ++k) { ve += M[5 * j + k] * eq[k]; vp += r[k] * M[5 * j + k]; }
            diff[j] = vp - ve;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            double v = 0.;
#pragma unroll
            for (int k = 0; k < 5; ++k) v += A[(size_t)t * 25 + 5 * j + k] * diff[k];
            eq[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) r[j] = r[j] + eq[j];
    }
    tile_out<5>(lds, COMP(g, t, 5), n0, N, r);
}
static inline void launch_tr_collide_mrt(hipStream_t st, i64 N, int nT, const double *vx, const double *vy, const double *C, double *g, const double *M,
                                         const double *A)
{
    if (N > 0 && nT > 0) k_tr_collide_mrt<<<GRID_NODES_T(N, nT)>>>(N, vx, vy, C, g, M, A);
}
// T:957-971 calValueTransportDomain: -(1 - H(rhoR - crit))
__global__ void k_tr_indicator(i64 N, double crit, double *ind, const double *rhoR)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) ind[n] = (rhoR[n] > crit) ? -(1. - 1.) : -(1. - 0.);
}
static inline void launch_tr_indicator(hipStream_t st, i64 N, double crit, double *ind, const double *rhoR)
{
    if (N > 0) k_tr_indicator<<<GRID_FLAT(N)>>>(N, crit, ind, rhoR);
}
// T:976-1053 calTransportWithInterfaceD2Q5: anti-diffusive term across the interface
__global__ __launch_bounds__(NB) void k_tr_interface(i64 N, const double *beta, const double *ind, const double *Gx, const double *Gy, const double *C, double *g)
{
    __shared__ double lds[NB * 5];
    THIS_NODE;
    const int t = blockIdx.y;
    double r[5];
    tile_in<5>(lds, COMP(g, t, 5), n0, N, r);
    if (on) {
        const double gx = Gx[n], gy = Gy[n], gn = sqrt(gx * gx + gy * gy);
        double ux = 0., uy = 0., un = 0.;
        if (gn > 1.0e-8) { ux = -gx / gn; uy = -gy / gn; un = sqrt(ux * ux + uy * uy); }
        const double c0 = COMP(C, t, 1)[n];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double eq = WT5[j + 1] * c0;
            const double en = sqrt(VX[j + 1] * VX[j + 1] + VY[j + 1] * VY[j + 1]);
            double c = 0.;
            if (un > 1.0e-8 && en > 1.0e-8) c = (VX[j + 1] * ux + VY[j + 1] * uy) / (en * un);
            r[j + 1] = r[j + 1] + beta[t] * ind[n] * eq * c;
        }
    }
    tile_out<5>(lds, COMP(g, t, 5), n0, N, r);
}
static inline void launch_tr_interface(hipStream_t st, i64 N, int nT, const double *beta, const double *ind, const double *Gx, const double *Gy, const double *C,
                                       double *g)
{
    if (N > 0 && nT > 0) k_tr_interface<<<GRID_NODES_T(N, nT)>>>(N, beta, ind, Gx, Gy, C, g);
}
// T:139-182 / T:184-194 streaming (pull + flat copy)
__global__ __launch_bounds__(NB) void k_tr_stream1(i64 N, const i64 *nbr, const double *g, double *gNew)
{
    __shared__ double lds[NB * 5];
    pull_tile<5, 4>(lds, N, nbr, COMP(g, blockIdx.y, 5), COMP(gNew, blockIdx.y, 5), OPP5);
}
static inline void launch_tr_stream1(hipStream_t st, i64 N, int nT, const i64 *nbr, const double *g, double *gNew)
{
    if (N > 0 && nT > 0) k_tr_stream1<<<GRID_NODES_T(N, nT)>>>(N, nbr, g, gNew);
}
static inline void launch_tr_stream2(hipStream_t st, i64 N, int nT, const double *gNew, double *g)
{
    if (N > 0 && nT > 0) k_copy_skip0<5><<<GRID_FLAT((i64)nT * 5 * N)>>>((i64)nT * 5 * N, gNew, g);
}
// T:461-530 calFreeConcBoundary3: row 0 <- its N neighbour
__global__ void k_tr_free_outlet(i64 N, int nT, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *g)
{
    const i64 n = row_node(fluidNodes, N, nx, 0);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[4 * n + 2], N);
    for (int t = 0; t < nT; ++t)
        for (int j = 0; j < 5; ++j) COMP(g, t, 5)[5 * n + j] = COMP(g, t, 5)[5 * q + j];
}
static inline void launch_tr_free_outlet(hipStream_t st, i64 N, int nT, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *g)
{
    if (N > 0) k_tr_free_outlet<<<GRID_ROW(nx)>>>(N, nT, nx, fluidNodes, nbr, g);
}
// T:682-700 calInamuroConstConcBoundary: row ny-1, unknown population 4
__global__ void k_tr_inlet_inamuro(i64 N, int nT, i64 ny, i64 nx, const i64 *fluidNodes, const double *cb, double *g)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 1);
    if (n < 0) return;
    for (int t = 0; t < nT; ++t) {
        double *r = COMP(g, t, 5) + 5 * n;
        const double s = r[0] + r[1] + r[2] + r[3];
        const double u = (cb[t] - s) / WT5[4];
        r[4] = WT5[4] * u;
    }
}
static inline void launch_tr_inlet_inamuro(hipStream_t st, i64 N, int nT, i64 ny, i64 nx, const i64 *fluidNodes, const double *cb, double *g)
{
    if (N > 0) k_tr_inlet_inamuro<<<GRID_ROW(nx)>>>(N, nT, ny, nx, fluidNodes, cb, g);
}
// T:95-111 calReactionTracersGPU: A + B -> C at rate k C_A C_B; flat over (tracer, node, direction)
__global__ void k_tr_reaction(i64 N, int nT, const double *rate, const double *J, const double *C, double *g)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nT * 5 * N) return;
    const int j = (int)(e % 5);
    const i64 in = e / 5, n = in % N;
    const int i = (int)(in / N);
    const double s = rate[0] * C[0 * N + n] * C[1 * N + n];
    const double S = i == 2 ? s : (i < 2 ? -s : 0.);
    g[e] = g[e] + J[i * 5 + j] * S;
}
static inline void launch_tr_reaction(hipStream_t st, i64 N, int nT, const double *rate, const double *J, const double *C, double *g)
{
    if (N > 0 && nT > 0) k_tr_reaction<<<GRID_FLAT((i64)nT * 5 * N)>>>(N, nT, rate, J, C, g);
}
