// sparse_rest_tr.h -- the kernels of RKCG2D/AccelerateTransport2DRK.py ("T:") that the working loop
// (Transport2DRK.py:1177-1485, D2Q5 + MRT) does not launch: the BGK and quadratic-equilibrium collisions, the
// moving-interface bookkeeping of the earlier single-phase transport (new / old fluid node lists, boolean domain
// masks), the other outlet / inlet rows, and the D2Q9 tracer scheme.  g[nT][N][5|9], C[nT][N]; conventions of
// sparse_kernels.hip / sparse_sc_tr.h; statement order of the reference; pinned one by one
// (tests/golden/kats_tr.npz, tests/test_kats_gpu.py).  Included by sparse_kernels.hip.
//
// Two reference kernels cannot run as written and are exported as entry points that say so:
//   T:927 calUpdateConcInTransportDomainByVQ9 fills a 5-entry shared array with 9 values (T:938-939) and indexes the
//         9-entry direction table by NODE (T:953);
//   T:596 calCollisionTransportQuadraticEqlMRTGPU indexes the 5-entry direction table by node as well (T:624): defined
//         for the first five nodes only, which is what its entry point accepts.

typedef unsigned char u8;

// T:118-131 calCollisionTransportGPU: BGK towards C (J_j + e_j.u / 2)
__global__ void k_tr_collide_bgk(i64 N, int nT, const double *vx, const double *vy, const double *tau, const double *J, const double *C, double *g)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nT * 5 * N) return;
    const int j = (int)(e % 5);
    const i64 in = e / 5, n = in % N;
    const int i = (int)(in / N);
    const double eq = C[in] * (J[i * 5 + j] + 1. / 2. * (VX[j] * vx[n] + VY[j] * vy[n]));
    g[e] = g[e] - 1. / tau[i] * (g[e] - eq);
}
static inline void launch_tr_collide_bgk(hipStream_t st, i64 N, int nT, const double *vx, const double *vy, const double *tau, const double *J, const double *C,
                                         double *g)
{
    if (N > 0 && nT > 0) k_tr_collide_bgk<<<GRID_FLAT((i64)nT * 5 * N)>>>(N, nT, vx, vy, tau, J, C, g);
}

// T:197-208 calUpdateDistributionGPU: boolean mask of the nodes where the carrier fluid is absent
__global__ void k_tr_update_distribution(i64 N, double crit, const double *rhoR, u8 *field)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) field[n] = rhoR[n] < crit ? 1 : 0;
}
static inline void launch_tr_update_distribution(hipStream_t st, i64 N, double crit, const double *rhoR, u8 *field)
{
    if (N > 0) k_tr_update_distribution<<<GRID_FLAT(N)>>>(N, crit, rhoR, field);
}

__device__ __forceinline__ bool in_list(const i64 *list, i64 len, i64 v)
{
    for (i64 m = 0; m < len; ++m)
        if (list[m] == v) return true;
    return false;
}
// T:216-240 calUpdateConcOnNewNodesGPU (Kang et al. 2007): a node that has just become fluid takes the mean
// concentration of its surrounding nodes that are inside the mask and are not new themselves.  One thread per list
// entry (the reference scans the list in every thread); a node listed twice is computed twice, as there.
__global__ void k_tr_conc_on_new_nodes(i64 N, int nT, const i64 *newList, i64 len, const i64 *surrounding, double *C, const u8 *field)
{
    const i64 m = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= len) return;
    const i64 n = newList[m];
    if (n < 0 || n >= N) return;
    for (int i = 0; i < nT; ++i) {
        double total = 0.;
        i64 count = 0;
        for (int j = 0; j < 8; ++j) {
            const i64 s = surrounding[8 * n + j];
            if (s >= 0 && field[s] && !in_list(newList, len, s)) {
                total += COMP(C, i, 1)[s];
                count += 1;
            }
        }
        COMP(C, i, 1)[n] = total / (double)count;
    }
}
static inline void launch_tr_conc_on_new_nodes(hipStream_t st, i64 N, int nT, const i64 *newList, i64 len, const i64 *surrounding, double *C, const u8 *field)
{
    if (N > 0 && len > 0) k_tr_conc_on_new_nodes<<<GRID_FLAT(len)>>>(N, nT, newList, len, surrounding, C, field);
}

// T:245-258 calUpdateConcOnOldNodesGPU: concentration and populations of the listed nodes set to zero
__global__ void k_tr_clear_listed(i64 N, int nT, const i64 *list, i64 len, double *C, double *g)
{
    const i64 m = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= len) return;
    const i64 n = list[m];
    if (n < 0 || n >= N) return;
    for (int i = 0; i < nT; ++i) {
        COMP(C, i, 1)[n] = 0.;
        for (int j = 0; j < 5; ++j) COMP(g, i, 5)[5 * n + j] = 0.;
    }
}
static inline void launch_tr_clear_listed(hipStream_t st, i64 N, int nT, const i64 *list, i64 len, double *C, double *g)
{
    if (N > 0 && len > 0) k_tr_clear_listed<<<GRID_FLAT(len)>>>(N, nT, list, len, C, g);
}

// T:267-278 calUpdateConcOnAllNewNodesGPU: the same for every node outside the boolean transport domain
__global__ void k_tr_clear_outside(i64 N, int nT, const u8 *dom, double *C, double *g)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nT * 5 * N) return;
    const i64 in = e / 5, n = in % N;
    if (dom[n]) return;
    g[e] = 0.;
    if (e % 5 == 0) C[in] = 0.;
}
static inline void launch_tr_clear_outside(hipStream_t st, i64 N, int nT, const u8 *dom, double *C, double *g)
{
    if (N > 0 && nT > 0) k_tr_clear_outside<<<GRID_FLAT((i64)nT * 5 * N)>>>(N, nT, dom, C, g);
}

// T:285-300 calUpdateConcWholeDomainGPU: concentrations rescaled so that the tracer mass survives a move of the interface
__global__ void k_tr_rescale_whole_domain(i64 N, int nT, double pert, const double *sumOldConc, const double *sumOldList, const double *sumNewList, double *Cnew,
                                          const double *C)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nT * N) return;
    const int i = (int)(e / N);
    const double leave = 0.;
    Cnew[e] = (1. + pert) * C[e] * (sumOldConc[i] / (sumNewList[i] + sumOldConc[i] - sumOldList[i] + leave));
}
static inline void launch_tr_rescale_whole_domain(hipStream_t st, i64 N, int nT, double pert, const double *sumOldConc, const double *sumOldList,
                                                  const double *sumNewList, double *Cnew, const double *C)
{
    if (N > 0 && nT > 0) k_tr_rescale_whole_domain<<<GRID_FLAT((i64)nT * N)>>>(N, nT, pert, sumOldConc, sumOldList, sumNewList, Cnew, C);
}

// T:310-337 calTransportInterfaceGPU (Q = 5) / T:839-880 calTransportInterfaceQ9GPU (Q = 9): a masked node takes the
// population that points at it from every unmasked neighbour into its own opposite slot, and clears it there.  Each
// (neighbour, direction) entry is touched by one node only.
template <int Q>
__global__ void k_tr_interface_exchange(i64 N, int nT, const i64 *nbr, double *g, const u8 *field)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || field[n] != 1) return;
    for (int i = 0; i < nT; ++i) {
        double *gi = COMP(g, i, Q);
        for (int d = 1; d < Q; ++d) {
            const i64 q = nbr[(Q - 1) * n + d - 1];
            if (q >= 0 && field[q] == 0) {
                const int o = Q == 5 ? OPP5[d] : OPP[d];
                gi[Q * n + o] = gi[Q * q + d];
                gi[Q * q + d] = 0.;
            }
        }
    }
}
template <int Q>
static inline void launch_tr_interface_exchange(hipStream_t st, i64 N, int nT, const i64 *nbr, double *g, const u8 *field)
{
    if (N > 0) k_tr_interface_exchange<Q><<<GRID_FLAT(N)>>>(N, nT, nbr, g, field);
}

// T:389-411 calUpdatedPDFWithNewRho: for a masked node the reference walks the WHOLE list: an entry equal to the node
// resets its populations to the equilibrium of the new concentration, every other entry rescales them by
// (C_new - C)/C once more (the `else` belongs to the `if` inside the `for`, T:400-410); kept entry by entry
__global__ void k_tr_pdf_with_new_rho(i64 N, int nT, const i64 *list, i64 len, const double *vx, const double *vy, const double *C, const double *Cnew,
                                      const double *J, double *g, const u8 *field)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || field[n] != 1) return;
    for (i64 m = 0; m < len; ++m) {
        if (list[m] == n) {
            for (int i = 0; i < nT; ++i)
                for (int j = 0; j < 5; ++j)
                    COMP(g, i, 5)[5 * n + j] = COMP(Cnew, i, 1)[n] * (J[i * 5 + j] + 1. / 2. * (VX[j] * vx[n] + VY[j] * vy[n]));
        } else {
            for (int i = 0; i < nT; ++i) {
                const double diff = COMP(Cnew, i, 1)[n] - COMP(C, i, 1)[n];
                for (int j = 0; j < 5; ++j) {
                    const double ratio = COMP(g, i, 5)[5 * n + j] / COMP(C, i, 1)[n];
                    COMP(g, i, 5)[5 * n + j] = COMP(g, i, 5)[5 * n + j] + diff * ratio;
                }
            }
        }
    }
}
static inline void launch_tr_pdf_with_new_rho(hipStream_t st, i64 N, int nT, const i64 *list, i64 len, const double *vx, const double *vy, const double *C,
                                              const double *Cnew, const double *J, double *g, const u8 *field)
{
    if (N > 0) k_tr_pdf_with_new_rho<<<GRID_FLAT(N)>>>(N, nT, list, len, vx, vy, C, Cnew, J, g, field);
}

// T:419-432 / T:440-453 calFreeConcBoundary1 / 2: grid row 2 / 1 <- its N neighbour (row 0: calFreeConcBoundary3, sparse_sc_tr.h)
__global__ void k_tr_free_row(i64 N, int nT, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *g)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[4 * n + 2], N);
    for (int t = 0; t < nT; ++t)
        for (int j = 0; j < 5; ++j) COMP(g, t, 5)[5 * n + j] = COMP(g, t, 5)[5 * q + j];
}
static inline void launch_tr_free_row(hipStream_t st, i64 N, int nT, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *g)
{
    if (N > 0) k_tr_free_row<<<GRID_ROW(nx)>>>(N, nT, nx, row, fluidNodes, nbr, g);
}

// T:480-495 calZeroConcenBoundary: row ny-2 <- its S neighbour, concentration re-summed
__global__ void k_tr_zero_gradient_inlet(i64 N, int nT, i64 nx, i64 ny, const i64 *fluidNodes, double *C, double *g, const i64 *nbr)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[4 * n + 3], N);
    for (int t = 0; t < nT; ++t) {
        double c = 0.;
        for (int j = 0; j < 5; ++j) {
            COMP(g, t, 5)[5 * n + j] = COMP(g, t, 5)[5 * q + j];
            c += COMP(g, t, 5)[5 * n + j];
        }
        COMP(C, t, 1)[n] = c;
    }
}
static inline void launch_tr_zero_gradient_inlet(hipStream_t st, i64 N, int nT, i64 nx, i64 ny, const i64 *fluidNodes, double *C, double *g, const i64 *nbr)
{
    if (N > 0) k_tr_zero_gradient_inlet<<<GRID_ROW(nx)>>>(N, nT, nx, ny, fluidNodes, C, g, nbr);
}

// T:500-520 calUpdateConcInTransportDomainByV: inside the boolean domain, where the fluid moves, the concentration
// grows by the share totalOld/totalTracer and the populations restart from the linear equilibrium
__global__ void k_tr_conc_by_velocity(i64 N, int nT, const double *totalTracer, const double *totalOld, const u8 *dom, const double *vx, const double *vy,
                                      const double *w, double *C, double *g)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double vn = sqrt(vx[n] * vx[n] + vy[n] * vy[n]);
    if (!(dom[n] == 1 && vn > 1e-10)) return;
    for (int i = 0; i < nT; ++i) {
        const double extra = COMP(C, i, 1)[n] * totalOld[i] / totalTracer[i];
        COMP(C, i, 1)[n] += extra;
        for (int j = 0; j < 5; ++j)
            COMP(g, i, 5)[5 * n + j] = COMP(C, i, 1)[n] * w[j] * (1. + 3. * (VX[j] * vx[n] + VY[j] * vy[n]));
    }
}
static inline void launch_tr_conc_by_velocity(hipStream_t st, i64 N, int nT, const double *totalTracer, const double *totalOld, const u8 *dom, const double *vx,
                                              const double *vy, const double *w, double *C, double *g)
{
    if (N > 0) k_tr_conc_by_velocity<<<GRID_FLAT(N)>>>(N, nT, totalTracer, totalOld, dom, vx, vy, w, C, g);
}

// T:596-645 calCollisionTransportQuadraticEqlMRTGPU: MRT about a quadratic equilibrium in which u_x is a local 0.0
// (T:613) and one factor takes unitVY[node] (T:624) -- see the header; N <= 5 is checked by the entry point
__global__ void k_tr_collide_mrt_quadratic(i64 N, int nT, const double *vy, const double *C, double *g, const double *M, const double *A, const double *w)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double ux = 0.0;
    for (int i = 0; i < nT; ++i) {
        double *r = COMP(g, i, 5) + 5 * n;
        double eq[5], me[5], mp[5];
        for (int j = 0; j < 5; ++j)
            eq[j] = COMP(C, i, 1)[n] * w[j] * (1. + 3. * (VX[j] * ux + VY[j] * vy[n]) + 4.5 * (VX[j] * ux + VY[n] * vy[n]) * (VX[j] * ux + VY[j] * vy[n]) -
                                               1.5 * (ux * ux + vy[n] * vy[n]));
        for (int j = 0; j < 5; ++j) {
            double a = 0., b = 0.;
            for (int k = 0; k < 5; ++k) { a += M[5 * j + k] * eq[k]; b += r[k] * M[5 * j + k]; }
            me[j] = a; mp[j] = b;
        }
        for (int j = 0; j < 5; ++j) {
            double a = 0., b = 0.;
            for (int k = 0; k < 5; ++k) { a += A[25 * i + 5 * j + k] * me[k]; b += A[25 * i + 5 * j + k] * mp[k]; }
            r[j] = -a + b + r[j];
        }
    }
}
static inline void launch_tr_collide_mrt_quadratic(hipStream_t st, i64 N, int nT, const double *vy, const double *C, double *g, const double *M, const double *A,
                                                   const double *w)
{
    if (N > 0) k_tr_collide_mrt_quadratic<<<GRID_FLAT(N)>>>(N, nT, vy, C, g, M, A, w);
}

// T:661-676 calAntiCollisionConcBoundary: anti-bounce-back of population 3 of row ny-2 into population 4 of the row above
__global__ void k_tr_anti_bounce_inlet(i64 N, int nT, i64 ny, i64 nx, const i64 *fluidNodes, const i64 *nbr, const double *cb, const double *w, double *g)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    const i64 up = nbr_node(nbr[4 * n + 2], N);
    for (int t = 0; t < nT; ++t) COMP(g, t, 5)[5 * up + 4] = -COMP(g, t, 5)[5 * n + 3] + 2. * w[3] * cb[t];
}
static inline void launch_tr_anti_bounce_inlet(hipStream_t st, i64 N, int nT, i64 ny, i64 nx, const i64 *fluidNodes, const i64 *nbr, const double *cb,
                                               const double *w, double *g)
{
    if (N > 0) k_tr_anti_bounce_inlet<<<GRID_ROW(nx)>>>(N, nT, ny, nx, fluidNodes, nbr, cb, w, g);
}

// ------------------------------------------------------------------------------------------ D2Q9 tracer scheme
// T:704-722 calCollisionQ9: BGK towards C w (1 + 3 e.u); flat
__global__ void k_tr9_collide_bgk(i64 N, int nT, const double *vx, const double *vy, const double *tau, const double *C, double *g, const double *w)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)nT * 9 * N) return;
    const int j = (int)(e % 9);
    const i64 in = e / 9, n = in % N;
    const int i = (int)(in / N);
    const double eq = C[in] * w[j] * (1. + 3. * (EX[j] * vx[n] + EY[j] * vy[n]));
    g[e] = -(g[e] - eq) / tau[i] + g[e];
}
static inline void launch_tr9_collide_bgk(hipStream_t st, i64 N, int nT, const double *vx, const double *vy, const double *tau, const double *C, double *g,
                                          const double *w)
{
    if (N > 0 && nT > 0) k_tr9_collide_bgk<<<GRID_FLAT((i64)nT * 9 * N)>>>(N, nT, vx, vy, tau, C, g, w);
}

// T:1019-1046 calTransportWithInterfaceD2Q9: the interface term beta v w_j C cos(e_j, -G) on the eight moving populations
__global__ __launch_bounds__(NB) void k_tr9_interface(i64 N, const double *beta, const double *ind, const double *Gx, const double *Gy, const double *w,
                                                      const double *C, double *g)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    const int t = blockIdx.y;
    double r[9];
    tile_in<9>(lds, COMP(g, t, 9), n0, N, r);
    if (on) {
        const double gn = sqrt(Gx[n] * Gx[n] + Gy[n] * Gy[n]);
        double ux, uy, un;
        if (gn > 1.0e-8) { ux = -Gx[n] / gn; uy = -Gy[n] / gn; un = sqrt(ux * ux + uy * uy); }
        else { ux = 0.; uy = 0.; un = 0.; }
        const double c = COMP(C, t, 1)[n];
#pragma unroll
        for (int j = 1; j < 9; ++j) {
            const double eq = w[j] * c;
            const double en = sqrt(EX[j] * EX[j] + EY[j] * EY[j]);
            double cs;
            if (un > 1.0e-8 && en > 1.0e-8) cs = (EX[j] * ux + EY[j] * uy) / (en * un);
            else cs = 0.;
            r[j] = r[j] + beta[t] * ind[n] * eq * cs;
        }
    }
    tile_out<9>(lds, COMP(g, t, 9), n0, N, r);
}
static inline void launch_tr9_interface(hipStream_t st, i64 N, int nT, const double *beta, const double *ind, const double *Gx, const double *Gy, const double *w,
                                        const double *C, double *g)
{
    if (N > 0 && nT > 0) k_tr9_interface<<<GRID_NODES_T(N, nT)>>>(N, beta, ind, Gx, Gy, w, C, g);
}

// T:1053-1087 calCollisionTransportLinearEqlMRTGPUD2Q9: g += A_t (M g - M g_eq), g_eq = C w (1 + 3 e.u), 9 x 9
__global__ __launch_bounds__(NB) void k_tr9_collide_mrt(i64 N, const double *vx, const double *vy, const double *C, double *g, const double *M, const double *A,
                                                        const double *w)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    const int t = blockIdx.y;
    double r[9];
    tile_in<9>(lds, COMP(g, t, 9), n0, N, r);
    if (on) {
        double eq[9], df[9], im[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) eq[j] = COMP(C, t, 1)[n] * w[j] * (1. + 3. * (EX[j] * vx[n] + EY[j] * vy[n]));
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double a = 0., b = 0.;
#pragma unroll
            for (int k = 0; k < 9; ++k) { a += M[9 * j + k] * eq[k]; b += r[k] * M[9 * j + k]; }
            df[j] = b - a;
        }
        mat9(A + 81 * t, df, im);
#pragma unroll
        for (int j = 0; j < 9; ++j) r[j] = r[j] + im[j];
    }
    tile_out<9>(lds, COMP(g, t, 9), n0, N, r);
}
static inline void launch_tr9_collide_mrt(hipStream_t st, i64 N, int nT, const double *vx, const double *vy, const double *C, double *g, const double *M,
                                          const double *A, const double *w)
{
    if (N > 0 && nT > 0) k_tr9_collide_mrt<<<GRID_NODES_T(N, nT)>>>(N, vx, vy, C, g, M, A, w);
}

// T:736-817 / T:823-834 calStreaming1GPU / calStreaming2GPU of the tracer module: the D2Q9 streaming of O:452 / O:539
// for `numFluids` = number of tracers (pull + flat copy, as there)
__global__ __launch_bounds__(NB) void k_tr9_stream1(i64 N, const i64 *nbr, const double *g, double *gNew)
{
    __shared__ double lds[NB * 9];
    pull_tile<9, 8>(lds, N, nbr, COMP(g, blockIdx.y, 9), COMP(gNew, blockIdx.y, 9), OPP);
}
static inline void launch_tr9_stream1(hipStream_t st, i64 N, int nT, const i64 *nbr, const double *g, double *gNew)
{
    if (N > 0 && nT > 0) k_tr9_stream1<<<GRID_NODES_T(N, nT)>>>(N, nbr, g, gNew);
}
static inline void launch_tr9_stream2(hipStream_t st, i64 N, int nT, const double *gNew, double *g)
{
    if (N > 0 && nT > 0) k_copy_skip0<9><<<GRID_FLAT((i64)nT * 9 * N)>>>((i64)nT * 9 * N, gNew, g);
}
