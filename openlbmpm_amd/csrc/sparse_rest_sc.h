// sparse_rest_sc.h -- the kernels of ShanChen2D/OptimizedD2Q9GPU.py ("O:") and ExplicitD2Q9GPU.py ("E:") that no
// working loop of the reference launches: the unfused generation of the original Shan-Chen step (force, equilibrium
// velocity, equilibrium, BGK as separate kernels), the equation-of-state helpers, the pressure inlet, Chang's
// boundary rows, the exact-difference and Guo forcing variants, streaming with the moving-wall link term, the
// free-flow outlet of the explicit-forcing loop.  Two fluids (like every Shan-Chen entry point), arrays [2][N](...)
// as the reference lays them out.  Conventions of sparse_kernels.hip / sparse_sc_tr.h; statement order of the
// reference; pinned one by one (tests/golden/kats_sc.npz, tests/test_kats_gpu.py).  Included by sparse_kernels.hip.

// force of component k at node n from the PRODUCTS psi_k(x) psi_j(x + e_i) (O:186-313, O:1478-1570, O:1817-1909):
// fluid links weighted by weightInter[i], wall links by 1/9, 1/36
__device__ __forceinline__ void sc_product_force(i64 N, i64 n, int k, const i64 q[8], const double *wi, const double *G, const double *Gs, const double *psi,
                                                 double &fx, double &fy)
{
    const double pk = COMP(psi, k, 1)[n];
    fx = 0.; fy = 0.;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double ex = EX[i + 1], ey = EY[i + 1];
        if (q[i] != -1) {
            for (int j = 0; j < NF; ++j) {
                const double pj = COMP(psi, j, 1)[q[i]];
                if (ex != 0.) fx += -wi[i] * G[k * NF + j] * pk * pj * (ex);
                if (ey != 0.) fy += -wi[i] * G[k * NF + j] * pk * pj * (ey);
            }
        } else {
            const double ws = i < 4 ? 1. / 9. : 1. / 36.;
            if (ex != 0.) fx += -ws * Gs[k] * pk * (ex);
            if (ey != 0.) fy += -ws * Gs[k] * pk * (ey);
        }
    }
}

// O:112-128 calFluidPotentialGPUPR: psi = sqrt(2 (p_PR(rho) - rho/3) / (c0 g)), Peng-Robinson p (Yuan & Schaefer)
__global__ void k_sc_potential_pr(i64 cnt, double R, double T, double a, double b, double alpha, double c0, double g, const double *rho, double *psi)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt) return;
    const double r = rho[e];
    const double p = (r * R * T) / (1. - b * r) - (a * alpha * r * r) / (1. + 2. * b * r - b * b * r * r);
    psi[e] = sqrt(2. / (c0 * g) * (p - 1. / 3. * r));
}
static inline void launch_sc_potential_pr(hipStream_t st, i64 N, double R, double T, double a, double b, double alpha, double c0, double g, const double *rho,
                                          double *psi)
{
    if (N > 0) k_sc_potential_pr<<<GRID_FLAT((i64)NF * N)>>>((i64)NF * N, R, T, a, b, alpha, c0, g, rho, psi);
}

// O:135-149 calMacroPressure (what = 0): p = rho/3 + 3/2 G_01 rho_0 rho_1
// E:371-386 calPressureExpGPU (what = 1): p = rho/3 + 6/2 G_01 psi_0 psi_1
// E:19-33 calMacroPressureEX (what = 2): p = rho/3 + 3 G_11 psi_1 psi_1 -- `tmpPart2 =` (E:30) overwrites instead of
//                                        accumulating, so only the last (i, j) pair survives
__global__ void k_sc_pressure(i64 N, int what, const double *G, const double *rho, const double *psi, double *p)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double r = 0.;
    for (int k = 0; k < NF; ++k) r += COMP(rho, k, 1)[n];
    if (what == 0) {
        double t = 1. / 3. * r;
        t += 3. / 2. * G[1] * rho[n] * COMP(rho, 1, 1)[n];
        p[n] = t;
    } else if (what == 1) {
        double t = 1. / 3. * r;
        t += 6. / 2. * G[1] * psi[n] * COMP(psi, 1, 1)[n];
        p[n] = t;
    } else {
        const double part2 = G[NF * 1 + 1] * COMP(psi, 1, 1)[n] * COMP(psi, 1, 1)[n];
        p[n] = 1. / 3. * r + 3. * part2;
    }
}
static inline void launch_sc_pressure(hipStream_t st, i64 N, int what, const double *G, const double *rho, const double *psi, double *p)
{
    if (N > 0) k_sc_pressure<<<GRID_FLAT(N)>>>(N, what, G, rho, psi, p);
}

// O:186-313 calInteractionForce / O:1804-1909 interactionForceGuo: the product force of both components, stored
__global__ __launch_bounds__(NB) void k_sc_product_force(i64 N, const i64 *nbr, const double *wi, const double *G, const double *Gs, const double *psi,
                                                         double *Fx, double *Fy)
{
    __shared__ i64 lds[NB * 8];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(lds, nbr, n0, N, q);
    if (!on) return;
    for (int k = 0; k < NF; ++k) {
        double fx, fy;
        sc_product_force(N, n, k, q, wi, G, Gs, psi, fx, fy);
        COMP(Fx, k, 1)[n] = fx; COMP(Fy, k, 1)[n] = fy;
    }
}
static inline void launch_sc_product_force(hipStream_t st, i64 N, const i64 *nbr, const double *wi, const double *G, const double *Gs, const double *psi,
                                           double *Fx, double *Fy)
{
    if (N > 0) k_sc_product_force<<<GRID_NODES(N)>>>(N, nbr, wi, G, Gs, psi, Fx, Fy);
}

// O:320-330 addBodyForceGPU: the body force goes to the component with index 2 (`if (i == 2)`, O:327) -- with two
// fluids there is none, the launch changes nothing.  Kept as an entry point that does exactly that.
static inline void launch_sc_add_body_force(hipStream_t, i64, double, double, double *, double *, const double *) {}

// O:361-373 calEquilibriumVGPU: u_eq,k = u' + tau_k F_k / rho_k; flat over [2][N]
__global__ void k_sc_equilibrium_velocity(i64 N, const double *tau, const double *rho, const double *Fx, const double *Fy, const double *mvx, const double *mvy,
                                          double *ux, double *uy)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * N) return;
    const int k = (int)(e / N);
    const i64 n = e % N;
    ux[e] = mvx[n] + tau[k] * Fx[e] / rho[e];
    uy[e] = mvy[n] + tau[k] * Fy[e] / rho[e];
}
static inline void launch_sc_equilibrium_velocity(hipStream_t st, i64 N, const double *tau, const double *rho, const double *Fx, const double *Fy,
                                                  const double *mvx, const double *mvy, double *ux, double *uy)
{
    if (N > 0) k_sc_equilibrium_velocity<<<GRID_FLAT((i64)NF * N)>>>(N, tau, rho, Fx, Fy, mvx, mvy, ux, uy);
}

// O:379-429 calEquilibriumFuncGPU: f_eq of each component about ITS OWN u_eq,k; flat over [2][N][9]
__global__ void k_sc_equilibrium(i64 N, const double *w, const double *rho, const double *ux, const double *uy, double *feq)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * 9 * N) return;
    const int j = (int)(e % 9);
    const i64 kn = e / 9;
    const double vx = ux[kn], vy = uy[kn], r = rho[kn];
    const double sq = vx * vx + vy * vy;
    double v;
    switch (j) {
    case 0: v = w[0] * r * (1. - sq / (2. * 1. / 3.)); break;
    case 1: v = w[1] * r * (1. + 3. * vx + 9. / 2. * (vx * vx) - sq / (2. * 1. / 3.)); break;
    case 2: v = w[2] * r * (1. + 3. * vy + 9. / 2. * (vy * vy) - sq / (2. * 1. / 3.)); break;
    case 3: v = w[3] * r * (1. + 3. * (-1. * vx) + 9. / 2. * (vx * vx) - sq / (2. * 1. / 3.)); break;
    case 4: v = w[4] * r * (1. + 3. * (-1. * vy) + 9. / 2. * (vy * vy) - sq / (2. * 1. / 3.)); break;
    case 5: v = w[5] * r * (1. + 3. * (vx + vy) + 9. / 2. * (vx + vy) * (vx + vy) - sq / (2. * 1. / 3.)); break;
    case 6: v = w[6] * r * (1. + 3. * (-vx + vy) + 9. / 2. * (-vx + vy) * (-vx + vy) - sq / (2. * 1. / 3.)); break;
    case 7: v = w[7] * r * (1. + 3. * (-vx - vy) + 9. / 2. * (-vx - vy) * (-vx - vy) - sq / (2. * 1. / 3.)); break;
    default: v = w[8] * r * (1. + 3. * (vx - vy) + 9. / 2. * (vx - vy) * (vx - vy) - sq / (2. * 1. / 3.)); break;
    }
    feq[e] = v;
}
static inline void launch_sc_equilibrium(hipStream_t st, i64 N, const double *w, const double *rho, const double *ux, const double *uy, double *feq)
{
    if (N > 0) k_sc_equilibrium<<<GRID_FLAT((i64)NF * 9 * N)>>>(N, w, rho, ux, uy, feq);
}

// O:435-445 calCollisionSRTGPU; flat
__global__ void k_sc_collide_srt(i64 N, const double *tau, double *f, const double *feq)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)NF * 9 * N) return;
    const int k = (int)(e / (9 * N));
    f[e] = f[e] - 1. / tau[k] * (f[e] - feq[e]);
}
static inline void launch_sc_collide_srt(hipStream_t st, i64 N, const double *tau, double *f, const double *feq)
{
    if (N > 0) k_sc_collide_srt<<<GRID_FLAT((i64)NF * 9 * N)>>>(N, tau, f, feq);
}

// O:625-652 constantPressureZouHeBoundaryHigher (row ny-2): Zou-He density inlet, densityH split by the components'
// share of the local density
__global__ void k_sc_inlet_pressure_row(i64 N, i64 nx, i64 ny, double densityH, const i64 *fluidNodes, double *rho, double *f)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    double tot = 0.;
    for (int k = 0; k < NF; ++k) tot += COMP(rho, k, 1)[n];
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(f, k, 9) + 9 * n;
        const double d = COMP(rho, k, 1)[n] / tot * densityH;
        const double vy = -1. + (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / d;
        g[4] = g[2] - 2. / 3. * d * vy;
        g[7] = g[5] + 1. / 2. * (g[1] - g[3]) - 1. / 6. * d * vy;
        g[8] = g[6] - 1. / 2. * (g[1] - g[3]) - 1. / 6. * d * vy;
        COMP(rho, k, 1)[n] = d;
    }
}
static inline void launch_sc_inlet_pressure_row(hipStream_t st, i64 N, i64 nx, i64 ny, double densityH, const i64 *fluidNodes, double *rho, double *f)
{
    if (N > 0) k_sc_inlet_pressure_row<<<GRID_ROW(nx)>>>(N, nx, ny, densityH, fluidNodes, rho, f);
}
// O:659-703 ghostPointsConstantPressureInlet: row 0 <- its N neighbour, then row ny-1 <- its S neighbour
static inline void launch_sc_ghost_pressure_inlet(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const i64 *nbr, double *rho, double *f)
{
    if (N <= 0) return;
    k_sc_ghost_row<<<GRID_ROW(nx)>>>(N, nx, 0, 1, fluidNodes, nbr, rho, f);
    k_sc_ghost_row<<<GRID_ROW(nx)>>>(N, nx, ny - 1, 3, fluidNodes, nbr, rho, f);
}

// O:1127-1161 calVelocityBoundaryHigherChangGPU, O:1172-1215 calPressureBoundaryHigherChangGPU (row ny-2),
// O:1222-1267 calPressureBoundaryLowerChangGPU (row 1): Chang et al. 2009, the unknown populations from the
// pre-streaming ("old") and post-streaming ("new") ones.  The pressure forms zero the force first (O:1188, O:1238),
// so their force terms vanish; O:1211 reads fluidPDFOld[.., 5] where the pattern of O:1203 has fluidPDFNew (kept).
__global__ void k_sc_chang_velocity_high(i64 N, i64 nx, i64 ny, const double *vyIn, const i64 *fluidNodes, double *rho, const double *fOld, double *fNew)
{
    const i64 n = row_node(fluidNodes, N, nx, ny - 2);
    if (n < 0) return;
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(fNew, k, 9) + 9 * n;
        const double *o = COMP(fOld, k, 9) + 9 * n;
        const double v = vyIn[k];
        const double r = (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / (1. + v);
        COMP(rho, k, 1)[n] = r;
        g[4] = o[4] - 2. / 3. * (r * v + o[4] + o[7] + o[8]) + 2. / 3. * (g[2] + g[5] + g[6]);
        g[7] = o[7] + 1. / 2. * (g[1] - g[3]) + 1. / 6. * (g[2] - o[4]) + 2. / 3. * (g[5] - o[7]) - 1. / 3. * (g[6] - o[8]) - 1. / 6. * r * v;
        g[8] = o[8] - 1. / 6. * r * v - 1. / 2. * (g[1] - g[3]) + 1. / 6. * (g[2] - o[4]) - 1. / 3. * (g[5] - o[7]) + 2. / 3. * (g[6] - o[8]);
    }
}
static inline void launch_sc_chang_velocity_high(hipStream_t st, i64 N, i64 nx, i64 ny, const double *vyIn, const i64 *fluidNodes, double *rho,
                                                 const double *fOld, double *fNew)
{
    if (N > 0) k_sc_chang_velocity_high<<<GRID_ROW(nx)>>>(N, nx, ny, vyIn, fluidNodes, rho, fOld, fNew);
}
__global__ void k_sc_chang_pressure(i64 N, i64 nx, i64 row, int high, double rhoSet, const i64 *fluidNodes, double *rho, double *Fx, double *Fy,
                                    const double *fOld, double *fNew)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    double tot = 0.;
    for (int k = 0; k < NF; ++k) tot += COMP(rho, k, 1)[n];
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(fNew, k, 9) + 9 * n;
        const double *o = COMP(fOld, k, 9) + 9 * n;
        COMP(Fx, k, 1)[n] = 0.; COMP(Fy, k, 1)[n] = 0.;
        const double fx = 0., fy = 0.;
        const double d = COMP(rho, k, 1)[n] / tot * rhoSet;
        if (high) {
            const double v = -1. + (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / d + 1. / 2. * fy / d;
            g[4] = o[4] - 2. / 3. * (d * v + o[4] + o[7] + o[8] - g[2] - g[5] - g[6] - 1. / 2. * fy);
            g[7] = o[7] - 1. / 2. * (g[3] + g[6] + o[7] - g[1] - g[5] - o[8] - 1. / 2. * fx) -
                   1. / 6. * (d * v + o[7] + o[8] + o[4] - g[2] - g[5] - g[6] - 1. / 2. * fy);
            g[8] = o[8] + 1. / 2. * (g[3] + g[6] + o[7] - g[1] - g[5] - o[8] - 1. / 2. * fx) -
                   1. / 6. * (d * v + o[7] + o[8] + o[4] - g[2] - o[5] - g[6] - 1. / 2. * fy);
        } else {
            const double v = 1. - (g[0] + g[1] + g[3] + 2. * (g[4] + g[7] + g[8])) / d + 1. / 2. * fy / d;
            g[2] = o[2] + 2. / 3. * (d * v - o[2] + g[4] - o[5] - o[6] + g[7] + g[8]);
            g[5] = o[5] + 1. / 2. * (-g[1] + g[3] - o[5] + o[6] + g[7] - g[8] - 1. / 2. * fy) +
                   1. / 6. * (d * v - o[2] + g[4] - o[5] - o[6] + g[7] + g[8] - 1. / 2. * fy);
            g[6] = o[6] - 1. / 2. * (-g[1] + g[3] - o[5] + o[6] + g[7] - g[8] - 1. / 2. * fy) +
                   1. / 6. * (d * v - o[2] + g[4] - o[5] - o[6] + g[7] + g[8] - 1. / 2. * fy);
        }
        COMP(rho, k, 1)[n] = d;
    }
}
static inline void launch_sc_chang_pressure(hipStream_t st, i64 N, i64 nx, i64 row, int high, double rhoSet, const i64 *fluidNodes, double *rho, double *Fx,
                                            double *Fy, const double *fOld, double *fNew)
{
    if (N > 0) k_sc_chang_pressure<<<GRID_ROW(nx)>>>(N, nx, row, high, rhoSet, fluidNodes, rho, Fx, Fy, fOld, fNew);
}

// O:1454-1623 interactionCollisionEOFProcess: product force, equilibrium about the common velocity u', forcing term
// (F.(e_i - u')) 3/rho f_eq,i; the update reads f (1 - w) + f_eq + F_i (1 - w/2) with w = tauReverse -- f_eq enters
// unweighted (O:1619-1621), kept
__global__ __launch_bounds__(NB) void k_sc_eof_collision(i64 N, const double *wi, const double *om, const double *G, const double *Gs, const double *rho,
                                                         const double *psi, double *f, const i64 *nbr, double *Fx, double *Fy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (on) {
        double vxt = 0., vyt = 0., rt = 0.;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            vxt += (g[k][1] - g[k][3] + g[k][5] - g[k][6] - g[k][7] + g[k][8]) * om[k];
            vyt += (g[k][2] - g[k][4] + g[k][5] + g[k][6] - g[k][7] - g[k][8]) * om[k];
            rt += COMP(rho, k, 1)[n] * om[k];
        }
        const double ux = vxt / rt, uy = vyt / rt, usq = ux * ux + uy * uy;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            double fx, fy;
            sc_product_force(N, n, k, q, wi, G, Gs, psi, fx, fy);
            COMP(Fx, k, 1)[n] = fx; COMP(Fy, k, 1)[n] = fy;
            const double rk = COMP(rho, k, 1)[n], rcs = 3. / rk;
            double eq[9], ef[9];
            eq[0] = 4. / 9 * rk * (1. - 1.5 * usq);
            ef[0] = (fx * (-ux) + fy * (-uy)) * rcs * eq[0];
            eq[1] = 1. / 9. * rk * (1. + 3. * ux + +4.5 * (ux) * (ux) - 1.5 * usq);
            ef[1] = (fx * (1. - ux) + fy * (-uy)) * rcs * eq[1];
            eq[2] = 1. / 9. * rk * (1. + 3. * uy + +4.5 * uy * uy - 1.5 * usq);
            ef[2] = (fx * (-ux) + fy * (1. - uy)) * rcs * eq[2];
            eq[3] = 1. / 9. * rk * (1. + 3. * (-ux) + +4.5 * (-ux) * (-ux) - 1.5 * usq);
            ef[3] = (fx * (-1. - ux) + fy * (-uy)) * rcs * eq[3];
            eq[4] = 1. / 9. * rk * (1. + 3. * (-uy) + +4.5 * (-uy) * (-uy) - 1.5 * usq);
            ef[4] = (fx * (-ux) + fy * (-1. - uy)) * rcs * eq[4];
            eq[5] = 1. / 36. * rk * (1. + 3. * (ux + uy) + 4.5 * (ux + uy) * (ux + uy) - 1.5 * usq);
            ef[5] = (fx * (1. - ux) + fy * (1. - uy)) * rcs * eq[5];
            eq[6] = 1. / 36. * rk * (1. + 3. * (-ux + uy) + 4.5 * (-ux + uy) * (-ux + uy) - 1.5 * usq);
            ef[6] = (fx * (-1. - ux) + fy * (1. - uy)) * rcs * eq[6];
            eq[7] = 1. / 36. * rk * (1. + 3. * (-ux - uy) + 4.5 * (-ux - uy) * (-ux - uy) - 1.5 * usq);
            ef[7] = (fx * (-1. - ux) + fy * (-1. - uy)) * rcs * eq[7];
            eq[8] = 1. / 36. * rk * (1. + 3. * (ux - uy) + 4.5 * (ux - uy) * (ux - uy) - 1.5 * usq);
            ef[8] = (fx * (1. - ux) + fy * (-1. - uy)) * rcs * eq[8];
#pragma unroll
            for (int j = 0; j < 9; ++j) g[k][j] = g[k][j] * (1. - om[k]) + eq[j] + ef[j] * (1. - 0.5 * om[k]);
        }
    }
    for (int k = 0; k < NF; ++k) tile_out<9>(lds, COMP(f, k, 9), n0, N, g[k]);
}
static inline void launch_sc_eof_collision(hipStream_t st, i64 N, const double *wi, const double *om, const double *G, const double *Gs, const double *rho,
                                           const double *psi, double *f, const i64 *nbr, double *Fx, double *Fy)
{
    if (N > 0) k_sc_eof_collision<<<GRID_NODES(N)>>>(N, wi, om, G, Gs, rho, psi, f, nbr, Fx, Fy);
}

// O:1674-1793 calStreaming1withLinkGPU as a pull: where the upstream node is a wall the population comes back with
// the moving-wall correction -6 rho_k w_d (e_d.u), d = the direction it left in (O:1664-1666 calLinkBounceBack)
__global__ __launch_bounds__(NB) void k_sc_stream1_link(i64 N, const i64 *nbr, const double *rho, const double *f, double *fNew, const double *vx, const double *vy,
                                                        const double *w)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    for (int k = 0; k < NF; ++k) {
        double r[9];
        r[0] = 0.;
        if (on) {
            const double *fk = COMP(f, k, 9);
            const double d = COMP(rho, k, 1)[n], ux = vx[n], uy = vy[n];
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                const int o = OPP[i];
                const i64 up = q[o - 1];
                if (up >= 0) r[i] = fk[9 * up + i];
                else r[i] = fk[9 * n + o] - 3. * (2. * d * w[o] * (EX[o] * ux + EY[o] * uy));
            }
        }
        tile_out<9, true>(lds, COMP(fNew, k, 9), n0, N, r);
    }
}
static inline void launch_sc_stream1_link(hipStream_t st, i64 N, const i64 *nbr, const double *rho, const double *f, double *fNew, const double *vx,
                                          const double *vy, const double *w)
{
    if (N > 0) k_sc_stream1_link<<<GRID_NODES(N)>>>(N, nbr, rho, f, fNew, vx, vy, w);
}

// O:1917-1945 calCollisionGuo: BGK about the physical velocity with Guo's forcing term
__global__ __launch_bounds__(NB) void k_sc_collide_guo(i64 N, const double *tau, const double *w, const double *rho, const double *Fx, const double *Fy,
                                                       const double *vx, const double *vy, double *f)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    for (int k = 0; k < NF; ++k) {
        double g[9];
        tile_in<9>(lds, COMP(f, k, 9), n0, N, g);
        if (on) {
            const double ux = vx[n], uy = vy[n], u2 = ux * ux + uy * uy;
            const double fx = COMP(Fx, k, 1)[n], fy = COMP(Fy, k, 1)[n], rk = COMP(rho, k, 1)[n], tk = tau[k];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double term = w[j] * ((3. * (EX[j] - ux) + 9. * EX[j] * (EX[j] * ux + EY[j] * uy)) * fx +
                                            (3. * (EY[j] - uy) + 9. * EY[j] * (EX[j] * ux + EY[j] * uy)) * fy);
                const double eq = w[j] * rk * (1. + 3. * (EX[j] * ux + EY[j] * uy) + 4.5 * (EX[j] * ux + EY[j] * uy) * (EX[j] * ux + EY[j] * uy) - 1.5 * u2);
                g[j] = (1. - 1. / tk) * g[j] + 1. / tk * eq + (1. - 1. / (2. * tk)) * term;
            }
        }
        tile_out<9>(lds, COMP(f, k, 9), n0, N, g);
    }
}
static inline void launch_sc_collide_guo(hipStream_t st, i64 N, const double *tau, const double *w, const double *rho, const double *Fx, const double *Fy,
                                         const double *vx, const double *vy, double *f)
{
    if (N > 0) k_sc_collide_guo<<<GRID_NODES(N)>>>(N, tau, w, rho, Fx, Fy, vx, vy, f);
}

// E:38-44 calEffectiveMassPR: the reference kernel has no statements after its index computation
static inline void launch_sc_effective_mass_pr(hipStream_t, i64, double, const double *, const double *, double *) {}

// E:311-332 calTotalVelocityGPU: u = (sum_k sum_i e_i f_k,i + F_k/2) / sum_k sum_i f_k,i, one running sum over both
// components in the reference's order
__global__ __launch_bounds__(NB) void k_sc_total_velocity(i64 N, const double *Fx, const double *Fy, const double *f, double *vx, double *vy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double g[NF][9];
    for (int k = 0; k < NF; ++k) tile_in<9>(lds, COMP(f, k, 9), n0, N, g[k]);
    if (!on) return;
    double mx = 0., my = 0., r = 0.;
    for (int k = 0; k < NF; ++k) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            mx += g[k][j] * EX[j];
            my += g[k][j] * EY[j];
            r += g[k][j];
        }
        mx += 1. / 2. * COMP(Fx, k, 1)[n];
        my += 1. / 2. * COMP(Fy, k, 1)[n];
    }
    vx[n] = mx / r; vy[n] = my / r;
}
static inline void launch_sc_total_velocity(hipStream_t st, i64 N, const double *Fx, const double *Fy, const double *f, double *vx, double *vy)
{
    if (N > 0) k_sc_total_velocity<<<GRID_NODES(N)>>>(N, Fx, Fy, f, vx, vy);
}

// E:1476-1563 convectiveOutletGPUEFS / ...Ghost2GPUEFS / ...Ghost3GPUEFS (the loop's 'Freeflow' outlet,
// ShanChenD2Q9.py:1865-1884): the transformed populations, the forcing term and the equilibrium of grid row `row`
// taken from its N neighbour; the density re-summed
__global__ void k_sc_freeflow_row(i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, double *rho, double *ff, double *feq)
{
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const i64 q = nbr_node(nbr[8 * n + 1], N);
    for (int k = 0; k < NF; ++k) {
        double *g = COMP(f, k, 9), *a = COMP(ff, k, 9), *b = COMP(feq, k, 9);
        double r = 0.;
        for (int j = 0; j < 9; ++j) {
            g[9 * n + j] = g[9 * q + j];
            a[9 * n + j] = a[9 * q + j];
            b[9 * n + j] = b[9 * q + j];
            r += g[9 * q + j];
        }
        COMP(rho, k, 1)[n] = r;
    }
}
static inline void launch_sc_freeflow_row(hipStream_t st, i64 N, i64 nx, i64 row, const i64 *fluidNodes, const i64 *nbr, double *f, double *rho, double *ff,
                                          double *feq)
{
    if (N > 0) k_sc_freeflow_row<<<GRID_ROW(nx)>>>(N, nx, row, fluidNodes, nbr, f, rho, ff, feq);
}
