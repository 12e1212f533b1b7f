// sparse_rest_rk.h -- the kernels of RKCG2D/AcceleratedRKGPU2D.py that none of the reference's loops launches any
// more (earlier generations of the colour-gradient step: rest-weight equilibria C_i, the three-collision form on
// f_tot, outlet copies).  Same conventions as sparse_kernels.hip (tile access, row kernels, statement order of the
// reference); pinned one by one to the real kernel bodies (tests/golden/kats_rk.npz, tests/test_kats_gpu.py).
// Included by sparse_kernels.hip.

// A:151-166 calTau1AtLocation / calTau2AtLocation: parabolic blend of the two relaxation times across the interface
__device__ __forceinline__ double rk_tau_parabolic(double Phi, double delta, double tauR, double tauB)
{
    double tau = 1.0;
    if (Phi > delta) tau = tauR;
    else if (Phi > 0 && Phi <= delta) {
        const double S1 = 2. * tauR * tauB / (tauR + tauB);
        const double S2 = 2. * (tauR - S1) / delta;
        const double S3 = -S2 / (2. * delta);
        tau = S1 + S2 * Phi + S3 * Phi * Phi;
    } else if (Phi <= 0 && Phi >= -delta) {
        const double T1 = 2. * tauR * tauB / (tauR + tauB);
        const double T2 = 2. * (T1 - tauB) / delta;
        const double T3 = T2 / (2. * delta);
        tau = T1 + T2 * Phi + T3 * Phi * Phi;
    } else if (Phi < -delta) tau = tauB;
    return tau;
}
// A:181-186 calEquilibriumRK2DOriginal: rest-weight form rho (C_i + w_i (3 e.u + 4.5 (e.u)^2 - 1.5 u^2))
__device__ __forceinline__ double rk_feq_c(double rho, double c, double w, double ex, double ey, double vx, double vy)
{
    return rho * (c + w * (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) * (ex * vx + ey * vy) - 1.5 * (vx * vx + vy * vy)));
}

// A:194-237 calRKCollision1GPU2DSRT: BGK of each colour towards its rest-weight equilibrium, in place
__global__ __launch_bounds__(NB) void k_rk_old_collide1_srt(i64 N, double delta, double tauR, double tauB, const double *cR, const double *cB, const double *w,
                                                            const double *vx, const double *vy, const double *rhoR, const double *rhoB, double *fR, double *fB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double r[9], b[9];
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (on) {
        const double rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
        const double Phi = (rR - rB) / (rR + rB);
        const double tau = rk_tau_parabolic(Phi, delta, tauR, tauB);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double eR = rk_feq_c(rR, cR[i], w[i], EX[i], EY[i], ux, uy);
            const double c1R = -1. / tau * (r[i] - eR);
            const double eB = rk_feq_c(rB, cB[i], w[i], EX[i], EY[i], ux, uy);
            const double c1B = -1. / tau * (b[i] - eB);
            r[i] = r[i] + c1R;
            b[i] = b[i] + c1B;
        }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_old_collide1_srt(hipStream_t st, i64 N, double delta, double tauR, double tauB, const double *cR, const double *cB, const double *w,
                                              const double *vx, const double *vy, const double *rhoR, const double *rhoB, double *fR, double *fB)
{
    if (N > 0) k_rk_old_collide1_srt<<<GRID_NODES(N)>>>(N, delta, tauR, tauB, cR, cB, w, vx, vy, rhoR, rhoB, fR, fB);
}

// A:429-505 calRKCollision1GPU2DMRT: the same in moment space, S[7] = S[8] = 1/tau
__global__ __launch_bounds__(NB) void k_rk_old_collide1_mrt(i64 N, double delta, double tauR, double tauB, const double *cR, const double *cB, const double *w,
                                                            const double *vx, const double *vy, const double *rhoR, const double *rhoB, double *fR, double *fB,
                                                            const double *M, const double *Minv, const double *S)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double r[9], b[9];
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (on) {
        const double rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
        const double Phi = (rR - rB) / (rR + rB);
        const double tau = rk_tau_parabolic(Phi, delta, tauR, tauB);
        double s[9], eR[9], eB[9], mR[9], mB[9], qR[9], qB[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) s[i] = S[i];
        s[7] = 1. / tau; s[8] = 1. / tau;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            eR[i] = rk_feq_c(rR, cR[i], w[i], EX[i], EY[i], ux, uy);
            eB[i] = rk_feq_c(rB, cB[i], w[i], EX[i], EY[i], ux, uy);
        }
        mat9(M, r, mR); mat9(M, eR, qR);
        mat9(M, b, mB); mat9(M, eB, qB);
#pragma unroll
        for (int i = 0; i < 9; ++i) { mR[i] = (mR[i] - qR[i]) * s[i]; mB[i] = (mB[i] - qB[i]) * s[i]; }
        mat9(Minv, mR, qR);
        mat9(Minv, mB, qB);
#pragma unroll
        for (int i = 0; i < 9; ++i) { r[i] = r[i] - qR[i]; b[i] = b[i] - qB[i]; }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_old_collide1_mrt(hipStream_t st, i64 N, double delta, double tauR, double tauB, const double *cR, const double *cB, const double *w,
                                              const double *vx, const double *vy, const double *rhoR, const double *rhoB, double *fR, double *fB,
                                              const double *M, const double *Minv, const double *S)
{
    if (N > 0) k_rk_old_collide1_mrt<<<GRID_NODES(N)>>>(N, delta, tauR, tauB, cR, cB, w, vx, vy, rhoR, rhoB, fR, fB, M, Minv, S);
}

// A:511-601 calRKCollision23GPU (the second definition of that name is the one Python keeps): gradient of
// rhoR - rhoB (NOT normalised) with the scheme weights, every non-fluid neighbour contributing solidDiff; the
// perturbation of each colour with its own A_k; recolouring with the rest-weight densities rhoR C^R_i + rhoB C^B_i.
// CGX receives G_x, CGY is zeroed (:548-549).
__global__ __launch_bounds__(NB) void k_rk_old_collide23(i64 N, double beta, double AkR, double AkB, double solidDiff, const i64 *nbr, const double *Bc,
                                                         const double *w, const double *scheme, const double *rhoR, const double *rhoB, const double *cR,
                                                         const double *cB, double *fR, double *fB, double *CGX, double *CGY)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double r[9], b[9];
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (on) {
        double gx = 0., gy = 0.;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double d = (q[i] != -1) ? (rhoR[q[i]] - rhoB[q[i]]) : solidDiff;
            gx += scheme[i + 1] * EX[i + 1] * d;
            gy += scheme[i + 1] * EY[i + 1] * d;
        }
        const double g2 = gx * gx + gy * gy, gn = sqrt(g2);
        CGX[n] = gx; CGY[n] = 0.;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double c2R = 0., c2B = 0.;
            if (g2 != 0.) {
                const double eg = EX[i] * gx + EY[i] * gy;
                const double part = w[i] * (eg * eg) / g2;
                c2R = AkR * 0.5 * gn * (part - Bc[i]);
                c2B = AkB * 0.5 * gn * (part - Bc[i]);
            }
            r[i] = r[i] + c2R;
            b[i] = b[i] + c2B;
        }
        const double rR = rhoR[n], rB = rhoB[n], rs = rR + rB, rm = rR * rB, rs2 = rs * rs;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double en = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            double c = 0.;
            if (!(en == 0. || gn == 0.)) c = (EX[i] * gx + EY[i] * gy) / (sqrt(EX[i] * EX[i] + EY[i] * EY[i]) * gn);
            const double feqRho = rR * cR[i] + rB * cB[i];
            const double sum = r[i] + b[i];
            r[i] = rR / rs * sum + (beta * rm / rs2) * feqRho * c;
            b[i] = rB / rs * sum - (beta * rm / rs2) * feqRho * c;
        }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_old_collide23(hipStream_t st, i64 N, double beta, double AkR, double AkB, double solidDiff, const i64 *nbr, const double *Bc,
                                           const double *w, const double *scheme, const double *rhoR, const double *rhoB, const double *cR, const double *cB,
                                           double *fR, double *fB, double *CGX, double *CGY)
{
    if (N > 0) k_rk_old_collide23<<<GRID_NODES(N)>>>(N, beta, AkR, AkB, solidDiff, nbr, Bc, w, scheme, rhoR, rhoB, cR, cB, fR, fB, CGX, CGY);
}

// A:887-900 copyFluidPDFLastStep / A:906-919 copyFluidPDFRecoverOutlet: the populations of grid rows 0..2, both
// colours, copied to / from a second pair of arrays.  fluidNodes ascends: those are the leading compact nodes.
__global__ void k_rk_copy_outlet_rows(i64 N, i64 nx, const i64 *fluidNodes, const double *srcR, const double *srcB, double *dstR, double *dstB)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n = k / 9;
    if (n >= N || fluidNodes[n] >= 3 * nx) return;
    dstR[k] = srcR[k];
    dstB[k] = srcB[k];
}
static inline void launch_rk_copy_outlet_rows(hipStream_t st, i64 N, i64 nx, const i64 *fluidNodes, const double *srcR, const double *srcB, double *dstR, double *dstB)
{
    const i64 lead = N < 3 * nx ? N : 3 * nx;
    if (lead > 0) k_rk_copy_outlet_rows<<<GRID_FLAT(9 * lead)>>>(N, nx, fluidNodes, srcR, srcB, dstR, dstB);
}

// A:1363-1376 calNeumannPhiOutlet: the phase field of row 2 copied onto rows 1 and 0 (through the N and S links)
__global__ void k_rk_neumann_phi_outlet(i64 N, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *phi)
{
    const i64 n = row_node(fluidNodes, N, nx, 1);
    if (n < 0) return;
    const i64 up = nbr_node(nbr[8 * n + 1], N), lo = nbr_node(nbr[8 * n + 3], N);
    phi[n] = phi[up];
    phi[lo] = phi[up];
}
static inline void launch_rk_neumann_phi_outlet(hipStream_t st, i64 N, i64 nx, const i64 *fluidNodes, const i64 *nbr, double *phi)
{
    if (N > 0) k_rk_neumann_phi_outlet<<<GRID_ROW(nx)>>>(N, nx, fluidNodes, nbr, phi);
}

// A:1382-1408 calModifiedPeriodicBoundary: the populations that have just wrapped around in y change colour
// (directions 2, 5, 6 on row 0; 4, 7, 8 on row ny-1)
__global__ void k_rk_modified_periodic(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, double *fR, double *fB)
{
    const i64 row = blockIdx.y == 0 ? 0 : ny - 1;
    const i64 n = row_node(fluidNodes, N, nx, row);
    if (n < 0) return;
    const int d0 = blockIdx.y == 0 ? 2 : 4, d1 = blockIdx.y == 0 ? 5 : 7, d2 = blockIdx.y == 0 ? 6 : 8;
    double *r = fR + 9 * n, *b = fB + 9 * n;
    const double s0 = r[d0], s1 = r[d1], s2 = r[d2];
    r[d0] = b[d0]; r[d1] = b[d1]; r[d2] = b[d2];
    b[d0] = s0; b[d1] = s1; b[d2] = s2;
}
static inline void launch_rk_modified_periodic(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, double *fR, double *fB)
{
    if (N <= 0) return;
    // ny == 1 would make both tests of the reference true for the same node (swap twice); rows are distinct otherwise
    k_rk_modified_periodic<<<dim3((unsigned)((nx + 63) / 64), ny > 1 ? 2 : 1), dim3(64), 0, st>>>(N, nx, ny, fluidNodes, fR, fB);
}

// A:1430-1462 calRKCollision1TotalGPU2DSRT: BGK of f_tot (harmonic tau(phi)) into a separate array
__global__ __launch_bounds__(NB) void k_rk_total_collide1_srt(i64 N, double tauR, double tauB, const double *w, const double *vx, const double *vy,
                                                              const double *rhoR, const double *rhoB, const double *phi, const double *fT, double *c1)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double t[9], o[9];
    tile_in<9>(lds, fT, n0, N, t);
    if (on) {
        const double Phi = phi[n];
        const double tau = 0.5 + 1. / ((1. + Phi) / (2. * (tauR - 0.5)) + (1. - Phi) / (2. * (tauB - 0.5)));
        const double rR = rhoR[n], rB = rhoB[n], ux = vx[n], uy = vy[n];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double eR = rk_feq(rR, w[i], EX[i], EY[i], ux, uy);
            const double eB = rk_feq(rB, w[i], EX[i], EY[i], ux, uy);
            const double eT = eR + eB;
            o[i] = -1. / tau * (t[i] - eT) + t[i];
        }
    }
    tile_out<9>(lds, c1, n0, N, o);
}
static inline void launch_rk_total_collide1_srt(hipStream_t st, i64 N, double tauR, double tauB, const double *w, const double *vx, const double *vy,
                                                const double *rhoR, const double *rhoB, const double *phi, const double *fT, double *c1)
{
    if (N > 0) k_rk_total_collide1_srt<<<GRID_NODES(N)>>>(N, tauR, tauB, w, vx, vy, rhoR, rhoB, phi, fT, c1);
}

// A:1468-1513 calRKCollision2TotalGPUNew: G = 3 sum w e phi (solidPhi on non-fluid neighbours), stored; the
// perturbation (A/2)|G|(w (e.G)^2/|G|^2 - B_i) into a separate array
__global__ __launch_bounds__(NB) void k_rk_total_collide2(i64 N, double A, double solidPhi, const i64 *nbr, const double *Bc, const double *w, const double *phi,
                                                          double *c2, double *Gx, double *Gy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    i64 q[8];
    tile_in<8>(reinterpret_cast<i64 *>(lds), nbr, n0, N, q);
    double o[9];
    if (on) {
        double gx = 0., gy = 0.;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double ph = (q[i] != -1) ? phi[q[i]] : solidPhi;
            gx += w[i + 1] * ph * EX[i + 1];
            gy += w[i + 1] * ph * EY[i + 1];
        }
        gx = 3. * gx; gy = 3. * gy;
        const double g2 = gx * gx + gy * gy, gn = sqrt(g2);
        Gx[n] = gx; Gy[n] = gy;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (g2 == 0) o[i] = 0.;
            else o[i] = A / 2. * gn * (w[i] * (EX[i] * gx + EY[i] * gy) * (EX[i] * gx + EY[i] * gy) / g2 - Bc[i]);
        }
    }
    tile_out<9>(lds, c2, n0, N, o);
}
static inline void launch_rk_total_collide2(hipStream_t st, i64 N, double A, double solidPhi, const i64 *nbr, const double *Bc, const double *w, const double *phi,
                                            double *c2, double *Gx, double *Gy)
{
    if (N > 0) k_rk_total_collide2<<<GRID_NODES(N)>>>(N, A, solidPhi, nbr, Bc, w, phi, c2, Gx, Gy);
}

// A:1519-1554 calRecoloringProcess: (collision 1 + collision 2) split by colour fraction, the recolouring term
// beta rhoR rhoB / rho w_i cos, ADDED to what fR, fB hold
__global__ __launch_bounds__(NB) void k_rk_recolor_add(i64 N, double beta, const double *w, const double *rhoR, const double *rhoB, const double *Gx,
                                                       const double *Gy, const double *c1, const double *c2, double *fR, double *fB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double a[9], c[9], r[9], b[9];
    tile_in<9>(lds, c1, n0, N, a);
    tile_in<9>(lds, c2, n0, N, c);
    tile_in<9>(lds, fR, n0, N, r);
    tile_in<9>(lds, fB, n0, N, b);
    if (on) {
        const double gx = Gx[n], gy = Gy[n];
        const double gn = sqrt(gx * gx + gy * gy);
        const double rR = rhoR[n], rB = rhoB[n], tot = rR + rB;
        double cs = 0.;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double en = sqrt(EX[i] * EX[i] + EY[i] * EY[i]);
            if (gn == 0. || en == 0.) cs = 0.;
            else if (gn > 0. && en > 0.) cs = (EX[i] * gx + EY[i] * gy) / (en * gn);
            const double tp = a[i] + c[i];
            r[i] = rR / tot * tp + beta * rR * rB / tot * w[i] * cs + r[i];
            b[i] = rB / tot * tp - beta * rR * rB / tot * w[i] * cs + b[i];
        }
    }
    tile_out<9>(lds, fR, n0, N, r);
    tile_out<9>(lds, fB, n0, N, b);
}
static inline void launch_rk_recolor_add(hipStream_t st, i64 N, double beta, const double *w, const double *rhoR, const double *rhoB, const double *Gx,
                                         const double *Gy, const double *c1, const double *c2, double *fR, double *fB)
{
    if (N > 0) k_rk_recolor_add<<<GRID_NODES(N)>>>(N, beta, w, rhoR, rhoB, Gx, Gy, c1, c2, fR, fB);
}

// A:1907-1930 calPhysicalVelocityRKGPU2DVNew / A:2610-2626 calMacroDensityRKGPU2DNew: as A:2634 / A:103, below the
// two inlet rows only (grid index < (ny-2) nx)
__global__ __launch_bounds__(NB) void k_rk_velocity_below_inlet(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const double *fT, const double *rhoR,
                                                                const double *rhoB, double *vx, double *vy, const double *Fx, const double *Fy)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double f[9];
    tile_in<9>(lds, fT, n0, N, f);
    if (!on || fluidNodes[n] >= (ny - 2) * nx) return;
    const double rs = rhoB[n] + rhoR[n];
    const double ux = f[1] - f[3] + f[5] - f[6] - f[7] + f[8] + 0.5 * Fx[n];
    vx[n] = ux / rs;
    const double uy = f[2] - f[4] + f[5] + f[6] - f[7] - f[8] + 0.5 * Fy[n];
    vy[n] = uy / rs;
}
static inline void launch_rk_velocity_below_inlet(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const double *fT, const double *rhoR,
                                                  const double *rhoB, double *vx, double *vy, const double *Fx, const double *Fy)
{
    if (N > 0) k_rk_velocity_below_inlet<<<GRID_NODES(N)>>>(N, nx, ny, fluidNodes, fT, rhoR, rhoB, vx, vy, Fx, Fy);
}
__global__ __launch_bounds__(NB) void k_rk_density_below_inlet(i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const double *fR, const double *fB, double *rhoR,
                                                               double *rhoB)
{
    __shared__ double lds[NB * 9];
    THIS_NODE;
    double a[9], b[9];
    tile_in<9>(lds, fR, n0, N, a);
    tile_in<9>(lds, fB, n0, N, b);
    if (!on || fluidNodes[n] >= (ny - 2) * nx) return;
    double r = 0., s = 0.;
#pragma unroll
    for (int i = 0; i < 9; ++i) { r += a[i]; s += b[i]; }
    rhoR[n] = r; rhoB[n] = s;
}
static inline void launch_rk_density_below_inlet(hipStream_t st, i64 N, i64 nx, i64 ny, const i64 *fluidNodes, const double *fR, const double *fB, double *rhoR,
                                                 double *rhoB)
{
    if (N > 0) k_rk_density_below_inlet<<<GRID_NODES(N)>>>(N, nx, ny, fluidNodes, fR, fB, rhoR, rhoB);
}
