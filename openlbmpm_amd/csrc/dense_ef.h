// dense_ef.h -- kernel-level entry points for the LEGACY dense explicit-forcing pipeline of ShanChen2D/AccelerateGPU2D.py (SURVEY.md section 8
// row a16; no working driver of the reference reaches these kernels: ShanChenD2Q9.py:969, :1191 are dead).  Included by sparse_kernels.hip.
//
// Layout is the reference file's own: DENSE, direction-major f[9][ny * nx] float64 (a 2-D Numba array float64[:, :] passed as one device
// pointer), node fields [ny * nx], boolean masks one byte per node (isDomain, isSolid), periodic wrap on all four edges inside every kernel
// (AccelerateGPU2D.py:1345-1348).  One HIP kernel per reference kernel, arithmetic in the reference's statement order (bit parity with the
// captured known answers, tests/golden/dense_kernels.npz: 1e-13), the file's two quirks replicated on purpose:
//   * calEquilibriumFuncEFGPU (:2354): rest weight 1/6 with 2/3 u^2, "1.5 +" in the moving directions, and the typo of direction 7
//     (:2393 multiplies (-vx - vy) by (-vy - vy));
//   * calMacroVelocityGPU1D (:80): v_y is not divided by the density (:92), isDomain is ignored (solid nodes divide by their zero density).
// Access: a lane owns one node, direction-major arrays make every access of a wave a unit-stride run; the streaming kernel is written as
// the reference has it (a push: scattered stores) -- these kernels exist for name-level compatibility and parity, not for speed (the fused
// solvers are the performance path).
namespace dense {

struct Nbr { i64 F, B, U, L, FU, BU, BL, FL; };
// the eight neighbours of node id with the reference's periodic wrap and its names: F(orward) = x + 1, B = x - 1, U = y + 1, L = y - 1
__device__ __forceinline__ Nbr neighbours(i64 id, i64 nx, i64 ny)
{
    const i64 row = id / nx, col = id % nx;
    const i64 rl = row > 0 ? row - 1 : ny - 1, ru = row < ny - 1 ? row + 1 : 0;
    const i64 cb = col > 0 ? col - 1 : nx - 1, cf = col < nx - 1 ? col + 1 : 0;
    return Nbr{row * nx + cf, row * nx + cb, ru * nx + col, rl * nx + col, ru * nx + cf, ru * nx + cb, rl * nx + cb, rl * nx + cf};
}
#define DENSE_NODE const i64 id = (i64)blockIdx.x * blockDim.x + threadIdx.x; const i64 n = nx * ny; if (id >= n) return

// :54  copies the nine populations (every node), sums the density of the domain's nodes
__global__ void macro_density_1d(i64 nx, i64 ny, double *rho, const double *fc, double *fn, const uint8_t *dom)
{
    DENSE_NODE;
    double f[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { f[i] = fc[i * n + id]; fn[i * n + id] = f[i]; }
    if (dom[id]) rho[id] = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + f[8];
}
// :80  (v_y not divided: :92; every node)
__global__ void macro_velocity_1d(i64 nx, i64 ny, double *vx, double *vy, const double *rho, const double *f)
{
    DENSE_NODE;
    vx[id] = (f[1 * n + id] - f[3 * n + id] + f[5 * n + id] - f[6 * n + id] - f[7 * n + id] + f[8 * n + id]) / rho[id];
    vy[id] = (f[2 * n + id] - f[4 * n + id] + f[5 * n + id] + f[6 * n + id] - f[7 * n + id] - f[8 * n + id]);
}
// :1336  push of every node's populations to its eight neighbours (periodic), into the middle array
__global__ void streaming_step1(i64 nx, i64 ny, const double *fo, double *fm)
{
    DENSE_NODE;
    const Nbr b = neighbours(id, nx, ny);
    fm[id] = fo[id];
    fm[1 * n + b.F] = fo[1 * n + id]; fm[3 * n + b.B] = fo[3 * n + id];
    fm[2 * n + b.U] = fo[2 * n + id]; fm[4 * n + b.L] = fo[4 * n + id];
    fm[5 * n + b.FU] = fo[5 * n + id]; fm[6 * n + b.BU] = fo[6 * n + id];
    fm[7 * n + b.BL] = fo[7 * n + id]; fm[8 * n + b.FL] = fo[8 * n + id];
}
// :1372
__global__ void streaming_step2(i64 nx, i64 ny, double *fnew, const double *fm)
{
    DENSE_NODE;
#pragma unroll
    for (int i = 0; i < 9; ++i) fnew[i * n + id] = fm[i * n + id];
}
// :1392  fluid-fluid force of the explicit-forcing model, isotropy 4 (weights 1/3, 1/12), differences of the OTHER fluid's potential over
// the neighbours that belong to the domain, in the reference's order F, FU, FL, B, BU, BL, U, L
__global__ void interaction_force_ef(i64 nx, i64 ny, double constC, double G, const double *psi0, const double *psi1, double *f0x, double *f0y,
                                     double *f1x, double *f1y, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
    const Nbr b = neighbours(id, nx, ny);
    const double p0 = psi0[id], p1 = psi1[id];
    double dx0 = 0., dy0 = 0., dx1 = 0., dy1 = 0.;
    if (dom[b.F]) { dx0 += 1. / 3. * (psi0[b.F] - p0); dx1 += 1. / 3. * (psi1[b.F] - p1); }
    if (dom[b.FU]) { dx0 += 1. / 12. * (psi0[b.FU] - p0); dx1 += 1. / 12. * (psi1[b.FU] - p1); dy0 += 1. / 12. * (psi0[b.FU] - p0); dy1 += 1. / 12. * (psi1[b.FU] - p1); }
    if (dom[b.FL]) { dx0 += 1. / 12. * (psi0[b.FL] - p0); dx1 += 1. / 12. * (psi1[b.FL] - p1); dy0 += -1. / 12. * (psi0[b.FL] - p0); dy1 += -1. / 12. * (psi1[b.FL] - p1); }
    if (dom[b.B]) { dx0 += -1. / 3. * (psi0[b.B] - p0); dx1 += -1. / 3. * (psi1[b.B] - p1); }
    if (dom[b.BU]) { dx0 += -1. / 12. * (psi0[b.BU] - p0); dx1 += -1. / 12. * (psi1[b.BU] - p1); dy0 += 1. / 12. * (psi0[b.BU] - p0); dy1 += 1. / 12. * (psi1[b.BU] - p1); }
    if (dom[b.BL]) { dx0 += -1. / 12. * (psi0[b.BL] - p0); dx1 += -1. / 12. * (psi1[b.BL] - p1); dy0 += -1. / 12. * (psi0[b.BL] - p0); dy1 += -1. / 12. * (psi1[b.BL] - p1); }
    if (dom[b.U]) { dy0 += 1. / 3. * (psi0[b.U] - p0); dy1 += 1. / 3. * (psi1[b.U] - p1); }
    if (dom[b.L]) { dy0 += -1. / 3. * (psi0[b.L] - p0); dy1 += -1. / 3. * (psi1[b.L] - p1); }
    f0x[id] = -constC * p0 * G * dx1; f0y[id] = -constC * p0 * G * dy1;
    f1x[id] = -constC * p1 * G * dx0; f1y[id] = -constC * p1 * G * dy0;
}
// :2209 (weights 1/9, 1/36; every non-solid node) and :2257 (weights 1/3, 1/12; the domain's nodes): fluid-solid force, added
template <bool EF>
__global__ void external_force_solid(i64 nx, i64 ny, double gs0, double gs1, const double *psi0, const double *psi1, double *f0x, double *f0y,
                                     double *f1x, double *f1y, const uint8_t *dom, const uint8_t *solid)
{
    DENSE_NODE;
    if (EF ? !dom[id] : (solid[id] != 0)) return;
    const Nbr b = neighbours(id, nx, ny);
    constexpr double wa = EF ? 1. / 3. : 1. / 9., wd = EF ? 1. / 12. : 1. / 36.;
    double sx = 0., sy = 0.;
    if (solid[b.F]) { sx += wa; sy += wa * 0.; }
    if (solid[b.U]) { sx += wa * 0.; sy += wa; }
    if (solid[b.B]) { sx += wa * (-1.); sy += wa * 0.; }
    if (solid[b.L]) { sx += wa * 0.; sy += wa * (-1.); }
    if (solid[b.FU]) { sx += wd * 1.; sy += wd * 1.; }
    if (solid[b.BU]) { sx += wd * (-1.); sy += wd * (1.); }
    if (solid[b.BL]) { sx += wd * (-1.); sy += wd * (-1.); }
    if (solid[b.FL]) { sx += wd * (1.); sy += wd * (-1.); }
    f0x[id] += -gs0 * psi0[id] * sx; f0y[id] += -gs0 * psi0[id] * sy;
    f1x[id] += -gs1 * psi1[id] * sx; f1y[id] += -gs1 * psi1[id] * sy;
}
// :2309 (rho v / tau) and :2332 (rho v s)
template <bool MRT>
__global__ void effective_v(i64 nx, i64 ny, double a0, double a1, const double *r0, const double *r1, const double *vx0, const double *vy0,
                            const double *vx1, const double *vy1, double *ux, double *uy, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
    if (MRT) {
        ux[id] = (r0[id] * vx0[id] * a0 + r1[id] * vx1[id] * a1) / (r0[id] * a0 + r1[id] * a1);
        uy[id] = (r0[id] * vy0[id] * a0 + r1[id] * vy1[id] * a1) / (r0[id] * a0 + r1[id] * a1);
    } else {
        ux[id] = (r0[id] * vx0[id] / a0 + r1[id] * vx1[id] / a1) / (r0[id] / a0 + r1[id] / a1);
        uy[id] = (r0[id] * vy0[id] / a0 + r1[id] * vy1[id] / a1) / (r0[id] / a0 + r1[id] / a1);
    }
}
// :2354  THIS file's equilibrium (not the standard one of ExplicitD2Q9GPU.py:227), with the typo of direction 7 (:2393)
__global__ void equilibrium_ef(i64 nx, i64 ny, const double *rho, const double *ux, const double *uy, double *feq, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
    const double r = rho[id], vx = ux[id], vy = uy[id];
    const double sq = (vx * vx + vy * vy);
    feq[id] = r * (1. / 6. - 2. * sq / 3.);
    feq[1 * n + id] = 1. / 9. * r * (1.5 + 3. * vx + 4.5 * vx * vx - sq / (2. * 1. / 3.));
    feq[2 * n + id] = 1. / 9. * r * (1.5 + 3. * vy + 4.5 * vy * vy - sq / (2. * 1. / 3.));
    feq[3 * n + id] = 1. / 9. * r * (1.5 + 3. * (-vx) + 4.5 * (-vx) * (-vx) - sq / (2. * 1. / 3.));
    feq[4 * n + id] = 1. / 9. * r * (1.5 + 3. * (-vy) + 4.5 * (-vy) * (-vy) - sq / (2. * 1. / 3.));
    feq[5 * n + id] = 1. / 36. * r * (1.5 + 3. * (vx + vy) + 4.5 * (vx + vy) * (vx + vy) - sq / (2. * 1. / 3.));
    feq[6 * n + id] = 1. / 36. * r * (1.5 + 3. * (-vx + vy) + 4.5 * (-vx + vy) * (-vx + vy) - sq / (2. * 1. / 3.));
    feq[7 * n + id] = 1. / 36. * r * (1.5 + 3. * (-vx - vy) + 4.5 * (-vx - vy) * (-vy - vy) - sq / (2. * 1. / 3.));      // (-vy - vy): the reference's typo
    feq[8 * n + id] = 1. / 36. * r * (1.5 + 3. * (vx - vy) + 4.5 * (vx - vy) * (vx - vy) - sq / (2. * 1. / 3.));
}
// :2403
__global__ void forcing_term_ef(i64 nx, i64 ny, const double *rho, const double *fx, const double *fy, const double *ux, const double *uy,
                                const double *feq, double *ff, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
    constexpr double EX[9] = {0., 1., 0., -1., 0., 1., -1., -1., 1.}, EY[9] = {0., 0., 1., 0., -1., 1., 1., -1., -1.};
    const double Fx = fx[id], Fy = fy[id], vx = ux[id], vy = uy[id], r = rho[id];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        // (the reference writes -vx for 0 - vx and 1. - vx, -1. - vx for the moving directions)
        const double cx = EX[i] == 0. ? -vx : EX[i] - vx, cy = EY[i] == 0. ? -vy : EY[i] - vy;
        ff[i * n + id] = (Fx * cx + Fy * cy) * feq[i * n + id] / (1. / 3. * r);
    }
}
// :2444
__global__ void transformed_distr(i64 nx, i64 ny, double *f, const double *ff, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
#pragma unroll
    for (int i = 0; i < 9; ++i) f[i * n + id] = f[i * n + id] - 1. / 2. * ff[i * n + id];
}
// :2460
__global__ void macro_velocity_ef(i64 nx, i64 ny, const double *rho, const double *fx, const double *fy, const double *f, double *vx, double *vy,
                                  const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
    vx[id] = ((f[1 * n + id] - f[3 * n + id] + f[5 * n + id] - f[6 * n + id] + f[8 * n + id] - f[7 * n + id]) + 1. / 2. * fx[id]) / rho[id];
    vy[id] = ((f[2 * n + id] - f[4 * n + id] + f[5 * n + id] + f[6 * n + id] - f[7 * n + id] - f[8 * n + id]) + 1. / 2. * fy[id]) / rho[id];
}
// :2487
__global__ void collision_ef(i64 nx, i64 ny, double tau, double *f, const double *feq, const double *ff, const uint8_t *dom)
{
    DENSE_NODE;
    if (!dom[id]) return;
#pragma unroll
    for (int i = 0; i < 9; ++i) f[i * n + id] = f[i * n + id] + 1. / tau * (feq[i * n + id] - f[i * n + id] - 1. / 2. * ff[i * n + id]) + 1. * ff[i * n + id];
}
// :2698  half-way bounce-back as the dense file does it: a fluid node writes the reversed population into its SOLID neighbour's slot, from
// where the streaming step pushes it back (every (direction, solid node) slot has exactly one writer)
__global__ void half_wall_bounce_back(i64 nx, i64 ny, double *f, const uint8_t *dom, const uint8_t *solid)
{
    DENSE_NODE;
    if (!dom[id]) return;
    const Nbr b = neighbours(id, nx, ny);
    if (solid[b.F]) f[3 * n + b.F] = f[1 * n + id];
    if (solid[b.U]) f[4 * n + b.U] = f[2 * n + id];
    if (solid[b.B]) f[1 * n + b.B] = f[3 * n + id];
    if (solid[b.L]) f[2 * n + b.L] = f[4 * n + id];
    if (solid[b.FU]) f[7 * n + b.FU] = f[5 * n + id];
    if (solid[b.BU]) f[8 * n + b.BU] = f[6 * n + id];
    if (solid[b.BL]) f[5 * n + b.BL] = f[7 * n + id];
    if (solid[b.FL]) f[6 * n + b.FL] = f[8 * n + id];
}
#undef DENSE_NODE

inline dim3 grid_of(i64 nx, i64 ny) { return dim3((unsigned)((nx * ny + 255) / 256)); }
}  // namespace dense

#define DENSE_LAUNCH(kernel, ...) dense::kernel<<<dense::grid_of(nx, ny), dim3(256), 0, st>>>(nx, ny, __VA_ARGS__)
static inline void launch_de_macro_density(hipStream_t st, i64 nx, i64 ny, double *rho, const double *fc, double *fn, const uint8_t *dom) { DENSE_LAUNCH(macro_density_1d, rho, fc, fn, dom); }
static inline void launch_de_macro_velocity(hipStream_t st, i64 nx, i64 ny, double *vx, double *vy, const double *rho, const double *f) { DENSE_LAUNCH(macro_velocity_1d, vx, vy, rho, f); }
static inline void launch_de_stream1(hipStream_t st, i64 nx, i64 ny, const double *fo, double *fm) { DENSE_LAUNCH(streaming_step1, fo, fm); }
static inline void launch_de_stream2(hipStream_t st, i64 nx, i64 ny, double *fnew, const double *fm) { DENSE_LAUNCH(streaming_step2, fnew, fm); }
static inline void launch_de_force(hipStream_t st, i64 nx, i64 ny, double c, double g, const double *p0, const double *p1, double *a, double *b, double *cc, double *d, const uint8_t *dom)
{ DENSE_LAUNCH(interaction_force_ef, c, g, p0, p1, a, b, cc, d, dom); }
template <bool EF>
static inline void launch_de_force_solid(hipStream_t st, i64 nx, i64 ny, double g0, double g1, const double *p0, const double *p1, double *a, double *b, double *cc, double *d,
                                         const uint8_t *dom, const uint8_t *solid)
{ dense::external_force_solid<EF><<<dense::grid_of(nx, ny), dim3(256), 0, st>>>(nx, ny, g0, g1, p0, p1, a, b, cc, d, dom, solid); }
template <bool MRT>
static inline void launch_de_effective_v(hipStream_t st, i64 nx, i64 ny, double a0, double a1, const double *r0, const double *r1, const double *vx0, const double *vy0,
                                         const double *vx1, const double *vy1, double *ux, double *uy, const uint8_t *dom)
{ dense::effective_v<MRT><<<dense::grid_of(nx, ny), dim3(256), 0, st>>>(nx, ny, a0, a1, r0, r1, vx0, vy0, vx1, vy1, ux, uy, dom); }
static inline void launch_de_equilibrium(hipStream_t st, i64 nx, i64 ny, const double *rho, const double *ux, const double *uy, double *feq, const uint8_t *dom) { DENSE_LAUNCH(equilibrium_ef, rho, ux, uy, feq, dom); }
static inline void launch_de_forcing_term(hipStream_t st, i64 nx, i64 ny, const double *rho, const double *fx, const double *fy, const double *ux, const double *uy, const double *feq,
                                          double *ff, const uint8_t *dom) { DENSE_LAUNCH(forcing_term_ef, rho, fx, fy, ux, uy, feq, ff, dom); }
static inline void launch_de_transform(hipStream_t st, i64 nx, i64 ny, double *f, const double *ff, const uint8_t *dom) { DENSE_LAUNCH(transformed_distr, f, ff, dom); }
static inline void launch_de_velocity_ef(hipStream_t st, i64 nx, i64 ny, const double *rho, const double *fx, const double *fy, const double *f, double *vx, double *vy, const uint8_t *dom)
{ DENSE_LAUNCH(macro_velocity_ef, rho, fx, fy, f, vx, vy, dom); }
static inline void launch_de_collision(hipStream_t st, i64 nx, i64 ny, double tau, double *f, const double *feq, const double *ff, const uint8_t *dom) { DENSE_LAUNCH(collision_ef, tau, f, feq, ff, dom); }
static inline void launch_de_bounce_back(hipStream_t st, i64 nx, i64 ny, double *f, const uint8_t *dom, const uint8_t *solid) { DENSE_LAUNCH(half_wall_bounce_back, f, dom, solid); }
#undef DENSE_LAUNCH
