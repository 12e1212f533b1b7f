// sparse_kernels.hip -- kernel-level (drop-in) path: the reference's @cuda.jit kernels one by one on
// the reference's sparse AoS arrays, as HIP kernels for gfx950 behind include/lbmpm_kernels.h.
// This is the compatibility path that lets an unmodified reference-style driver loop run on
// MI355X (numba.cuda-shaped Python shim in openlbmpm_amd/numba_shim); the performance path is the
// fused solvers (rk2d.hip, sc2d.hip, rk3d.hip).
#include "lbmpm_common.h"
#include "../../include/lbmpm_kernels.h"

#include <cmath>

namespace {

using lbmpm::set_error;
typedef int64_t i64;

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

#define NF 2
#define F(f, k, n, j) (f)[((size_t)(k) * N + (n)) * 9 + (j)]
#define R(r, k, n) (r)[(size_t)(k) * N + (n)]
#define Gq(g, t, n, j) (g)[((size_t)(t) * N + (n)) * 5 + (j)]
#define Cc(c, t, n) (c)[(size_t)(t) * N + (n)]

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
__device__ double rk_feq(double rho, double w, double ex, double ey, double vx, double vy)
{
    return rho * w * (1 + (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) * (ex * vx + ey * vy) - 1.5 * (vx * vx + vy * vy)));
}

#include "sparse_kernels_gen.h"

// ---- kernels whose reference form loops over components / tracers around the node loop
__global__ void k_sc_rho(i64 N, double *rho, const double *f)       // OptimizedD2Q9GPU.py:84-94
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k) {
        double r = 0.;
        for (int j = 0; j < 9; ++j) r += F(f, k, n, j);
        R(rho, k, n) = r;
    }
}
static inline void launch_sc_rho(hipStream_t st, i64 N, double *rho, const double *f)
{
    if (N > 0) k_sc_rho<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st>>>(N, rho, f);
}

__global__ void k_sc_stream1(i64 N, const i64 *nbr, const double *f, double *fNew)   // O:452-534
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k)
        for (int i = 1; i < 9; ++i) {
            const i64 q = nbr[8 * n + i - 1];
            if (q != -1) F(fNew, k, q, i) = F(f, k, n, i);
            else F(fNew, k, n, OPP[i]) = F(f, k, n, i);
        }
}
__global__ void k_sc_stream2(i64 N, const double *fNew, double *f)                   // O:539-550
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k)
        for (int j = 1; j < 9; ++j) F(f, k, n, j) = F(fNew, k, n, j);
}
static inline void launch_sc_stream1(hipStream_t st, i64 N, const i64 *nbr, const double *f, double *fNew)
{
    if (N > 0) k_sc_stream1<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st>>>(N, nbr, f, fNew);
}
static inline void launch_sc_stream2(hipStream_t st, i64 N, const double *fNew, double *f)
{
    if (N > 0) k_sc_stream2<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st>>>(N, fNew, f);
}

// ExplicitD2Q9GPU.py:1379-1399 transformPDFandEquil (f_eq overwritten by Lambda f_eq)
__global__ void k_sc_mrt_transform_pdf_eq(i64 N, const double *f, double *feq, const double *Lam, double *fM)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k) {
        const double *L = Lam + (size_t)k * 81;
        double tP[9], tE[9];
        for (int j = 0; j < 9; ++j) {
            double a = 0., b = 0.;
            for (int m = 0; m < 9; ++m) { a += L[9 * j + m] * F(f, k, n, m); b += L[9 * j + m] * F(feq, k, n, m); }
            tP[j] = a; tE[j] = b;
        }
        for (int j = 0; j < 9; ++j) { F(fM, k, n, j) = tP[j]; F(feq, k, n, j) = tE[j]; }
    }
}
// ExplicitD2Q9GPU.py:1404-1420 transfromForceTerm
__global__ void k_sc_mrt_transform_force(i64 N, const double *ff, const double *Lam, double *ffM)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k) {
        const double *L = Lam + (size_t)k * 81;
        double tF[9];
        for (int j = 0; j < 9; ++j) {
            double a = 0.;
            for (int m = 0; m < 9; ++m) a += L[9 * j + m] * F(ff, k, n, m);
            tF[j] = a;
        }
        for (int j = 0; j < 9; ++j) F(ffM, k, n, j) = tF[j];
    }
}
// ExplicitD2Q9GPU.py:1457-1469 calAfterCollisionMRT
__global__ void k_sc_mrt_after_collision(i64 N, double *f, const double *ff, const double *feq, const double *fM,
                                         const double *ffM)
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < NF; ++k)
        for (int j = 0; j < 9; ++j) {
            const double c = (F(feq, k, n, j) - F(fM, k, n, j) - 1. / 2. * F(ffM, k, n, j));
            F(f, k, n, j) = F(f, k, n, j) + c + 1. * F(ff, k, n, j);
        }
}
#define LAUNCH1D(kern, N, ...) do { if ((N) > 0) kern<<<dim3((unsigned)(((N) + 255) / 256)), dim3(256), 0, st>>>(N, __VA_ARGS__); } while (0)
static inline void launch_sc_mrt_transform_pdf_eq(hipStream_t st, i64 N, const double *f, double *feq, const double *Lam, double *fM) { LAUNCH1D(k_sc_mrt_transform_pdf_eq, N, f, feq, Lam, fM); }
static inline void launch_sc_mrt_transform_force(hipStream_t st, i64 N, const double *ff, const double *Lam, double *ffM) { LAUNCH1D(k_sc_mrt_transform_force, N, ff, Lam, ffM); }
static inline void launch_sc_mrt_after_collision(hipStream_t st, i64 N, double *f, const double *ff, const double *feq, const double *fM, const double *ffM) { LAUNCH1D(k_sc_mrt_after_collision, N, f, ff, feq, fM, ffM); }

__global__ void k_tr_concentration(i64 N, int nT, double *C, const double *g)   // AccelerateTransport2DRK.py:78-90
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int t = 0; t < nT; ++t) {
        double c = 0.;
        for (int j = 0; j < 5; ++j) c += Gq(g, t, n, j);
        Cc(C, t, n) = c;
    }
}
__global__ void k_tr_stream1(i64 N, int nT, const i64 *nbr, const double *g, double *gNew)   // T:139-182
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int t = 0; t < nT; ++t)
        for (int j = 1; j < 5; ++j) {
            const i64 q = nbr[4 * n + j - 1];
            if (q != -1) Gq(gNew, t, q, j) = Gq(g, t, n, j);
            else Gq(gNew, t, n, OPP5[j]) = Gq(g, t, n, j);
        }
}
__global__ void k_tr_stream2(i64 N, int nT, const double *gNew, double *g)                     // T:184-194
{
    const i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int t = 0; t < nT; ++t)
        for (int j = 1; j < 5; ++j) Gq(g, t, n, j) = Gq(gNew, t, n, j);
}
static inline void launch_tr_concentration(hipStream_t st, i64 N, int nT, double *C, const double *g) { LAUNCH1D(k_tr_concentration, N, nT, C, g); }
static inline void launch_tr_stream1(hipStream_t st, i64 N, int nT, const i64 *nbr, const double *g, double *gNew) { LAUNCH1D(k_tr_stream1, N, nT, nbr, g, gNew); }
static inline void launch_tr_stream2(hipStream_t st, i64 N, int nT, const double *gNew, double *g) { LAUNCH1D(k_tr_stream2, N, nT, gNew, g); }

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
extern "C" int lbmpm_device_free(void *ptr) { if (ptr) LBMPM_HIP_TRY(hipFree(ptr)); return LBMPM_OK; }
extern "C" int lbmpm_memcpy_h2d(void *dst, const void *src, int64_t bytes)
{
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
