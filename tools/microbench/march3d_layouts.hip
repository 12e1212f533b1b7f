// Skeleton of the z-marching D3Q19 kernel once more (round 5): what the ACCESS SHAPE and the step structure allow, before touching the real
// kernel.  19 eight-byte pulls + 19 eight-byte stores per cell (the pure-colour state of rk3dq_fused: no records, no rim pulls), 64 x 8 tile, one
// 512-thread workgroup per CU marching along z, pulls two planes ahead, one barrier per plane, WORK x 19 dependent fp64 FMAs per cell, all fluid.
//   LAYOUT 0: plane-major  f[z][q][y][x]            (today: a wave's 19 pulls go to 19 regions 2 MB apart)
//   LAYOUT 1: blocked      f[z][y][x / 16][q][16]   (the 19 directions of 16 consecutive cells = one 2 432-byte run)
//   LAYOUT 2: blocked      f[z][y][x / 64][q][64]   (a wave's row segment = one 9 728-byte run)
//   LAYOUT 3: plane-major with an ODD direction stride: f[z][q][ny * nx + 1168]  (round 6: at 512^2 all-fluid the 19 regions of LAYOUT 0 lie exactly 2 MiB
//             apart -- a power of two; the porous bench lattice's planes hold ~ 172 k cells, its regions are 1.38 MB apart.  Is the blocked layouts' gain
//             the blocking, or the end of that stride?)
//   BAR: with / without the barrier and the LDS phase-field ring;  NT: non-temporal stores
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/march3d_layouts.hip -o /tmp/m3l && /tmp/m3l
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
constexpr int Q = 19;
__device__ constexpr int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__device__ constexpr int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__device__ constexpr int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

template <int LAYOUT>
__device__ __forceinline__ size_t at(int z, int q, int y, int x, int nx, int ny)
{
    if (LAYOUT == 0) return (((size_t)z * Q + q) * ny + y) * nx + x;
    if (LAYOUT == 3) return ((size_t)z * Q + q) * ((size_t)ny * nx + 1168) + (size_t)y * nx + x;
    constexpr int B = LAYOUT == 1 ? 16 : 64;
    return ((((size_t)z * ny + y) * (nx / B) + x / B) * Q + q) * B + (x % B);
}

template <int LAYOUT, int WORK, bool BAR, bool NT, int TY>
__global__ __launch_bounds__(64 * TY) void march(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int nz, int rows_per_xcd, int chunk_len)
{
    constexpr int TX = 64, FX = TX + 2, FY = TY + 2;
    __shared__ double sphi[4][FY][FX];
    const int tilesX = nx / TX, tilesY = ny / TY;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int tx = slot % tilesX, r = slot / tilesX, ty = xcd * rows_per_xcd + r % rows_per_xcd, chunk = r / rows_per_xcd;
    if (ty >= tilesY) return;
    const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6;
    const int x = tx * TX + lx, y = ty * TY + ly;
    auto wrap = [](int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); };
    const int za = 1 + chunk * chunk_len, zb = min(za + chunk_len - 1, nz - 2);
    const int xs[3] = {wrap(x - 1, nx), x, wrap(x + 1, nx)}, ys[3] = {wrap(y - 1, ny), y, wrap(y + 1, ny)};
    auto pull = [&](int z, double v[Q]) {
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = in[at<LAYOUT>(z - CZ[i], i, ys[1 - CY[i]], xs[1 - CX[i]], nx, ny)];
    };
    double raw[Q], cur[Q], ft[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) { cur[i] = 0.; ft[i] = 0.; }
    pull(za - 1 < 1 ? 1 : za - 1, raw);
    for (int z = za - 2; z <= zb; ++z) {
        double a = 0.;
#pragma unroll
        for (int i = 0; i < Q; ++i) a += raw[i];
        if (BAR) sphi[(z + 1) & 3][ly + 1][lx + 1] = a;
#pragma unroll
        for (int i = 0; i < Q; ++i) { ft[i] = cur[i]; cur[i] = raw[i]; }
        if (z + 2 <= zb + 1) pull(min(z + 2, nz - 2), raw);
        if (BAR) __syncthreads();
        if (z >= za) {
            double g = a;
            if (BAR) {
                g = 0.;
#pragma unroll
                for (int i = 1; i < Q; ++i) g += sphi[(z + CZ[i]) & 3][ly + 1 + CY[i]][lx + 1 + CX[i]];
            }
            g *= 1e-300;
#pragma unroll 1
            for (int w = 0; w < WORK; ++w) {
#pragma unroll
                for (int i = 0; i < Q; ++i) ft[i] = fma(ft[i], 1.0000001, g);
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                double *d = out + at<LAYOUT>(z, i, y, x, nx, ny);
                if (NT) __builtin_nontemporal_store(ft[i], d); else *d = ft[i];
            }
        }
    }
}

template <int LAYOUT, int WORK, bool BAR, bool NT, int TY>
int run(const char *name, int n, int nz, int chunk_len)
{
    const size_t plane = (size_t)n * n, cells = (plane + (LAYOUT == 3 ? 1168 : 0)) * nz;
    double *a, *b;
    CK(hipMalloc(&a, Q * cells * sizeof(double))); CK(hipMalloc(&b, Q * cells * sizeof(double)));
    CK(hipMemset(a, 0, Q * cells * sizeof(double))); CK(hipMemset(b, 0, Q * cells * sizeof(double)));
    const int tilesX = n / 64, tilesY = n / TY, rpx = (tilesY + 7) / 8, nchunks = (nz - 2 + chunk_len - 1) / chunk_len;
    const int blocks = 8 * tilesX * rpx * nchunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { march<LAYOUT, WORK, BAR, NT, TY><<<blocks, 64 * TY>>>(a, b, n, n, nz, rpx, chunk_len); std::swap(a, b); };
    for (int w = 0; w < 3; ++w) go();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int w = 0; w < R; ++w) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    const double own = (double)plane * (nz - 2) * 2.0 * Q * 8.0;
    printf("%-66s chunk %3d: %7.3f ms  %6.0f Mcells/s  %.2f TB/s\n", name, chunk_len, ms, (double)plane * (nz - 2) / ms * 1e-3, own / ms * 1e-9);
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}

int main()
{
    const int n = 512, nz = 258;
    run<0, 0, true, false, 8>("plane-major, barrier, no fma", n, nz, 256);
    run<0, 30, true, false, 8>("plane-major, barrier, 570 fma", n, nz, 256);
    run<0, 52, true, false, 8>("plane-major, barrier, 988 fma  (the real kernel's instruction count)", n, nz, 256);
    run<0, 52, true, false, 8>("plane-major, barrier, 988 fma", n, nz, 32);
    run<0, 52, false, false, 8>("plane-major, NO barrier / ring, 988 fma", n, nz, 256);
    run<0, 0, false, false, 8>("plane-major, NO barrier / ring, no fma", n, nz, 256);
    run<0, 52, true, true, 8>("plane-major, barrier, 988 fma, non-temporal stores", n, nz, 256);
    run<3, 0, true, false, 8>("plane-major, direction stride 2 MiB + 9 344 B, barrier, no fma", n, nz, 256);
    run<3, 52, true, false, 8>("plane-major, direction stride 2 MiB + 9 344 B, barrier, 988 fma", n, nz, 256);
    run<1, 0, true, false, 8>("blocked x16, barrier, no fma", n, nz, 256);
    run<1, 52, true, false, 8>("blocked x16, barrier, 988 fma", n, nz, 256);
    run<2, 0, true, false, 8>("blocked x64, barrier, no fma", n, nz, 256);
    run<2, 52, true, false, 8>("blocked x64, barrier, 988 fma", n, nz, 256);
    run<2, 52, false, false, 8>("blocked x64, NO barrier, 988 fma", n, nz, 256);
    run<0, 52, true, false, 4>("plane-major, 64 x 4 tiles (256 threads), barrier, 988 fma", n, nz, 256);
    run<2, 52, true, false, 4>("blocked x64, 64 x 4 tiles (256 threads), barrier, 988 fma", n, nz, 256);
    return 0;
}
