// Skeleton of the z-marching D3Q19 kernel (rk3dc_fused's access shape, no lattice arithmetic) to decide
// whether storing 23 doubles per cell (19 colour-blind populations + k_R + a 3-vector, DESIGN.md section 4
// "rk3dq_fused") instead of 38 pays BEFORE the real kernel is written:
//   E = double2 : 19 sixteen-byte pulls + 19 sixteen-byte stores per cell                (today's layout, 608 B)
//   E = double  : 19 eight-byte pulls + one 32-byte scalar record read, 19 eight-byte stores + one 32-byte
//                 record written per cell; the scalar records of the tile + a 2-cell halo staged in LDS (368 B)
// Both with the rim of 148 cells per 64 x 8 tile and plane (19 pulls each, reduced to one LDS value), one
// barrier per march step, pulls two planes ahead, WORK x 19 dependent fp64 FMAs per cell, all-fluid lattice.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/march3d_pull_copy.hip -o /tmp/m3d && /tmp/m3d
// Output of round 3: profiles/r03_march3d_pull_copy.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)

constexpr int Q = 19;
__device__ constexpr int CX[Q] = {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0};
__device__ constexpr int CY[Q] = {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1};
__device__ constexpr int CZ[Q] = {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1};

__device__ __forceinline__ double val(double v) { return v; }
__device__ __forceinline__ double val(double2 v) { return v.x + v.y; }
__device__ __forceinline__ void upd(double &v, double a) { v = fma(v, 1.0000001, a); }
__device__ __forceinline__ void upd(double2 &v, double a) { v.x = fma(v.x, 1.0000001, a); v.y = fma(v.y, 0.9999999, a); }

// f[z][q][y][x] of E; s[z][y][x] of double4 (only for E = double)
template <typename E, int WORK, bool RIM, bool SCAL>
__global__ __launch_bounds__(512) void march(const E *__restrict__ in, E *__restrict__ out, const double4 *__restrict__ sin, double4 *__restrict__ sout,
                                             int nx, int ny, int nz, int rows_per_xcd, int chunk_len)
{
    constexpr int TX = 64, TY = 8, FX = TX + 2, FY = TY + 2;
    __shared__ double sphi[4][FY][FX];
    __shared__ double4 ssc[SCAL ? 4 : 1][SCAL ? TY + 4 : 1][SCAL ? TX + 4 : 1];
    const int tilesX = nx / TX, tilesY = ny / TY;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int tx = slot % tilesX, r = slot / tilesX, ty = xcd * rows_per_xcd + r % rows_per_xcd, chunk = r / rows_per_xcd;
    if (ty >= tilesY) return;
    const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6;
    const int x = tx * TX + lx, y = ty * TY + ly;
    const size_t plane = (size_t)nx * ny;
    auto wrap = [](int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); };
    // rim cell of this thread
    int hlx = 0, hly = 0, hx = 0, hy = 0;
    const bool has_rim = RIM && tid < 2 * TX + 2 * FY;
    if (has_rim) {
        if (tid < 2 * TX) { hlx = 1 + tid % TX; hly = tid < TX ? 0 : FY - 1; }
        else { const int k = tid - 2 * TX; hlx = k < FY ? 0 : FX - 1; hly = k % FY; }
        hx = wrap(tx * TX + hlx - 1, nx); hy = wrap(ty * TY + hly - 1, ny);
    }
    const int za = 1 + chunk * chunk_len, zb = min(za + chunk_len - 1, nz - 2);
    auto pull = [&](int cx, int cy, int z, E v[Q]) {
        const int xs[3] = {wrap(cx - 1, nx), cx, wrap(cx + 1, nx)}, ys[3] = {wrap(cy - 1, ny), cy, wrap(cy + 1, ny)};
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = in[((size_t)(z - CZ[i]) * Q + i) * plane + (size_t)ys[1 - CY[i]] * nx + xs[1 - CX[i]]];
    };
    // scalar records of plane z: tile + halo 2 = 12 rows x 68 columns; 8 own rows by their waves, the other 4 rows by waves 3..6,
    // the 4 x 12 side cells by wave 7
    auto fetch_s = [&](int z, double4 &a, double4 &b) {
        if (!SCAL) return;
        a = sin[(size_t)z * plane + (size_t)y * nx + x];
        if (ly >= 3 && ly <= 6) {
            const int k = ly - 3, row = k < 2 ? -2 + k : TY + k - 2;
            b = sin[(size_t)z * plane + (size_t)wrap(ty * TY + row, ny) * nx + x];
        } else if (ly == 7 && lx < 48) {
            const int row = lx / 4 - 2, c = lx & 3, col = c < 2 ? c - 2 : TX + c - 2;
            b = sin[(size_t)z * plane + (size_t)wrap(ty * TY + row, ny) * nx + wrap(tx * TX + col, nx)];
        }
    };
    auto put_s = [&](int z, const double4 &a, const double4 &b) {
        if (!SCAL) return;
        ssc[z & 3][ly + 2][lx + 2] = a;
        if (ly >= 3 && ly <= 6) { const int k = ly - 3, row = k < 2 ? k : TY + k; ssc[z & 3][row][lx + 2] = b; }
        else if (ly == 7 && lx < 48) { const int row = lx / 4, c = lx & 3, col = c < 2 ? c : TX + c; ssc[z & 3][row][col] = b; }
    };
    E raw[Q], cur[Q], ft[Q];
    double4 sa = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < Q; ++i) { cur[i] = E{}; ft[i] = E{}; }
    if (SCAL) { for (int z = za - 2; z <= za; ++z) { fetch_s(max(z, 0), sa, sb); put_s(z, sa, sb); } fetch_s(za + 1, sa, sb); }
    pull(x, y, za - 1 < 1 ? 1 : za - 1, raw);
    for (int z = za - 2; z <= zb; ++z) {
        const int zn = min(max(z + 1, 1), nz - 2);
        if (SCAL) { put_s(z + 3, sa, sb); fetch_s(min(z + 4, nz - 1), sa, sb); }
        if (has_rim) {
            E v[Q];
            pull(hx, hy, zn, v);
            double a = 0.;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                a += val(v[i]);
                if (SCAL) { const double4 s = ssc[(zn - CZ[i]) & 3][hly + 1 - CY[i]][hlx + 1 - CX[i]]; a = fma(s.x, a, s.y + s.z * s.w); }
            }
            sphi[(z + 1) & 3][hly][hlx] = a;
        }
        {
            double a = 0.;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                a += val(raw[i]);
                if (SCAL) { const double4 s = ssc[(zn - CZ[i]) & 3][ly + 2 - CY[i]][lx + 2 - CX[i]]; a = fma(s.x, a, s.y + s.z * s.w); }
            }
            sphi[(z + 1) & 3][ly + 1][lx + 1] = a;
#pragma unroll
            for (int i = 0; i < Q; ++i) { ft[i] = cur[i]; cur[i] = raw[i]; }
        }
        if (z + 2 <= zb + 1) pull(x, y, min(z + 2, nz - 2), raw);
        __syncthreads();
        if (z >= za) {
            double g = 0.;
#pragma unroll
            for (int i = 1; i < Q; ++i) g += sphi[(z + CZ[i]) & 3][ly + 1 + CY[i]][lx + 1 + CX[i]];
            g *= 1e-300;
#pragma unroll 1
            for (int w = 0; w < WORK; ++w) {
#pragma unroll
                for (int i = 0; i < Q; ++i) upd(ft[i], g);
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) out[((size_t)z * Q + i) * plane + (size_t)y * nx + x] = ft[i];
            if (SCAL) sout[(size_t)z * plane + (size_t)y * nx + x] = double4{g, g, g, g};
        }
    }
}

template <typename E, int WORK, bool RIM, bool SCAL>
int run(const char *name, int n, int nz, int chunk_len)
{
    const size_t plane = (size_t)n * n, cells = plane * nz;
    E *a, *b;
    double4 *sa = nullptr, *sb = nullptr;
    CK(hipMalloc(&a, Q * cells * sizeof(E))); CK(hipMalloc(&b, Q * cells * sizeof(E)));
    CK(hipMemset(a, 0, Q * cells * sizeof(E))); CK(hipMemset(b, 0, Q * cells * sizeof(E)));
    if (SCAL) { CK(hipMalloc(&sa, cells * sizeof(double4))); CK(hipMalloc(&sb, cells * sizeof(double4))); CK(hipMemset(sa, 0, cells * sizeof(double4))); CK(hipMemset(sb, 0, cells * sizeof(double4))); }
    const int tilesX = n / 64, tilesY = n / 8, rpx = (tilesY + 7) / 8, nchunks = (nz - 2 + chunk_len - 1) / chunk_len;
    const int blocks = 8 * tilesX * rpx * nchunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { march<E, WORK, RIM, SCAL><<<blocks, 512>>>(a, b, sa, sb, n, n, nz, rpx, chunk_len); std::swap(a, b); std::swap(sa, sb); };
    for (int w = 0; w < 3; ++w) go();
    CK(hipEventRecord(e0));
    const int R = 10;
    for (int w = 0; w < R; ++w) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    const double own = (double)plane * (nz - 2) * (2.0 * Q * sizeof(E) + (SCAL ? 64.0 : 0.0));
    printf("%-58s %d^2 x %d chunk %d: %8.3f ms  %6.0f Mcells/s  %.2f TB/s (own bytes)\n", name, n, nz, chunk_len, ms, (double)plane * (nz - 2) / ms * 1e-3, own / ms * 1e-9);
    CK(hipFree(a)); CK(hipFree(b)); if (sa) { CK(hipFree(sa)); CK(hipFree(sb)); }
    return 0;
}

int main()
{
    const int n = 512, nz = 258;
    run<double2, 0, true, false>("38 values (16-B pairs), rim, no fma", n, nz, 32);
    run<double2, 30, true, false>("38 values (16-B pairs), rim, 1140 fma", n, nz, 32);
    run<double2, 30, false, false>("38 values (16-B pairs), no rim, 1140 fma", n, nz, 32);
    run<double, 0, true, true>("23 values (8-B + 32-B record), rim, no fma", n, nz, 32);
    run<double, 30, true, true>("23 values (8-B + 32-B record), rim, 570 fma", n, nz, 32);
    run<double, 60, true, true>("23 values (8-B + 32-B record), rim, 1140 fma", n, nz, 32);
    run<double, 60, false, true>("23 values (8-B + 32-B record), no rim, 1140 fma", n, nz, 32);
    run<double, 60, true, false>("19 values (8-B), no records, rim, 1140 fma", n, nz, 32);
    return 0;
}
