// What does the ACCESS SHAPE of the fused 2-D kernels cost, with no lattice arithmetic at all?
// A block of 512 threads owns a 64 x 8 tile of a dense [9][ny][nx] array of 16-byte population pairs (the layout of
// rk2d_fused / sc2d_fused), pulls the nine D2Q9 neighbours of its node, optionally pulls a second (halo) node of the
// three rings around the tile, optionally waits at barriers and runs a stretch of dependent fp64 FMAs, and stores its nine
// values.  The sweep in main() adds one ingredient of the real kernels at a time; results of this round: profiles/r02_tile_pull_copy.txt
// and DESIGN.md section 6.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/tile_pull_copy.hip -o /tmp/tile_pull_copy && /tmp/tile_pull_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
__device__ __forceinline__ int xcd_tile(int b, int nb) { const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3; return k * q + (k < r ? k : r) + j; }

// pull D2Q9 pairs for own cell (+ halo cell for tid < NH), optional LDS bytes to cap occupancy, WORK dummy fp64 fma per value between load and store, NBAR barriers
template <int LDSKB, int WORK, int NBAR, int NH>
__global__ __launch_bounds__(512) void tk(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny, size_t plane, int tiles_x)
{
    constexpr int Q = 9;
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    extern __shared__ double lds[];
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    const int tx0 = (t % tiles_x) * 64, ty0 = (t / tiles_x) * 8;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = tx0 + lx, y = ty0 + ly;
    double2 v[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        int xs = x - EX[i], ys = y - EY[i]; xs = xs < 0 ? xs + nx : (xs >= nx ? xs - nx : xs); ys = ys < 0 ? ys + ny : (ys >= ny ? ys - ny : ys);
        v[i] = in[i * plane + (size_t)ys * nx + xs];
    }
    double2 acc = {0., 0.};
    if (threadIdx.x < NH) {
        int n = threadIdx.x, hx, hy;
        if (n < 210) { hy = ty0 - 3 + n / 70; hx = tx0 - 3 + n % 70; }
        else if (n < 420) { n -= 210; hy = ty0 + 8 + n / 70; hx = tx0 - 3 + n % 70; }
        else { n -= 420; hy = ty0 + n / 6; const int c = n % 6; hx = c < 3 ? tx0 - 3 + c : tx0 + 64 + c - 3; }
        hx = (hx + nx) % nx; hy = (hy + ny) % ny;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            int xs = hx - EX[i], ys = hy - EY[i];
            xs = xs < 0 ? xs + nx : (xs >= nx ? xs - nx : xs); ys = ys < 0 ? ys + ny : (ys >= ny ? ys - ny : ys);
            const double2 w = in[i * plane + (size_t)ys * nx + xs];
            acc.x += w.x; acc.y += w.y;
        }
    }
    if (LDSKB > 0) lds[threadIdx.x] = acc.x + v[0].x;
    for (int b = 0; b < NBAR; ++b) {
        __syncthreads();
        if (LDSKB > 0) { acc.y += lds[(threadIdx.x + 65 * (b + 1)) & 511]; }
        // a third of the dummy work per phase
#pragma unroll 1
        for (int w = 0; w < WORK / (NBAR > 0 ? NBAR : 1); ++w) {
#pragma unroll
            for (int i = 0; i < Q; ++i) { v[i].x = fma(v[i].x, 1.0000001, acc.y * 1e-300); v[i].y = fma(v[i].y, 0.9999999, acc.x * 1e-300); }
        }
        if (LDSKB > 0) lds[threadIdx.x] = v[b % Q].x;
    }
    v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y;      // keeps the halo pull alive
#pragma unroll
    for (int i = 0; i < Q; ++i) out[i * plane + (size_t)y * nx + x] = v[i];
}

template <int LDSKB, int WORK, int NBAR, int NH>
int run(const char *name, int nx, int ny)
{
    constexpr int Q = 9;
    const size_t plane = (size_t)nx * ny;
    double2 *a, *b;
    CK(hipMalloc(&a, Q * plane * sizeof(double2))); CK(hipMalloc(&b, Q * plane * sizeof(double2)));
    CK(hipMemset(a, 0, Q * plane * sizeof(double2))); CK(hipMemset(b, 0, Q * plane * sizeof(double2)));
    const int tiles_x = nx / 64, tiles = tiles_x * (ny / 8);
    const size_t lds = LDSKB > 0 ? (size_t)LDSKB * 1024 : 0;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void *)tk<LDSKB, WORK, NBAR, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 10; ++w) { tk<LDSKB, WORK, NBAR, NH><<<tiles, 512, lds>>>(a, b, nx, ny, plane, tiles_x); std::swap(a, b); }
    CK(hipEventRecord(e0));
    const int R = 100;
    for (int w = 0; w < R; ++w) { tk<LDSKB, WORK, NBAR, NH><<<tiles, 512, lds>>>(a, b, nx, ny, plane, tiles_x); std::swap(a, b); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    const double bytes = 2.0 * Q * plane * 16;
    printf("%-46s %dx%d: %.4f ms  %.2f TB/s (own bytes)\n", name, nx, ny, ms, bytes / ms * 1e-9);
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}

int main()
{
    // WORK w: 18 fp64 fma per iteration per thread -> instructions per wave = 18 w
    run<0, 0, 0, 0>("pull, 4 blocks/CU", 1024, 1024);
    run<0, 0, 0, 468>("pull+halo, 4 blocks/CU", 1024, 1024);
    run<70, 60, 3, 468>("pull+halo, 2 blocks/CU, 3 bar, 1080 fma", 1024, 1024);
    run<70, 120, 3, 468>("pull+halo, 2 blocks/CU, 3 bar, 2160 fma", 1024, 1024);
    run<0, 0, 0, 0>("pull, 4 blocks/CU", 2048, 2048);
    run<0, 0, 0, 468>("pull+halo, 4 blocks/CU", 2048, 2048);
    run<70, 0, 0, 468>("pull+halo, 2 blocks/CU", 2048, 2048);
    run<70, 0, 3, 468>("pull+halo, 2 blocks/CU, 3 barriers", 2048, 2048);
    run<70, 30, 3, 468>("pull+halo, 2 blocks/CU, 3 bar, 540 fma", 2048, 2048);
    run<70, 60, 3, 468>("pull+halo, 2 blocks/CU, 3 bar, 1080 fma", 2048, 2048);
    run<70, 120, 3, 468>("pull+halo, 2 blocks/CU, 3 bar, 2160 fma", 2048, 2048);
    run<35, 120, 3, 468>("pull+halo, 4 blocks/CU, 3 bar, 2160 fma", 2048, 2048);
    run<70, 120, 3, 0>("pull, 2 blocks/CU, 3 bar, 2160 fma", 2048, 2048);
    run<35, 60, 3, 468>("pull+halo, 4 blocks/CU, 3 bar, 1080 fma", 2048, 2048);
    return 0;
}
