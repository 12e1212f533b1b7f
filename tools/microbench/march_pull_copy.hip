// The marching counterpart of tile_pull_copy.hip: a workgroup of 512 threads owns a strip of 128 columns and walks it 4 rows per
// step over a chunk of L rows, pulling one or two steps ahead into registers (one workgroup per CU, 256 VGPRs available), one barrier
// and one LDS exchange per step, a stretch of dependent fp64 FMAs, nine 16-byte stores.  No halo pulls at all.  Would a marching
// rewrite of the fused 2-D kernels pay?  Output of this round: profiles/r02_march_pull_copy.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/march_pull_copy.hip -o /tmp/march_pull_copy && /tmp/march_pull_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
__device__ __forceinline__ int xcd_tile(int b, int nb) { const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3; return k * q + (k < r ? k : r) + j; }

template <int WORK, int DEPTH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void march(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny, size_t plane, int nstrips, int L)
{
    constexpr int Q = 9;
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    __shared__ double lds[4][4][130];
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    const int strip = t % nstrips, chunk = t / nstrips;
    const int lx = threadIdx.x & 127, ly = threadIdx.x >> 7;
    const int x = strip * 128 + lx, y0 = chunk * L;
    const int nsteps = L / 4;
    double2 A[Q], B[Q], C[Q];
    auto pull = [&](int s, double2 v[Q]) {
        const int y = y0 + 4 * s + ly;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            int xs = x - EX[i], ys = y - EY[i]; xs = xs < 0 ? xs + nx : (xs >= nx ? xs - nx : xs); ys = ys < 0 ? ys + ny : (ys >= ny ? ys - ny : ys);
            v[i] = in[i * plane + (size_t)ys * nx + xs];
        }
    };
    pull(0, A);
    if (DEPTH >= 2) pull(1, B);
    for (int s = 0; s < nsteps; ++s) {
        if (DEPTH >= 2) { if (s + 2 < nsteps) pull(s + 2, C); }
        else { if (s + 1 < nsteps) pull(s + 1, B); }
        lds[s & 3][ly][1 + lx] = A[0].x + A[1].y;
        __syncthreads();
        double acc = lds[s & 3][ly][lx] + lds[s & 3][ly][2 + lx] + lds[s & 3][(ly + 1) & 3][1 + lx];
#pragma unroll 1
        for (int w = 0; w < WORK; ++w) {
#pragma unroll
            for (int i = 0; i < Q; ++i) { A[i].x = fma(A[i].x, 1.0000001, acc * 1e-300); A[i].y = fma(A[i].y, 0.9999999, acc * 1e-300); }
        }
        const int y = y0 + 4 * s + ly;
#pragma unroll
        for (int i = 0; i < Q; ++i) out[i * plane + (size_t)y * nx + x] = A[i];
#pragma unroll
        for (int i = 0; i < Q; ++i) { A[i] = B[i]; if (DEPTH >= 2) B[i] = C[i]; }
    }
}

template <int WORK, int DEPTH>
int run(const char *name, int nx, int ny, int L)
{
    constexpr int Q = 9;
    const size_t plane = (size_t)nx * ny;
    double2 *a, *b;
    CK(hipMalloc(&a, Q * plane * sizeof(double2))); CK(hipMalloc(&b, Q * plane * sizeof(double2)));
    CK(hipMemset(a, 0, Q * plane * sizeof(double2))); CK(hipMemset(b, 0, Q * plane * sizeof(double2)));
    const int nstrips = nx / 128, blocks = nstrips * (ny / L);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 10; ++w) { march<WORK, DEPTH><<<blocks, 512>>>(a, b, nx, ny, plane, nstrips, L); std::swap(a, b); }
    CK(hipEventRecord(e0));
    const int R = 100;
    for (int w = 0; w < R; ++w) { march<WORK, DEPTH><<<blocks, 512>>>(a, b, nx, ny, plane, nstrips, L); std::swap(a, b); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    printf("%-40s %dx%d L=%d blocks=%d: %.4f ms  %.2f TB/s (own bytes)\n", name, nx, ny, L, blocks, ms, 2.0 * Q * plane * 16 / ms * 1e-9);
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}

int main()
{
    for (int L : {16, 32, 64}) {
        run<0, 2>("march depth 2, no fma", 2048, 2048, L);
        run<60, 2>("march depth 2, 1080 fma", 2048, 2048, L);
        run<120, 2>("march depth 2, 2160 fma", 2048, 2048, L);
        run<60, 1>("march depth 1, 1080 fma", 2048, 2048, L);
    }
    run<60, 2>("march depth 2, 1080 fma", 1024, 1024, 16);
    run<60, 2>("march depth 2, 1080 fma", 1024, 1024, 32);
    run<120, 2>("march depth 2, 2160 fma", 1024, 1024, 16);
    return 0;
}
