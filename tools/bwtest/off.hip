// Does the relative placement of the read and the write stream matter?  copy a -> a + 4 GiB + delta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void copy1(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <int NQ>
__global__ __launch_bounds__(256) void copyq(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cpp, int tiles)
{
    const int plane = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t o = (size_t)plane * NQ * cpp + (size_t)tile * 256 + threadIdx.x;
    double2 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = a[o + q * cpp];
#pragma unroll
    for (int q = 0; q < NQ; ++q) b[o + q * cpp] = v[q];
}
int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    char *base; CK(hipMalloc(&base, 2 * bytes + ((size_t)64 << 20)));
    CK(hipMemset(base, 0, 2 * bytes + ((size_t)64 << 20)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t deltas[] = {0, 256, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 8192, 3 << 20, (4 << 20) + 65536, 7 << 20, 16 << 20, (16 << 20) + 4096, (33 << 20) + 12288};
    for (size_t d : deltas) {
        const double2 *a = (const double2 *)base; double2 *b = (double2 *)(base + bytes + d);
        float best1 = 1e9f, bestq = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) copy1<<<(unsigned)(n / 256), 256>>>(a, b, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best1) best1 = ms;
            const size_t cpp = 172032; const int planes = (int)(n / (19 * cpp));
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) copyq<19><<<(unsigned)(cpp / 256 * planes), 256>>>(a, b, cpp, (int)(cpp / 256));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); ms *= (float)((double)n / (19.0 * cpp * planes)); if (ms < bestq) bestq = ms;
        }
        printf("delta %10zu B: plain copy %6.0f GB/s   19-array copy %6.0f GB/s\n", d, 2.0 * bytes * 5 / (best1 * 1e-3) / 1e9, 2.0 * bytes * 5 / (bestq * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
