// Review of round 4, item 3: would a PERSISTENT, software-pipelined tile kernel -- one workgroup per CU walking its band of tiles, the
// next tile's own + halo populations landing in LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs) while this tile computes -- beat the
// short-lived workgroups of sc2d_fused (64 x 4 tile + 1 halo ring, 256 threads, up to four workgroups per CU)?  This is the skeleton of
// exactly that shape, next to the skeleton of the shipped shape, on the same dense [9][ny][nx] array of 16-byte population pairs, with
// the same stretch of dependent fp64 FMAs standing in for the collision (sc2d_fused: ~1250 vector instructions per wave and tile).
//
//   tile<WORK>     : the shipped shape.  Each lane pulls the nine neighbours of its node, lanes 0..139 also those of a halo node (sum
//                    only), one barrier, WORK FMAs per value pair, nine stores.  Occupancy is whatever the launch bounds allow (4 / CU).
//   persist<WORK>  : 256 threads, ONE workgroup per CU (two 57 KB buffers = 114 KB of the 160 KB LDS: a second workgroup does not fit,
//                    and a 512-thread workgroup on a 64 x 8 tile needs 95 KB per buffer).  Per tile: issue the DMA of the NEXT tile
//                    (9 directions x 66 x 6 region, 56 wave-instructions of 1 KB, 14 per wave), wait for THIS tile's DMA by count
//                    (in-order retirement: nine stores + fourteen DMA pieces are younger), raw s_barrier, nine ds_read_b128 per lane
//                    (+ nine for the halo lanes), s_barrier (the buffer is free again), WORK FMAs, nine stores.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/tile_persist.hip -o tools/microbench/tp && tools/microbench/tp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
#define VMCNT(n) (0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14))
__device__ __forceinline__ int xcd_tile(int b, int nb) { const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3; return k * q + (k < r ? k : r) + j; }
__device__ __forceinline__ int wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
constexpr int Q = 9, TW = 64, TH = 4, RW = TW + 2, RH = TH + 2, RN = RW * RH, NHALO = RN - TW * TH;      // 396 region nodes, 140 of them halo
__device__ __forceinline__ int halo_index(int n)
{
    if (n < RW) return n;                                   // bottom row
    if (n < 2 * RW) return (RH - 1) * RW + (n - RW);        // top row
    n -= 2 * RW;
    return (1 + n / 2) * RW + ((n & 1) ? RW - 1 : 0);       // the two side columns
}

template <int WORK>
__device__ __forceinline__ void work(double2 v[Q], double2 acc)
{
#pragma unroll 1
    for (int w = 0; w < WORK / (2 * Q); ++w) {
#pragma unroll
        for (int i = 0; i < Q; ++i) { v[i].x = fma(v[i].x, 1.0000001, acc.y * 1e-300); v[i].y = fma(v[i].y, 0.9999999, acc.x * 1e-300); }
    }
}

template <int WORK, int OCC>
__global__ __launch_bounds__(256, OCC) void tile(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny, size_t plane, int tiles_x)
{
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    __shared__ double s[RN];
    const int t = xcd_tile(blockIdx.x, gridDim.x);
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = tx0 + lx, y = ty0 + ly;
    double2 v[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] = in[i * plane + (size_t)wrap(y - EY[i], ny) * nx + wrap(x - EX[i], nx)];
    double2 acc = {0., 0.};
    if (threadIdx.x < NHALO) {
        const int n = halo_index(threadIdx.x);
        const int hx = wrap(tx0 - 1 + n % RW, nx), hy = wrap(ty0 - 1 + n / RW, ny);
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const double2 w = in[i * plane + (size_t)wrap(hy - EY[i], ny) * nx + wrap(hx - EX[i], nx)];
            acc.x += w.x; acc.y += w.y;
        }
        s[n] = acc.x + acc.y;
    }
    s[(ly + 1) * RW + lx + 1] = v[0].x;
    __syncthreads();
    acc.x += s[(ly + 1) * RW + lx] + s[(ly + 2) * RW + lx + 1]; acc.y += s[(ly + 1) * RW + lx + 2] + s[ly * RW + lx + 1];
    work<WORK>(v, acc);
    v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y;
#pragma unroll
    for (int i = 0; i < Q; ++i) out[i * plane + (size_t)y * nx + x] = v[i];
}

constexpr int E = Q * RN, CH = (E + 63) / 64;          // 3564 sixteen-byte elements per tile = 56 wave-instructions of 1 KB
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
typedef double d2 __attribute__((ext_vector_type(2)));

template <int WORK, int NT>
__global__ __launch_bounds__(256, 1) void persist(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny, size_t plane, int tiles_x, int ntiles)
{
    __shared__ double2 buf0[CH * 64];
    __shared__ double2 buf1[CH * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = tid & 63, ly = tid >> 6;
    const int k = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = gridDim.x >> 3, q = ntiles >> 3;
    const int t_end = (k + 1) * q;
    auto issue = [&](int t, double2 *buf) {
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int cc = 0; cc < CH / 4; ++cc) {
            const int c = wave + 4 * cc;
            int e = c * 64 + lane;
            e = e < E ? e : E - 1;
            const int i = e / RN, n = e - i * RN;
            const int ry = n / RW, rx = n - ry * RW;
            const int ex = (int)((0x20919u >> (2 * i)) & 3u) - 1;         // 1,2,1,0,1,2,0,0,2  (value + 1, two bits each, direction 0 lowest)
            const int ey = (int)((0x02865u >> (2 * i)) & 3u) - 1;         // 1,1,2,1,0,2,2,0,0
            const double2 *g = in + i * plane + (size_t)wrap(ty0 - 1 + ry - ey, ny) * nx + wrap(tx0 - 1 + rx - ex, nx);
            if (NT) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + c * 64), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + c * 64), 16, 0, 0);
        }
    };
    // The LDS reads are asm: hipcc tracks an LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in front of a ds_read it cannot
    // tell apart from the DMA's target -- which drains the NEXT tile's DMA as well (seen in the first build of this file).
    auto consume = [&](int t, const double2 *buf) {
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
        d2 vv[Q], h[Q];
        double2 v[Q], acc = {0., 0.};
        const unsigned base = (unsigned)(size_t)(lptr_t)buf;
        const unsigned a0 = base + (unsigned)((ly + 1) * RW + lx + 1) * 16u;
        const bool halo = tid < NHALO;
        const unsigned a1 = base + (unsigned)halo_index(halo ? tid : 0) * 16u;
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vv[i]) : "v"(a0), "n"(i * RN * 16));
        if (__ballot(halo) != 0ull) {
#pragma unroll
            for (int i = 0; i < Q; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(h[i]) : "v"(a1), "n"(i * RN * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < Q; ++i) { asm volatile("" : "+v"(vv[i]), "+v"(h[i])); v[i].x = vv[i].x; v[i].y = vv[i].y; }
        __builtin_amdgcn_s_barrier();                      // every wave has its values: the buffer may be refilled
        if (halo) {
#pragma unroll
            for (int i = 0; i < Q; ++i) { acc.x += h[i].x; acc.y += h[i].y; }
        }
        work<WORK>(v, acc);
        v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y;
        const size_t o = (size_t)(ty0 + ly) * nx + tx0 + lx;
#pragma unroll
        for (int i = 0; i < Q; ++i) out[i * plane + o] = v[i];
    };
    int t = k * q + j;
    if (t >= t_end) return;
    issue(t, buf0);
    int m = 0;
    for (;; ) {
        // even step: this tile in buf0, the next one into buf1
        {
            const int tn = t + wpx;
            const bool more = tn < t_end;
            if (more) {
                issue(tn, buf1);
                if (m == 0) __builtin_amdgcn_s_waitcnt(VMCNT(CH / 4)); else __builtin_amdgcn_s_waitcnt(VMCNT(CH / 4 + Q));
            } else __builtin_amdgcn_s_waitcnt(VMCNT(0));
            __builtin_amdgcn_s_barrier();
            consume(t, buf0);
            if (!more) break;
            t = tn; ++m;
        }
        {
            const int tn = t + wpx;
            const bool more = tn < t_end;
            if (more) { issue(tn, buf0); __builtin_amdgcn_s_waitcnt(VMCNT(CH / 4 + Q)); }
            else __builtin_amdgcn_s_waitcnt(VMCNT(0));
            __builtin_amdgcn_s_barrier();
            consume(t, buf1);
            if (!more) break;
            t = tn; ++m;
        }
    }
}

// persist2: the same pipeline with TWO lanes per node (lane 2n takes the first value of every pair -- one fluid component --, lane 2n + 1
// the second): 512 threads on the 64 x 4 tile = two waves per SIMD beside the same two 57 KB buffers, half the registers per lane.
// (In sc2d_fused the two components meet in the common velocity and the pseudopotential force: a two-lane exchange.)
template <int WORK>
__global__ __launch_bounds__(512, 1) void persist2(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, size_t plane, int tiles_x, int ntiles)
{
    __shared__ double2 buf0[CH * 64];
    __shared__ double2 buf1[CH * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int node = tid >> 1, comp = tid & 1, lx = node & 63, ly = node >> 6;
    const int k = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = gridDim.x >> 3, q = ntiles >> 3;
    const int t_end = (k + 1) * q;
    const double2 *in2 = reinterpret_cast<const double2 *>(in);
    auto issue = [&](int t, double2 *buf) {
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int cc = 0; cc < CH / 8; ++cc) {
            const int c = wave + 8 * cc;
            int e = c * 64 + lane;
            e = e < E ? e : E - 1;
            const int i = e / RN, n = e - i * RN;
            const int ry = n / RW, rx = n - ry * RW;
            const int ex = (int)((0x20919u >> (2 * i)) & 3u) - 1, ey = (int)((0x02865u >> (2 * i)) & 3u) - 1;
            const double2 *g = in2 + i * plane + (size_t)wrap(ty0 - 1 + ry - ey, ny) * nx + wrap(tx0 - 1 + rx - ex, nx);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + c * 64), 16, 0, 0);
        }
    };
    auto consume = [&](int t, const double2 *buf) {
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
        double v[Q], h[Q], acc = 0.;
        const unsigned base = (unsigned)(size_t)(lptr_t)buf + 8u * (unsigned)comp;
        const unsigned a0 = base + (unsigned)((ly + 1) * RW + lx + 1) * 16u;
        const bool halo = node < NHALO;
        const unsigned a1 = base + (unsigned)halo_index(halo ? node : 0) * 16u;
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a0), "n"(i * RN * 16));
        if (__ballot(halo) != 0ull) {
#pragma unroll
            for (int i = 0; i < Q; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(h[i]) : "v"(a1), "n"(i * RN * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("" : "+v"(v[i]), "+v"(h[i]));
        __builtin_amdgcn_s_barrier();
        if (halo) {
#pragma unroll
            for (int i = 0; i < Q; ++i) acc += h[i];
        }
        const double other = __shfl_xor(acc, 1);           // the partner lane's sum (tile<>: acc.x / acc.y cross over in the FMAs)
        const double mul = comp ? 0.9999999 : 1.0000001;
#pragma unroll 1
        for (int w = 0; w < WORK / (2 * Q); ++w) {
#pragma unroll
            for (int i = 0; i < Q; ++i) v[i] = fma(v[i], mul, other * 1e-300);
        }
        v[0] += 1e-300 * acc;
        const size_t o = ((size_t)(ty0 + ly) * nx + tx0 + lx) * 2 + comp;
#pragma unroll
        for (int i = 0; i < Q; ++i) out[i * plane * 2 + o] = v[i];
    };
    int t = k * q + j;
    if (t >= t_end) return;
    issue(t, buf0);
    int m = 0;
    for (;; ) {
        {
            const int tn = t + wpx;
            const bool more = tn < t_end;
            if (more) {
                issue(tn, buf1);
                if (m == 0) __builtin_amdgcn_s_waitcnt(VMCNT(CH / 8)); else __builtin_amdgcn_s_waitcnt(VMCNT(CH / 8 + Q));
            } else __builtin_amdgcn_s_waitcnt(VMCNT(0));
            __builtin_amdgcn_s_barrier();
            consume(t, buf0);
            if (!more) break;
            t = tn; ++m;
        }
        {
            const int tn = t + wpx;
            const bool more = tn < t_end;
            if (more) { issue(tn, buf0); __builtin_amdgcn_s_waitcnt(VMCNT(CH / 8 + Q)); }
            else __builtin_amdgcn_s_waitcnt(VMCNT(0));
            __builtin_amdgcn_s_barrier();
            consume(t, buf1);
            if (!more) break;
            t = tn; ++m;
        }
    }
}

// hybrid: persistent workgroups of the SHIPPED size (256 threads; three per CU, the LDS buffer leaves no room for a fourth); the tile in work lives in registers as
// today, the NEXT tile's own pulls (the part that comes from HBM: 256 nodes x 9 x 16 B = 36 KB) land meanwhile in a single LDS buffer by
// LDS-DMA, the next tile's halo pulls (L2 hits) in 18 registers.  A wave DMAs exactly the row it will read, so the own values need no
// barrier: wait by count (the nine stores of the previous tile are younger), nine ds_read_b128, then the buffer is free and the DMA of the
// tile after goes out at once -- a whole tile of look-ahead with one buffer.
template <int WORK, int OCC>
__global__ __launch_bounds__(256, OCC) void hybrid(const double2 *__restrict__ in, double2 *__restrict__ out, int nx, int ny, size_t plane, int tiles_x, int ntiles)
{
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    __shared__ double2 buf[Q * TW * TH];           // [wave = tile row][direction][lane]
    __shared__ double s[2][RN];
    const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.x & 7, j = blockIdx.x >> 3, wpx = gridDim.x >> 3, q = ntiles >> 3;
    const int t_end = (k + 1) * q;
    const bool halo = tid < NHALO;
    const int hn = halo_index(halo ? tid : 0);
    auto issue_own = [&](int t) {
        const int x = (t % tiles_x) * TW + lx, y = (t / tiles_x) * TH + ly;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const double2 *g = in + i * plane + (size_t)wrap(y - EY[i], ny) * nx + wrap(x - EX[i], nx);
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + (wave * Q + i) * 64), 16, 0, 0);
        }
    };
    d2 hq[Q];
    auto issue_halo = [&](int t) {                  // asm loads: hipcc must not wait for them (it would drain the DMA in front of them)
        const int hx = wrap((t % tiles_x) * TW - 1 + hn % RW, nx), hy = wrap((t / tiles_x) * TH - 1 + hn / RW, ny);
        if (__ballot(halo) != 0ull) {
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const double2 *g = in + i * plane + (size_t)wrap(hy - EY[i], ny) * nx + wrap(hx - EX[i], nx);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hq[i]) : "v"(g));
            }
        }
    };
    int t = k * q + j;
    if (t >= t_end) return;
    issue_own(t);
    issue_halo(t);
    const bool hw = __ballot(halo) != 0ull;          // this wave issues halo loads
    for (int m = 0;; ++m) {
        const int tn = t + wpx;
        const bool more = tn < t_end;
        // everything older than the previous tile's nine stores has landed
        if (m == 0) __builtin_amdgcn_s_waitcnt(VMCNT(0)); else __builtin_amdgcn_s_waitcnt(VMCNT(Q));
        d2 vv[Q];
        const unsigned a0 = (unsigned)(size_t)(lptr_t)buf + (unsigned)(wave * Q * 64 + lx) * 16u;
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vv[i]) : "v"(a0), "n"(i * 64 * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double2 v[Q], acc = {0., 0.};
#pragma unroll
        for (int i = 0; i < Q; ++i) { asm volatile("" : "+v"(vv[i]), "+v"(hq[i])); v[i].x = vv[i].x; v[i].y = vv[i].y; }
        if (halo) {
#pragma unroll
            for (int i = 0; i < Q; ++i) { acc.x += hq[i].x; acc.y += hq[i].y; }
        }
        // (every LDS access of this loop is asm: hipcc answers an LDS-DMA in flight with s_waitcnt vmcnt(0) in front of the next ds_write
        // and ds_read it emits itself, which drains the look-ahead)
        const unsigned sb = (unsigned)(size_t)(lptr_t)&s[m & 1][0];
        if (halo) asm volatile("ds_write_b64 %0, %1" :: "v"(sb + (unsigned)hn * 8u), "v"(acc.x + acc.y) : "memory");
        asm volatile("ds_write_b64 %0, %1" :: "v"(sb + (unsigned)((ly + 1) * RW + lx + 1) * 8u), "v"(v[0].x) : "memory");
        if (more) { issue_own(tn); issue_halo(tn); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            double n0, n1, n2, n3;
            const unsigned c = sb + (unsigned)((ly + 1) * RW + lx + 1) * 8u;
            asm volatile("ds_read_b64 %0, %1" : "=v"(n0) : "v"(c - 8u));
            asm volatile("ds_read_b64 %0, %1" : "=v"(n1) : "v"(c + 8u));
            asm volatile("ds_read_b64 %0, %1" : "=v"(n2) : "v"(c + (unsigned)RW * 8u));
            asm volatile("ds_read_b64 %0, %1" : "=v"(n3) : "v"(c - (unsigned)RW * 8u));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3));
            acc.x += n0 + n2; acc.y += n1 + n3;
        }
        work<WORK>(v, acc);
        v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y;
        const size_t o = (size_t)((t / tiles_x) * TH + ly) * nx + (t % tiles_x) * TW + lx;
#pragma unroll
        for (int i = 0; i < Q; ++i) out[i * plane + o] = v[i];
        if (!more) break;
        t = tn;
    }
    (void)hw;
}

int main()
{
    const int nx = 2048, ny = 2048;
    const size_t plane = (size_t)nx * ny, n = plane * Q;
    double2 *a, *b;
    CK(hipMalloc(&a, n * sizeof(double2))); CK(hipMalloc(&b, n * sizeof(double2)));
    {
        std::vector<double2> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = double2{(double)(i % 1009) * 1e-3, (double)(i % 917) * 1e-3};
        CK(hipMemcpy(a, h.data(), n * sizeof(double2), hipMemcpyHostToDevice));
    }
    CK(hipMemset(b, 0, n * sizeof(double2)));
    const int tiles_x = nx / TW, tiles_y = ny / TH, ntiles = tiles_x * tiles_y;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * n * sizeof(double2);
    auto run = [&](const char *name, auto launch) -> int {
        for (int w = 0; w < 3; ++w) { launch(a, b); launch(b, a); }
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int w = 0; w < 10; ++w) { launch(a, b); launch(b, a); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%-58s %.4f ms  %.2f TB/s\n", name, best / 20, bytes / (best / 20 * 1e-3) / 1e12);
        return 0;
    };
    // the two shapes must move the same data: compare a checksum of one output plane after one application each
    auto checksum = [&](double *out) -> int {
        std::vector<double2> h(plane);
        CK(hipMemcpy(h.data(), b + 3 * plane, plane * sizeof(double2), hipMemcpyDeviceToHost));
        double s = 0.; for (size_t i = 0; i < plane; ++i) s += h[i].x + 2. * h[i].y; *out = s; return 0;
    };
    double c1 = 0., c2 = 0.;
    CK(hipMemset(b, 0, n * sizeof(double2)));
    tile<0, 4><<<ntiles, 256>>>(a, b, nx, ny, plane, tiles_x); CK(hipDeviceSynchronize()); if (checksum(&c1)) return 1;
    CK(hipMemset(b, 0, n * sizeof(double2)));
    persist<0, 0><<<ncu, 256>>>(a, b, nx, ny, plane, tiles_x, ntiles); CK(hipDeviceSynchronize()); if (checksum(&c2)) return 1;
    double c4 = 0.;
    CK(hipMemset(b, 0, n * sizeof(double2)));
    hybrid<0, 3><<<ncu * 3, 256>>>(a, b, nx, ny, plane, tiles_x, ntiles); CK(hipDeviceSynchronize()); if (checksum(&c4)) return 1;
    printf("checksum of plane 3: tile %.10e  persist %.10e  hybrid %.10e  %s\n", c1, c2, c4, c1 == c2 && c1 == c4 ? "equal" : "DIFFERENT");
#define TILE(W, O) if (run("tile    64x4+1, 256 thr, <= " #O " workgroups/CU, " #W " FMAs", [&](double2 *i, double2 *o) { tile<W, O><<<ntiles, 256>>>(i, o, nx, ny, plane, tiles_x); })) return 1;
#define PERS(W, NT) if (run("persist 64x4+1, 256 thr, 1 workgroup/CU, LDS-DMA" #NT ", " #W " FMAs", [&](double2 *i, double2 *o) { persist<W, NT><<<ncu, 256>>>(i, o, nx, ny, plane, tiles_x, ntiles); })) return 1;
    TILE(0, 4) TILE(540, 4) TILE(1080, 4) TILE(1080, 3) TILE(1080, 2) TILE(2160, 4)
    PERS(0, 0) PERS(540, 0) PERS(1080, 0) PERS(2160, 0)
    PERS(0, 1) PERS(1080, 1)
#define PER2(W) if (run("persist2 64x4+1, 512 thr (2 lanes / node), 1 workgroup/CU, " #W " FMAs", [&](double2 *i, double2 *o) { persist2<W><<<ncu, 512>>>((const double *)i, (double *)o, nx, ny, plane, tiles_x, ntiles); })) return 1;
    PER2(0) PER2(540) PER2(1080) PER2(2160)
#define HYB(W, O) if (run("hybrid  64x4+1, 256 thr, " #O " persistent workgroups/CU, own pulls by LDS-DMA one tile ahead, " #W " FMAs", [&](double2 *i, double2 *o) { hybrid<W, O><<<ncu * O, 256>>>(i, o, nx, ny, plane, tiles_x, ntiles); })) return 1;
    // (buffer 36 KB + exchange tile 6 KB per workgroup: three fit a CU's 160 KB of LDS, four do not)
    HYB(0, 3) HYB(540, 3) HYB(1080, 3) HYB(2160, 3) HYB(0, 2) HYB(1080, 2) HYB(0, 1) HYB(540, 1) HYB(1080, 1)
    TILE(0, 2) TILE(0, 1) TILE(1080, 1)
    double c3 = 0.;
    CK(hipMemset(b, 0, n * sizeof(double2)));
    persist2<0><<<ncu, 512>>>((const double *)a, (double *)b, nx, ny, plane, tiles_x, ntiles); CK(hipDeviceSynchronize()); if (checksum(&c3)) return 1;
    printf("persist2 moves the same kind of data (the array has been cycled by the runs above): checksum %.6e\n", c3);
    return 0;
}
