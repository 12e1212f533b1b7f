// Stand-alone HBM stream experiments (tuning aid): which access pattern reaches the device's copy rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: grid-stride, 4 far-apart accesses per thread (the library's stream test)
__global__ __launch_bounds__(256) void copy_gridstride(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3;
    }
    for (; i < n; i += stride) b[i] = a[i];
}
// B: one contiguous chunk per block, U x 16 B per thread, one-shot grid
template <int U>
__global__ __launch_bounds__(256) void copy_chunk(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) if (base + k * 256 < n) v[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) if (base + k * 256 < n) b[base + k * 256] = v[k];
}
// C: like B but each thread owns 32 contiguous bytes (torch-style vec4 of double... here 2 x double2 adjacent)
template <int U>
__global__ __launch_bounds__(256) void copy_chunk32(const double4 *__restrict__ a, double4 *__restrict__ b, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    double4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) if (base + k * 256 < n4) v[k] = a[base + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) if (base + k * 256 < n4) b[base + k * 256] = v[k];
}
// D: persistent blocks walking contiguous chunks in block-cyclic order (chunk c -> block c % grid)
template <int U>
__global__ __launch_bounds__(256) void copy_cyclic(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    const size_t chunk = 256 * U, nchunks = (n + chunk - 1) / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) if (base + k * 256 < n) v[k] = a[base + k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) if (base + k * 256 < n) b[base + k * 256] = v[k];
    }
}
// E: 19 separate streams per block like the LBM kernel: q-th stream at offset q * plane; block handles 512 cells
__global__ __launch_bounds__(512) void copy_19streams(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int planes)
{
    // layout [plane][19][cells]; one block = 512 cells of one plane... marches over `planes` planes
    const size_t cell = (size_t)blockIdx.x * 512 + threadIdx.x;
    if (cell >= cells_per_plane) return;
    for (int p = 0; p < planes; ++p) {
        const size_t o = (size_t)p * 19 * cells_per_plane + cell;
        double2 v[19];
#pragma unroll
        for (int q = 0; q < 19; ++q) v[q] = a[o + q * cells_per_plane];
#pragma unroll
        for (int q = 0; q < 19; ++q) b[o + q * cells_per_plane] = v[q];
    }
}


// F: NQ streams, one-shot blocks: block = (plane, 512-cell tile); no marching
template <int NQ, int TPB>
__global__ __launch_bounds__(TPB) void copy_streams_oneshot(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int tiles)
{
    const int plane = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t cell = (size_t)tile * TPB + threadIdx.x;
    if (cell >= cells_per_plane) return;
    const size_t o = (size_t)plane * NQ * cells_per_plane + cell;
    double2 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = a[o + q * cells_per_plane];
#pragma unroll
    for (int q = 0; q < NQ; ++q) b[o + q * cells_per_plane] = v[q];
}
// G: marching with chunks of `len` planes per block (more blocks than CUs), TPB threads
template <int NQ, int TPB>
__global__ __launch_bounds__(TPB) void copy_streams_march(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int tiles, int planes, int len)
{
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t cell = (size_t)tile * TPB + threadIdx.x;
    if (cell >= cells_per_plane) return;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    for (int p = p0; p < p1; ++p) {
        const size_t o = (size_t)p * NQ * cells_per_plane + cell;
        double2 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = a[o + q * cells_per_plane];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[o + q * cells_per_plane] = v[q];
    }
}
// H: marching, software-pipelined: loads of plane p+1 in flight while plane p is stored
template <int NQ, int TPB>
__global__ __launch_bounds__(TPB) void copy_streams_march_pipe(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int tiles, int planes, int len)
{
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t cell = (size_t)tile * TPB + threadIdx.x;
    if (cell >= cells_per_plane) return;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    double2 v[NQ], w[NQ];
    {
        const size_t o = (size_t)p0 * NQ * cells_per_plane + cell;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = a[o + q * cells_per_plane];
    }
    for (int p = p0; p < p1; ++p) {
        const size_t o = (size_t)p * NQ * cells_per_plane + cell;
        if (p + 1 < p1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) w[q] = a[o + (size_t)NQ * cells_per_plane + q * cells_per_plane];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[o + q * cells_per_plane] = v[q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = w[q];
    }
}


// I: marching, 19 runs per tile-plane stored back to back: layout [plane][tile][q][TPB cells]
template <int NQ, int TPB>
__global__ __launch_bounds__(TPB) void copy_tiled_march(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int tiles, int planes, int len)
{
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    for (int p = p0; p < p1; ++p) {
        const size_t o = ((size_t)p * cells_per_plane + (size_t)tile * TPB) * NQ + threadIdx.x;
        double2 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = a[o + q * TPB];
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[o + q * TPB] = v[q];
    }
}


// J: like G (marching, NQ strided streams) but loads/stores in groups of GRP (finer read/write interleave), optional nontemporal
template <int NQ, int TPB, int GRP, bool NT>
__global__ __launch_bounds__(TPB) void copy_streams_groups(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cells_per_plane, int tiles, int planes, int len)
{
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t cell = (size_t)tile * TPB + threadIdx.x;
    if (cell >= cells_per_plane) return;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    for (int p = p0; p < p1; ++p) {
        const size_t o = (size_t)p * NQ * cells_per_plane + cell;
#pragma unroll
        for (int g = 0; g < NQ; g += GRP) {
            double2 v[GRP];
#pragma unroll
            for (int q = 0; q < GRP; ++q) if (g + q < NQ) {
                if (NT) { v[q].x = __builtin_nontemporal_load(&a[o + (g + q) * cells_per_plane].x); v[q].y = __builtin_nontemporal_load(&a[o + (g + q) * cells_per_plane].y); }
                else v[q] = a[o + (g + q) * cells_per_plane];
            }
#pragma unroll
            for (int q = 0; q < GRP; ++q) if (g + q < NQ) {
                if (NT) { __builtin_nontemporal_store(v[q].x, &b[o + (g + q) * cells_per_plane].x); __builtin_nontemporal_store(v[q].y, &b[o + (g + q) * cells_per_plane].y); }
                else b[o + (g + q) * cells_per_plane] = v[q];
            }
        }
    }
}


// K: one-shot chunk copy, U x 16 B per thread; ADJ: the thread's U accesses are adjacent (128 B per thread) instead of 4 KB apart;
//    dynamic LDS only to limit the number of resident blocks per CU
template <int U, bool ADJ>
__global__ __launch_bounds__(256) void copy_chunk_k(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    extern __shared__ char lds[];
    if (threadIdx.x == 9999) lds[0] = 1;
    const size_t blk = (size_t)blockIdx.x * (256 * U);
    double2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = ADJ ? blk + (size_t)threadIdx.x * U + k : blk + threadIdx.x + k * 256; if (i < n) v[k] = a[i]; }
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = ADJ ? blk + (size_t)threadIdx.x * U + k : blk + threadIdx.x + k * 256; if (i < n) b[i] = v[k]; }
}


// L: marching, double-buffered by manual 2x unrolling (no register copies): loads of plane p+1 in flight while plane p is stored;
//    STAG: interleave "load q of p+1, store q of p" instead of all loads first
template <int NQ, int TPB, bool STAG>
__global__ __launch_bounds__(TPB) void copy_march_db(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cpp, int tiles, int planes, int len)
{
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const size_t cell = (size_t)tile * TPB + threadIdx.x;
    if (cell >= cpp) return;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    double2 v[NQ], w[NQ];
    auto ld = [&](double2 *r, int p) {
        const size_t o = (size_t)p * NQ * cpp + cell;
#pragma unroll
        for (int q = 0; q < NQ; ++q) r[q] = a[o + q * cpp];
    };
    auto st = [&](const double2 *r, int p) {
        const size_t o = (size_t)p * NQ * cpp + cell;
#pragma unroll
        for (int q = 0; q < NQ; ++q) b[o + q * cpp] = r[q];
    };
    auto ldst = [&](double2 *rn, int pn, const double2 *rc, int pc) {      // staggered
        const size_t on = (size_t)pn * NQ * cpp + cell, oc = (size_t)pc * NQ * cpp + cell;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { rn[q] = a[on + q * cpp]; b[oc + q * cpp] = rc[q]; __builtin_amdgcn_sched_barrier(0); }
    };
    ld(v, p0);
    int p = p0;
    for (; p + 2 < p1; p += 2) {
        if (STAG) { ldst(w, p + 1, v, p); ldst(v, p + 2, w, p + 1); }
        else { ld(w, p + 1); st(v, p); ld(v, p + 2); st(w, p + 1); }
    }
    for (; p < p1; ++p) { st(v, p); if (p + 1 < p1) ld(v, p + 1); }
}


// M: marching copy of 19 streams where the 19 directions of a cell are split over NG wave groups of the workgroup:
//    TPB threads = (TPB / NG) cells x NG groups; group g (whole waves) moves directions g, g + NG, ...
template <int NQ, int TPB, int NG>
__global__ __launch_bounds__(TPB) void copy_march_split(const double2 *__restrict__ a, double2 *__restrict__ b, size_t cpp, int tiles, int planes, int len)
{
    constexpr int CELLS = TPB / NG, PER = (NQ + NG - 1) / NG;
    const int chunk = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int g = threadIdx.x / CELLS;
    const size_t cell = (size_t)tile * CELLS + threadIdx.x % CELLS;
    if (cell >= cpp) return;
    const int p0 = chunk * len, p1 = min(planes, p0 + len);
    for (int p = p0; p < p1; ++p) {
        const size_t o = (size_t)p * NQ * cpp + cell;
        double2 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int q = g + k * NG; if (q < NQ) v[k] = a[o + q * cpp]; }
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int q = g + k * NG; if (q < NQ) b[o + q * cpp] = v[k]; }
    }
}

template <typename F>
double time_it(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

int main()
{
    CK(hipFuncSetAttribute((const void *)copy_chunk_k<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000));
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    double2 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    auto rep = [&](const char *name, double t) { printf("%-44s %7.0f GB/s\n", name, 2.0 * bytes / t / 1e9); fflush(stdout); };
    for (int g : {256 * 8, 256 * 16, 256 * 32}) {
        char nm[64]; snprintf(nm, 64, "A grid-stride, %d blocks", g);
        rep(nm, time_it([&] { copy_gridstride<<<g, 256>>>(a, b, n); }, 10));
    }
    rep("B chunk U=1 (4 KB per block)", time_it([&] { copy_chunk<1><<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, 10));
    rep("B chunk U=2", time_it([&] { copy_chunk<2><<<(unsigned)((n + 511) / 512), 256>>>(a, b, n); }, 10));
    rep("B chunk U=4 (16 KB per block)", time_it([&] { copy_chunk<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); }, 10));
    rep("B chunk U=8", time_it([&] { copy_chunk<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); }, 10));
    rep("C chunk 32 B lanes U=1", time_it([&] { copy_chunk32<1><<<(unsigned)((n / 2 + 255) / 256), 256>>>((const double4 *)a, (double4 *)b, n / 2); }, 10));
    rep("C chunk 32 B lanes U=2", time_it([&] { copy_chunk32<2><<<(unsigned)((n / 2 + 511) / 512), 256>>>((const double4 *)a, (double4 *)b, n / 2); }, 10));
    rep("C chunk 32 B lanes U=4", time_it([&] { copy_chunk32<4><<<(unsigned)((n / 2 + 1023) / 1024), 256>>>((const double4 *)a, (double4 *)b, n / 2); }, 10));
    for (int g : {256 * 4, 256 * 8, 256 * 16}) {
        char nm[64]; snprintf(nm, 64, "D cyclic U=4, %d blocks", g);
        rep(nm, time_it([&] { copy_cyclic<4><<<g, 256>>>(a, b, n); }, 10));
    }
    {   // E: 512^2 * 0.65 ~ 170k cells per plane, 19 populations; planes so that total = 4 GiB
        const size_t cpp = 172032;            // multiple of 512
        const int planes = (int)(n / (19 * cpp));
        const double t = time_it([&] { copy_19streams<<<(unsigned)(cpp / 512), 512>>>(a, b, cpp, planes); }, 5);
        printf("%-44s %7.0f GB/s   (%d planes, 336 blocks marching)\n", "E 19 streams, blocks march over planes", 2.0 * 16 * 19 * cpp * planes / t / 1e9, planes);
        // same data, but chunked in z like the LBM kernel: 32 planes per block
    }

    {
        const size_t cpp = 172032;
        auto run = [&](const char *nm, auto launch, int nq) {
            const int planes = (int)(n / ((size_t)nq * cpp));
            const double t = time_it([&] { launch(planes); }, 5);
            printf("%-60s %7.0f GB/s\n", nm, 2.0 * 16 * nq * cpp * planes / t / 1e9); fflush(stdout);
        };
        run("F one-shot 19 streams, 512 thr", [&](int planes) { copy_streams_oneshot<19, 512><<<(unsigned)(cpp / 512 * planes), 512>>>(a, b, cpp, (int)(cpp / 512)); }, 19);
        run("F one-shot 19 streams, 256 thr", [&](int planes) { copy_streams_oneshot<19, 256><<<(unsigned)(cpp / 256 * planes), 256>>>(a, b, cpp, (int)(cpp / 256)); }, 19);
        run("F one-shot 4 streams, 256 thr", [&](int planes) { copy_streams_oneshot<4, 256><<<(unsigned)(cpp / 256 * planes), 256>>>(a, b, cpp, (int)(cpp / 256)); }, 4);
        run("F one-shot 1 stream, 256 thr", [&](int planes) { copy_streams_oneshot<1, 256><<<(unsigned)(cpp / 256 * planes), 256>>>(a, b, cpp, (int)(cpp / 256)); }, 1);
        run("F one-shot 8 streams, 256 thr", [&](int planes) { copy_streams_oneshot<8, 256><<<(unsigned)(cpp / 256 * planes), 256>>>(a, b, cpp, (int)(cpp / 256)); }, 8);
        for (int len : {82, 32, 8}) {
            char nm[96];
            snprintf(nm, 96, "G march 19 streams, 512 thr, %d planes per block", len);
            run(nm, [&](int planes) { copy_streams_march<19, 512><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            snprintf(nm, 96, "G march 19 streams, 256 thr, %d planes per block", len);
            run(nm, [&](int planes) { copy_streams_march<19, 256><<<(unsigned)(cpp / 256 * ((planes + len - 1) / len)), 256>>>(a, b, cpp, (int)(cpp / 256), planes, len); }, 19);
            snprintf(nm, 96, "H march pipelined 19 streams, 512 thr, %d planes per block", len);
            run(nm, [&](int planes) { copy_streams_march_pipe<19, 512><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
        }
        for (int len : {82, 32, 8, 1}) {
            char nm[96];
            snprintf(nm, 96, "I tiled march (19 runs back to back), 512 thr, %d planes/block", len);
            run(nm, [&](int planes) { copy_tiled_march<19, 512><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            snprintf(nm, 96, "I tiled march (19 runs back to back), 256 thr, %d planes/block", len);
            run(nm, [&](int planes) { copy_tiled_march<19, 256><<<(unsigned)(cpp / 256 * ((planes + len - 1) / len)), 256>>>(a, b, cpp, (int)(cpp / 256), planes, len); }, 19);
        }
        {
            const int len = 32;
            run("J groups of 19 (= G), 512 thr", [&](int planes) { copy_streams_groups<19, 512, 19, false><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            run("J groups of 5, 512 thr", [&](int planes) { copy_streams_groups<19, 512, 5, false><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            run("J groups of 2, 512 thr", [&](int planes) { copy_streams_groups<19, 512, 2, false><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            run("J groups of 1, 512 thr", [&](int planes) { copy_streams_groups<19, 512, 1, false><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            run("J groups of 19, nontemporal, 512 thr", [&](int planes) { copy_streams_groups<19, 512, 19, true><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
            run("J groups of 5, nontemporal, 512 thr", [&](int planes) { copy_streams_groups<19, 512, 5, true><<<(unsigned)(cpp / 512 * ((planes + len - 1) / len)), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); }, 19);
        }
    }
    {
        auto runk = [&](const char *nm, auto launch) { const double t = time_it(launch, 10); printf("%-60s %7.0f GB/s\n", nm, 2.0 * bytes / t / 1e9); fflush(stdout); };
        runk("K U=8 strided, 8 blocks/CU", [&] { copy_chunk_k<8, false><<<(unsigned)((n + 2047) / 2048), 256, 0>>>(a, b, n); });
        runk("K U=8 adjacent (128 B per thread), 8 blocks/CU", [&] { copy_chunk_k<8, true><<<(unsigned)((n + 2047) / 2048), 256, 0>>>(a, b, n); });
        runk("K U=8 strided, LDS-limited to 2 blocks/CU", [&] { copy_chunk_k<8, false><<<(unsigned)((n + 2047) / 2048), 256, 60000>>>(a, b, n); });
        runk("K U=8 strided, LDS-limited to 4 blocks/CU", [&] { copy_chunk_k<8, false><<<(unsigned)((n + 2047) / 2048), 256, 36000>>>(a, b, n); });
        runk("K U=1, LDS-limited to 2 blocks/CU", [&] { copy_chunk_k<1, false><<<(unsigned)((n + 255) / 256), 256, 60000>>>(a, b, n); });
        runk("K U=1, LDS-limited to 4 blocks/CU", [&] { copy_chunk_k<1, false><<<(unsigned)((n + 255) / 256), 256, 36000>>>(a, b, n); });
        runk("K U=1, 8 blocks/CU", [&] { copy_chunk_k<1, false><<<(unsigned)((n + 255) / 256), 256, 0>>>(a, b, n); });
        runk("K U=2 strided", [&] { copy_chunk_k<2, false><<<(unsigned)((n + 511) / 512), 256, 0>>>(a, b, n); });
        runk("K U=4 strided, LDS-limited to 2 blocks/CU", [&] { copy_chunk_k<4, false><<<(unsigned)((n + 1023) / 1024), 256, 60000>>>(a, b, n); });
        runk("K U=4 strided, 8 blocks/CU", [&] { copy_chunk_k<4, false><<<(unsigned)((n + 1023) / 1024), 256, 0>>>(a, b, n); });
        runk("K U=16 strided, LDS-limited to 1 block/CU", [&] { copy_chunk_k<16, false><<<(unsigned)((n + 4095) / 4096), 256, 100000>>>(a, b, n); });
        runk("K U=16 strided, LDS-limited to 2 blocks/CU", [&] { copy_chunk_k<16, false><<<(unsigned)((n + 4095) / 4096), 256, 60000>>>(a, b, n); });
    }
    {
        const size_t cpp = 172032; const int len = 32, nq = 19;
        const int planes = (int)(n / ((size_t)nq * cpp));
        auto runl = [&](const char *nm, auto launch) { const double t = time_it(launch, 5); printf("%-60s %7.0f GB/s\n", nm, 2.0 * 16 * nq * cpp * planes / t / 1e9); fflush(stdout); };
        const unsigned g512 = (unsigned)(cpp / 512 * ((planes + len - 1) / len)), g256 = (unsigned)(cpp / 256 * ((planes + len - 1) / len));
        runl("L double-buffered burst, 512 thr", [&] { copy_march_db<19, 512, false><<<g512, 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); });
        runl("L double-buffered staggered, 512 thr", [&] { copy_march_db<19, 512, true><<<g512, 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); });
        runl("L double-buffered burst, 256 thr", [&] { copy_march_db<19, 256, false><<<g256, 256>>>(a, b, cpp, (int)(cpp / 256), planes, len); });
        runl("L double-buffered staggered, 256 thr", [&] { copy_march_db<19, 256, true><<<g256, 256>>>(a, b, cpp, (int)(cpp / 256), planes, len); });
    }
    {
        const size_t cpp = 172032; const int len = 32, nq = 19;
        const int planes = (int)(n / ((size_t)nq * cpp));
        auto runm = [&](const char *nm, auto launch) { const double t = time_it(launch, 5); printf("%-60s %7.0f GB/s\n", nm, 2.0 * 16 * nq * cpp * planes / t / 1e9); fflush(stdout); };
        const int nch = (planes + len - 1) / len;
        runm("M split over 1 group  (512 thr = 512 cells)", [&] { copy_march_split<19, 512, 1><<<(unsigned)(cpp / 512 * nch), 512>>>(a, b, cpp, (int)(cpp / 512), planes, len); });
        runm("M split over 2 groups (512 thr = 256 cells)", [&] { copy_march_split<19, 512, 2><<<(unsigned)(cpp / 256 * nch), 512>>>(a, b, cpp, (int)(cpp / 256), planes, len); });
        runm("M split over 4 groups (512 thr = 128 cells)", [&] { copy_march_split<19, 512, 4><<<(unsigned)(cpp / 128 * nch), 512>>>(a, b, cpp, (int)(cpp / 128), planes, len); });
        runm("M split over 8 groups (512 thr = 64 cells)", [&] { copy_march_split<19, 512, 8><<<(unsigned)(cpp / 64 * nch), 512>>>(a, b, cpp, (int)(cpp / 64), planes, len); });
        runm("M split over 4 groups (1024 thr = 256 cells)", [&] { copy_march_split<19, 1024, 4><<<(unsigned)(cpp / 256 * nch), 1024>>>(a, b, cpp, (int)(cpp / 256), planes, len); });
        runm("M split over 2 groups (1024 thr = 512 cells)", [&] { copy_march_split<19, 1024, 2><<<(unsigned)(cpp / 512 * nch), 1024>>>(a, b, cpp, (int)(cpp / 512), planes, len); });
    }
    return 0;
}
