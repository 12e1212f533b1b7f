// "Missing 2" of the round-4 review: fluid-cells-only storage for the porous 2-D lattices, with the row-segment records read by SCALAR loads
// (the 3-D kernel's scheme) so that the look-up leaves the vector memory path -- the form that was not tried when compact storage lost in
// round 2 (slot table: + 8 %; LDS look-ups in front of every pull: + 21 %).  This is the skeleton of rk2d_fused's access shape (64 x 8 tile,
// 512 threads, one own node per lane + one halo node of the three rings for 468 lanes, three barriers, a stretch of dependent fp64 FMAs,
// nine 16-byte stores) on a 2048^2 lattice with a porous mask (discs r 6 - 20, porosity 0.65, as bench.py's c3 / c4) in two layouts:
//
//   dense    the shipped layout [9][ny][nx] of 16-byte pairs: region flags (bytes) and the own node's nine pulls go out together, ungated
//            (a lane on a solid node reads junk); flags -> LDS -> barrier; halo pulls where fluid; stores of every 128-byte line that holds
//            a fluid node (solid lanes write zeros).  NOT included: the patch loads of the product for upstream-solid directions (the
//            skeleton takes the junk) -- the comparison leans towards the dense layout.
//   compact  fluid cells only, numbered tile by tile and row by row inside a tile, tile runs padded to whole 128-byte lines; per (row,
//            64-node segment) one 16-byte record {fluid mask, slot of its first fluid cell, end of the tile's run}.  A wave owns a tile row:
//            the nine records around it (3 rows x 3 segments) are wave-uniform scalar loads, slot = first + popcount(mask below the lane) is
//            v_mbcnt on the (shifted) mask, "upstream solid" a bit test that redirects the pull to the own cell's opposite direction.  The
//            halo lanes take their records from an LDS copy of the 16 x 3 records around the tile (one vector load each for 48 lanes,
//            published by the barrier that publishes the flags in the dense form) and count bits with v_bcnt.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/tile_compact.hip -o tools/microbench/tc && tools/microbench/tc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
constexpr int Q = 9, TW = 64, TH = 8, H = 3, RW = TW + 2 * H, RH = TH + 2 * H, NHALO = 2 * H * RW + 2 * H * TH;
__device__ __forceinline__ int xcd_tile(int b, int nb) { const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3; return k * q + (k < r ? k : r) + j; }
__device__ __forceinline__ int wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
__device__ __forceinline__ void halo_cell(int n, int &rx, int &ry)
{
    if (n < H * RW) { ry = n / RW; rx = n % RW; }
    else if ((n -= H * RW) < H * RW) { ry = RH - H + n / RW; rx = n % RW; }
    else { n -= H * RW; ry = H + n / (2 * H); const int c = n % (2 * H); rx = c < H ? c : RW - 2 * H + c; }
}
struct Rec { unsigned long long mask; unsigned first, tile_end; };
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_nt(double2 *p, double2 v) { d2 w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<d2 *>(p)); }

template <int WORK>
__device__ __forceinline__ void work(double2 v[Q], double2 acc)
{
#pragma unroll 1
    for (int w = 0; w < WORK / (2 * Q); ++w) {
#pragma unroll
        for (int i = 0; i < Q; ++i) { v[i].x = fma(v[i].x, 1.0000001, acc.y * 1e-300); v[i].y = fma(v[i].y, 0.9999999, acc.x * 1e-300); }
    }
}

template <int WORK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void dense(const double2 *__restrict__ in, double2 *__restrict__ out, const uint8_t *__restrict__ fl,
                                                                                          int nx, int ny, size_t plane, int tiles_x)
{
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    __shared__ uint8_t s_fluid[RH * RW];
    __shared__ double s[RH * RW];
    const int t = xcd_tile(blockIdx.x, gridDim.x), tid = threadIdx.x;
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH, lx = tid & 63, ly = tid >> 6, x = tx0 + lx, y = ty0 + ly;
    unsigned f0, f1;
    {
        const int n0 = tid, n1 = min(tid + 512, RH * RW - 1);
        f0 = fl[(size_t)wrap(ty0 - H + n0 / RW, ny) * nx + wrap(tx0 - H + n0 % RW, nx)];
        f1 = fl[(size_t)wrap(ty0 - H + n1 / RW, ny) * nx + wrap(tx0 - H + n1 % RW, nx)];
    }
    double2 v[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] = in[i * plane + (size_t)wrap(y - EY[i], ny) * nx + wrap(x - EX[i], nx)];
    s_fluid[tid] = (uint8_t)f0;
    if (tid + 512 < RH * RW) s_fluid[tid + 512] = (uint8_t)f1;
    __syncthreads();
    double2 acc = {0., 0.};
    if (tid < NHALO) {
        int rx, ry;
        halo_cell(tid, rx, ry);
        if (s_fluid[ry * RW + rx]) {
            const int hx = wrap(tx0 - H + rx, nx), hy = wrap(ty0 - H + ry, ny);
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const double2 w = in[i * plane + (size_t)wrap(hy - EY[i], ny) * nx + wrap(hx - EX[i], nx)];
                acc.x += w.x; acc.y += w.y;
            }
            s[ry * RW + rx] = acc.x + acc.y;
        }
    }
    const bool fluid = s_fluid[(ly + H) * RW + lx + H];
    if (fluid) s[(ly + H) * RW + lx + H] = v[0].x + v[1].y;
    __syncthreads();
    if (fluid) acc.x += s[(ly + H) * RW + lx + H - 1] + s[(ly + H + 1) * RW + lx + H];
    __syncthreads();
    if (fluid) { acc.y += s[(ly + H) * RW + lx + H + 1]; work<WORK>(v, acc); v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y; }
    else {
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = double2{0., 0.};
    }
    const unsigned long long m = __ballot(fluid);
    const bool line8 = ((m >> (lx & 56)) & 0xffull) != 0;          // the 128-byte line (8 lanes of 16 bytes) holds a fluid node
    if (line8) {
#pragma unroll
        for (int i = 0; i < Q; ++i) st_nt(&out[i * plane + (size_t)y * nx + x], v[i]);
    }
}

// a record through the constant address space: hipcc emits s_load_dwordx4 for a wave-uniform address
__device__ __forceinline__ Rec ldrec_s(const Rec *p)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 q = *reinterpret_cast<const u4 __attribute__((address_space(4))) *>((uintptr_t)p);
    Rec r;
    r.mask = (unsigned long long)q.x | ((unsigned long long)q.y << 32); r.first = q.z; r.tile_end = q.w;
    return r;
}
__device__ __forceinline__ unsigned below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

template <int WORK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void compact(const double2 *__restrict__ in, double2 *__restrict__ out, const Rec *__restrict__ rec,
                                                                                            int nx, int ny, int nseg, size_t plane /* slots */, int tiles_x)
{
    constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1}, OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
    __shared__ Rec s_rec[RH + 2][3];
    __shared__ double s[RH * RW];
    const int t = xcd_tile(blockIdx.x, gridDim.x), tid = threadIdx.x;
    const int tx = t % tiles_x, ty = t / tiles_x, tx0 = tx * TW, ty0 = ty * TH, lx = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), ly = wave;
    const int sm = tx > 0 ? tx - 1 : nseg - 1, sp = tx + 1 < nseg ? tx + 1 : 0;
    // the records of the region's rows (+ one more row either side: the halo nodes' upstream rows), three segments each: 48 lanes, one 16-byte load
    if (tid < (RH + 2) * 3) {
        const int rr = tid / 3, k = tid % 3;
        s_rec[rr][k] = rec[(size_t)wrap(ty0 - H - 1 + rr, ny) * nseg + (k == 0 ? sm : (k == 1 ? tx : sp))];
    }
    // the nine records around this wave's row: wave-uniform addresses -> scalar loads
    Rec R[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const size_t row = (size_t)wrap(ty0 + ly - 1 + a, ny) * nseg;
        const unsigned i0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(row + sm)), i1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(row + tx)),
                       i2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(row + sp));
        R[a][0] = ldrec_s(rec + i0); R[a][1] = ldrec_s(rec + i1); R[a][2] = ldrec_s(rec + i2);
    }
    const bool fluid = (R[1][1].mask >> lx) & 1ull;
    const unsigned own = R[1][1].first + below(R[1][1].mask);
    double2 v[Q];
    if (fluid) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const Rec &c = R[1 - EY[i]][1];
            unsigned slot;
            bool ok;
            if (EX[i] == 0) { slot = c.first + below(c.mask); ok = (c.mask >> lx) & 1ull; }
            else if (EX[i] > 0) {                       // source column lx - 1
                slot = c.first + below(c.mask << 1); ok = ((c.mask << 1) >> lx) & 1ull;
                if (lx == 0) { const Rec &p = R[1 - EY[i]][0]; slot = p.first + (unsigned)__popcll(p.mask) - 1u; ok = p.mask >> 63; }
            } else {                                    // source column lx + 1
                slot = c.first + below(c.mask >> 1) + (unsigned)(c.mask & 1ull); ok = ((c.mask >> 1) >> lx) & 1ull;
                if (lx == 63) { const Rec &n = R[1 - EY[i]][2]; slot = n.first; ok = n.mask & 1ull; }
            }
            v[i] = ok ? in[i * plane + slot] : in[OPP[i] * plane + own];
        }
    }
    __syncthreads();
    double2 acc = {0., 0.};
    if (tid < NHALO) {
        int rx, ry;
        halo_cell(tid, rx, ry);
        const int gx = tx0 - H + rx;                     // may leave the tile's segment by up to three nodes
        const int k0 = gx < tx0 ? 0 : (gx >= tx0 + TW ? 2 : 1), b0 = gx & 63;
        const Rec me = s_rec[ry + 1][k0];
        if ((me.mask >> b0) & 1ull) {
            const unsigned hown = me.first + (unsigned)__popcll(me.mask & ((1ull << b0) - 1ull));
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                const int xs = gx - EX[i];
                const int k = xs < tx0 ? 0 : (xs >= tx0 + TW ? 2 : 1), b = xs & 63;
                const Rec c = s_rec[ry + 1 - EY[i]][k];
                const bool ok = (c.mask >> b) & 1ull;
                const unsigned slot = c.first + (unsigned)__popcll(c.mask & ((1ull << b) - 1ull));
                const double2 w = ok ? in[i * plane + slot] : in[OPP[i] * plane + hown];
                acc.x += w.x; acc.y += w.y;
            }
            s[ry * RW + rx] = acc.x + acc.y;
        }
    }
    if (fluid) s[(ly + H) * RW + lx + H] = v[0].x + v[1].y;
    __syncthreads();
    if (fluid) acc.x += s[(ly + H) * RW + lx + H - 1] + s[(ly + H + 1) * RW + lx + H];
    __syncthreads();
    if (fluid) {
        acc.y += s[(ly + H) * RW + lx + H + 1]; work<WORK>(v, acc); v[0].x += 1e-300 * acc.x; v[0].y += 1e-300 * acc.y;
#pragma unroll
        for (int i = 0; i < Q; ++i) st_nt(&out[i * plane + own], v[i]);
    } else if (wave == TH - 1) {
        // the idle lanes of the tile's last row complete the last line of its run with zeros
        const unsigned end = R[1][1].first + (unsigned)__popcll(R[1][1].mask), rank = below(~R[1][1].mask);
        if (end + rank < R[1][1].tile_end) {
#pragma unroll
            for (int i = 0; i < Q; ++i) out[i * plane + end + rank] = double2{0., 0.};
        }
    }
}

int main()
{
    const int nx = 2048, ny = 2048, nseg = nx / 64, tiles_x = nx / TW, tiles_y = ny / TH, ntiles = tiles_x * tiles_y;
    // porous mask: discs of radius 6 - 20 until 35 % of the nodes are solid (periodic)
    std::vector<uint8_t> fl((size_t)nx * ny, 1);
    {
        unsigned long long st = 20260929ull;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; };
        size_t solid = 0;
        while (solid < (size_t)(0.35 * nx * ny)) {
            const double cx = rnd() * nx, cy = rnd() * ny, r = 6. + 14. * rnd();
            for (int dy = -(int)r - 1; dy <= (int)r + 1; ++dy)
                for (int dx = -(int)r - 1; dx <= (int)r + 1; ++dx)
                    if (dx * dx + dy * dy < r * r) {
                        const int x = ((int)cx + dx + nx) % nx, y = ((int)cy + dy + ny) % ny;
                        if (fl[(size_t)y * nx + x]) { fl[(size_t)y * nx + x] = 0; ++solid; }
                    }
        }
    }
    size_t nfluid = 0;
    for (auto f : fl) nfluid += f;
    // compact numbering
    std::vector<Rec> rec((size_t)ny * nseg);
    size_t slots = 0;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            const size_t base = slots;
            size_t n = 0;
            for (int r = 0; r < TH; ++r) {
                unsigned long long m = 0;
                const int y = ty * TH + r;
                for (int b = 0; b < 64; ++b) if (fl[(size_t)y * nx + tx * 64 + b]) m |= 1ull << b;
                rec[(size_t)y * nseg + tx].mask = m; rec[(size_t)y * nseg + tx].first = (unsigned)(base + n);
                n += __builtin_popcountll(m);
            }
            slots = (base + n + 7) & ~(size_t)7;
            for (int r = 0; r < TH; ++r) rec[(size_t)(ty * TH + r) * nseg + tx].tile_end = (unsigned)slots;
        }
    printf("2048^2, %zu fluid nodes (%.1f %%), %zu slots (padding %.2f %%)\n", nfluid, 100. * nfluid / ((double)nx * ny), slots, 100. * (slots - nfluid) / (double)nfluid);
    const size_t plane_d = (size_t)nx * ny, plane_c = slots;
    double2 *a, *b, *ca, *cb; uint8_t *dfl; Rec *drec;
    CK(hipMalloc(&a, plane_d * Q * sizeof(double2))); CK(hipMalloc(&b, plane_d * Q * sizeof(double2)));
    CK(hipMalloc(&ca, plane_c * Q * sizeof(double2))); CK(hipMalloc(&cb, plane_c * Q * sizeof(double2)));
    CK(hipMalloc(&dfl, fl.size())); CK(hipMalloc(&drec, rec.size() * sizeof(Rec)));
    CK(hipMemcpy(dfl, fl.data(), fl.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(drec, rec.data(), rec.size() * sizeof(Rec), hipMemcpyHostToDevice));
    {   // the same values in both layouts: value of (direction, node) = a function of both
        std::vector<double2> hd(plane_d * Q), hc(plane_c * Q, double2{0., 0.});
        for (int i = 0; i < Q; ++i)
            for (int y = 0; y < ny; ++y)
                for (int tx = 0; tx < nseg; ++tx) {
                    const Rec &r = rec[(size_t)y * nseg + tx];
                    unsigned j = r.first;
                    for (int bb = 0; bb < 64; ++bb) {
                        const size_t node = (size_t)y * nx + tx * 64 + bb;
                        const double2 val = {1e-3 * (double)((node * 7 + i * 13) % 1009), 1e-3 * (double)((node * 3 + i * 5) % 917)};
                        hd[i * plane_d + node] = fl[node] ? val : double2{0., 0.};
                        if (fl[node]) hc[i * plane_c + j++] = val;
                    }
                }
        CK(hipMemcpy(a, hd.data(), hd.size() * sizeof(double2), hipMemcpyHostToDevice)); CK(hipMemcpy(ca, hc.data(), hc.size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    CK(hipMemset(b, 0, plane_d * Q * sizeof(double2))); CK(hipMemset(cb, 0, plane_c * Q * sizeof(double2)));
    // one application each with no arithmetic: the fluid nodes' direction-3 values must agree (the dense form pulls junk where the upstream node
    // is solid, the compact one bounces back: compare the nodes whose W neighbour is fluid)
    dense<0><<<ntiles, 512>>>(a, b, dfl, nx, ny, plane_d, tiles_x);
    compact<0><<<ntiles, 512>>>(ca, cb, drec, nx, ny, nseg, plane_c, tiles_x);
    CK(hipDeviceSynchronize());
    {
        std::vector<double2> hd(plane_d), hc(plane_c);
        CK(hipMemcpy(hd.data(), b + 1 * plane_d, plane_d * sizeof(double2), hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), cb + 1 * plane_c, plane_c * sizeof(double2), hipMemcpyDeviceToHost));
        size_t cmp = 0, bad = 0;
        for (int y = 0; y < ny; ++y)
            for (int tx = 0; tx < nseg; ++tx) {
                const Rec &r = rec[(size_t)y * nseg + tx];
                unsigned j = r.first;
                for (int bb = 0; bb < 64; ++bb) {
                    const size_t node = (size_t)y * nx + tx * 64 + bb;
                    if (!fl[node]) continue;
                    const size_t up = (size_t)y * nx + (tx * 64 + bb - 1 + nx) % nx;          // direction 1 comes from x - 1
                    if (fl[up]) { ++cmp; if (hd[node].y != hc[j].y) ++bad; }
                    ++j;
                }
            }
        printf("direction 1 after one application, nodes with a fluid upstream node: %zu compared, %zu differ\n", cmp, bad);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = 2.0 * nfluid * Q * sizeof(double2);
    auto run = [&](const char *name, auto launch, auto &p, auto &q) -> int {
        for (int w = 0; w < 3; ++w) { launch(p, q); launch(q, p); }
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int w = 0; w < 10; ++w) { launch(p, q); launch(q, p); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%-44s %.4f ms   %.2f TB/s of the fluid nodes' bytes (%.3f GB)\n", name, best / 20, alg / (best / 20 * 1e-3) / 1e12, alg / 1e9);
        return 0;
    };
#define DENSE(W) if (run("dense,   " #W " FMAs", [&](double2 *i, double2 *o) { dense<W><<<ntiles, 512>>>(i, o, dfl, nx, ny, plane_d, tiles_x); }, a, b)) return 1;
#define COMP(W) if (run("compact, " #W " FMAs", [&](double2 *i, double2 *o) { compact<W><<<ntiles, 512>>>(i, o, drec, nx, ny, nseg, plane_c, tiles_x); }, ca, cb)) return 1;
    DENSE(0) COMP(0) DENSE(540) COMP(540) DENSE(1080) COMP(1080) DENSE(2160) COMP(2160)
    return 0;
}
