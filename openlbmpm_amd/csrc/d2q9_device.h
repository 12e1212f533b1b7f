// d2q9_device.h -- device helpers shared by the D2Q9 solvers (gfx950).
#pragma once
#include "lbmpm_common.h"

namespace lbmpm_dev {

__device__ __forceinline__ int wrapi(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }

__device__ __forceinline__ int wrapm(int v, int n)
{
    v %= n;
    return v < 0 ? v + n : v;
}
// v mod n for the coordinates a tile forms (tile origin < n, + at most a tile's region): on a lattice of at least 128 nodes they lie in
// [-n, 2n) and one conditional add does it; the integer division by the runtime n (~ 20 vector instructions) is left to the lattices
// smaller than a tile's region.  A wave-uniform choice, the same value either way.  For the kernels WITHOUT asm loads (sc2d.hip): the
// uniform branch it makes sits between the asm loads of rk2d_fused, where the static in-flight check (inflight.py) cannot follow it, and
// bought nothing there (c2, c4 within the noise; c3 - 1.1 %).
__device__ __forceinline__ int wrapm_fast(int v, int n)
{
    if (n >= 128) return wrapi(v, n);
    return wrapm(v, n);
}

// The two D2Q9 lattices of a solver (red/blue, component 0/1) are stored side by side,
// f[q][y*pitch + x] = {f_0, f_1}: one 16-byte access per direction moves both.
__host__ __device__ __forceinline__ size_t fslot(size_t plane, int q, size_t node, int c) { return ((size_t)q * plane + node) * 2 + (size_t)c; }
__device__ __forceinline__ void store_pair(double *f, size_t plane, int q, size_t node, double a, double b)
{
    double2 v;
    v.x = a; v.y = b;
    reinterpret_cast<double2 *>(f)[(size_t)q * plane + node] = v;
}

// The nine population pairs of a node.  stream: non-temporal stores -- what a step writes is read a whole lattice later, while the lines
// it READS twice (the halo nodes neighbouring tiles share) profit from every line of the XCD's L2 the stores leave alone: measured
// on 2048^2 lattices c3 0.254 -> 0.245 ms, c4 0.431 -> 0.413, c2 0.292 -> 0.284 (1024^2: 0.087 -> 0.085).  Off for lattices small
// enough to live in the caches from one step to the next (configs[0]).
// (STREAM is a template argument: behind a run-time flag hipcc merges the two store sequences and the non-temporal bit is lost.)
template <bool STREAM>
__device__ __forceinline__ void store_pairs(double *f, size_t plane, size_t node, const double a[9], const double b[9])
{
    // (plane base: uniform, 64 bits; the node inside a plane: one 32-bit byte offset for the nine stores)
    typedef double d2_t __attribute__((ext_vector_type(2)));
    char *base = reinterpret_cast<char *>(f);
    const unsigned off = (unsigned)node * 16u;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        d2_t v = {a[q], b[q]};
        d2_t *d = reinterpret_cast<d2_t *>(base + (size_t)q * plane * 16u + off);
        if (STREAM) __builtin_nontemporal_store(v, d);
        else *d = v;
    }
}

// Pull-streaming of the two lattices.
// Equivalent to the reference's push + in-place half-way bounce-back
// (AcceleratedRKGPU2D.py:340-417, OptimizedD2Q9GPU.py:452-550).  P needs the members
// nx, ny, pitch, plane, solidnbr, fin, first.
template <typename P>
__device__ __forceinline__ void pull_node(const P &p, int x, int y, double f0[9], double f1[9])
{
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY, OPP[9] = LBMPM_D2Q9_OPP;
    // p.first: the state is already "post-streaming" (initial condition) -> read in place
    const bool first = p.first != 0;
    // addresses as in pull_issue_asm below: the plane's 64-bit base (uniform) + a 32-bit byte offset inside the plane
    const unsigned pitch16 = (unsigned)p.pitch * 16u;
    unsigned xo[3], yo[3];                    // [1 + e]: column x - e, row y - e
    xo[1] = (unsigned)x * 16u; yo[1] = (unsigned)y * pitch16;
    xo[0] = first ? xo[1] : (unsigned)wrapi(x + 1, p.nx) * 16u; xo[2] = first ? xo[1] : (unsigned)wrapi(x - 1, p.nx) * 16u;
    yo[0] = first ? yo[1] : (unsigned)wrapi(y + 1, p.ny) * pitch16; yo[2] = first ? yo[1] : (unsigned)wrapi(y - 1, p.ny) * pitch16;
    const char *base = reinterpret_cast<const char *>(p.fin);
    const unsigned own = yo[1] + xo[1];
    const unsigned sn = first ? 0u : p.solidnbr[own >> 4];
    // All 9 loads are issued without waiting for the solid-neighbour byte (solid nodes hold
    // finite junk that is never used); the rare bounce-back links are patched afterwards.
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double2 v = *reinterpret_cast<const double2 *>(base + (size_t)i * p.plane * 16u + (yo[1 + EY[i]] + xo[1 + EX[i]]));
        f0[i] = v.x; f1[i] = v.y;
    }
    if (sn != 0) {
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const int o = OPP[i];
            if ((sn >> (o - 1)) & 1u) {        // x - e_i is solid: half-way bounce-back
                const double2 v = *reinterpret_cast<const double2 *>(base + (size_t)o * p.plane * 16u + own);
                f0[i] = v.x; f1[i] = v.y;
            }
        }
    }
}

// Loads as asm statements: absent from hipcc's s_waitcnt bookkeeping, and not split up or consumed one by one by its scheduler; the
// caller counts them and waits itself (LBMPM_VMCNT(n): at most n younger loads still in flight), then fences the registers.
typedef double lbmpm_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lbmpm_d2 asm_ld16(const void *a) { lbmpm_d2 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ double asm_ld8_nt(const void *a) { double v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ unsigned asm_ldu8(const void *a) { unsigned v; asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(a)); return v; }
// the same with a uniform 64-bit base (scalar registers) and a 32-bit byte offset per lane
__device__ __forceinline__ unsigned asm_ldu8(const void *base, unsigned off) { unsigned v; asm volatile("global_load_ubyte %0, %1, %2" : "=v"(v) : "v"(off), "s"(base)); return v; }
__device__ __forceinline__ double asm_ld8_nt(const void *base, unsigned off) { double v; asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(base)); return v; }
// s_waitcnt simm16 of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]; the other two counters left alone
#define LBMPM_VMCNT(n) (((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
// Head of a region guarded by `if (__ballot(c) != 0) { if (c) {` : states that some lane of the wave always enters.  hipcc still emits an
// s_cbranch_execz around the region; openlbmpm_amd/inflight.py (the static check of the asm loads in flight, tests/test_codeobj.py)
// drops that infeasible edge where it finds this comment in the assembly.
#define LBMPM_TAKEN asm volatile("; lbmpm-taken")
template <typename P>
__device__ __forceinline__ void pull_issue_asm(const P &p, int x, int y, lbmpm_d2 q[9])
{
    // address = <plane of direction i: 64-bit base in scalar registers> + <32-bit byte offset of the upstream node inside a plane>: the
    // three rows' and the three columns' offsets once (periodic), one v_add per direction -- instead of a 64-bit index and pointer per
    // direction (~ 80 -> ~ 35 vector instructions per node; a plane is < 4 GiB, checked at set-up)
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    // p.first: the state is already "post-streaming" (initial condition): read in place.  As a mask, not as a condition: hipcc turns a
    // uniform `first ? own : wrapped` into branches around the wrap arithmetic, and branches between asm loads are what the static
    // in-flight check (openlbmpm_amd/inflight.py) cannot follow
    const unsigned live = p.first != 0 ? 0u : ~0u;
    const unsigned pitch16 = (unsigned)p.pitch * 16u;
    unsigned xo[3], yo[3];                    // [1 + e]: column x - e, row y - e
    xo[1] = (unsigned)x * 16u; yo[1] = (unsigned)y * pitch16;
    xo[0] = xo[1] + (((unsigned)wrapi(x + 1, p.nx) * 16u - xo[1]) & live); xo[2] = xo[1] + (((unsigned)wrapi(x - 1, p.nx) * 16u - xo[1]) & live);
    yo[0] = yo[1] + (((unsigned)wrapi(y + 1, p.ny) * pitch16 - yo[1]) & live); yo[2] = yo[1] + (((unsigned)wrapi(y - 1, p.ny) * pitch16 - yo[1]) & live);
    const char *base = reinterpret_cast<const char *>(p.fin);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const unsigned off = yo[1 + EY[i]] + xo[1 + EX[i]];
        const char *bi = base + (size_t)i * p.plane * 16u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q[i]) : "v"(off), "s"(bi));
    }
}
// the bounce-back links of a node whose nine pairs pull_issue_asm fetched (sn: its solid-neighbour byte, 0 in the first step)
template <typename P>
__device__ __forceinline__ void pull_patch(const P &p, int x, int y, unsigned sn, double f0[9], double f1[9])
{
    constexpr int OPP[9] = LBMPM_D2Q9_OPP;
    if (sn != 0) {
        const unsigned own = ((unsigned)y * (unsigned)p.pitch + (unsigned)x) * 16u;       // (32-bit offset inside a plane, as in pull_issue_asm)
        const char *base = reinterpret_cast<const char *>(p.fin);
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const int o = OPP[i];
            if ((sn >> (o - 1)) & 1u) {        // x - e_i is solid: half-way bounce-back
                const double2 v = *reinterpret_cast<const double2 *>(base + (size_t)o * p.plane * 16u + own);
                f0[i] = v.x; f1[i] = v.y;
            }
        }
    }
}

__device__ __forceinline__ double sum9(const double f[9])
{   // accumulation order of calMacroDensityRKGPU2D / calFluidRhoGPU and of the ghost kernels
    double r = 0.;
#pragma unroll
    for (int i = 0; i < 9; ++i) r += f[i];
    return r;
}

// bit (i-1) of solidnbr <=> node + e_i is not fluid, periodic wrap on all four edges
// (the wrap of fillNeighboringNodes, AcceleratedRKGPU2D.py:25-28).
__global__ void setup_solidnbr(int nx, int ny, int pitch, const uint8_t *flags, uint8_t *solidnbr);

// XCD-aware tile index: workgroup b runs on XCD b % 8 (observed dispatch order); every XCD
// gets a contiguous band of tiles so halo rows are shared inside one L2.
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3;
    return k * q + (k < r ? k : r) + j;
}

// The same for lattices at least 32 tiles wide (2048 nodes): there the rows a tile row shares with the next one (its halo) have left
// the XCD's 4 MB L2 by the time the next tile row asks for them.  The whole tile rows of an XCD's share are walked in bands of four,
// column by column -- vertically adjacent tiles run close in time -- and every XCD starts at another column: all eight walking the same
// columns at once is 9 - 13 % SLOWER than whole rows (the same few channels from every XCD); staggered, c3 - 2 %, c4 - 3 % and 5 - 10 % fewer
// bytes fetched (profiles/r05_walk2d.txt).  Narrower lattices keep whole rows (c2 1024^2: + 4 % with bands).  A permutation of the tiles;
// results do not depend on it.
__device__ __forceinline__ int xcd_tile(int b, int nb, int tiles_x)
{
#if defined(LBMPM_DEV) && defined(LBMPM_WALK_BAND)
    constexpr int B = LBMPM_WALK_BAND;
#else
    constexpr int B = 4;
#endif
#if defined(LBMPM_DEV) && defined(LBMPM_WALK_ROWS)
    return xcd_tile(b, nb);
#endif
    if (tiles_x < 32) return xcd_tile(b, nb);
    const int q = nb >> 3, r = nb & 7, k = b & 7, j = b >> 3;
    const int base = k * q + (k < r ? k : r), cnt = q + (k < r ? 1 : 0);
    const int row0 = (base + tiles_x - 1) / tiles_x, row1 = (base + cnt) / tiles_x;      // the whole tile rows of this XCD's share
    const int lead = row0 * tiles_x - base;
    if (row1 <= row0 || j < lead || j >= lead + (row1 - row0) * tiles_x) return base + j;
    const int jj = j - lead, per = tiles_x * B;
    const int band = jj / per, w = jj - band * per;
    const int rows = min(B, row1 - row0 - band * B);
    const int col = (w / rows + k * tiles_x / 8) % tiles_x, row = w % rows;
    return (row0 + band * B + row) * tiles_x + col;
}

// Full-line stores: true when the 128-byte line (16 consecutive lanes = 16 doubles of a row) this
// lane would store into holds at least one active node.  Lanes of non-fluid cells of such a line
// store zeros into their (dead) slots with the same instruction, so the line leaves L2 complete:
// partially written lines cost the memory system a read-modify-write (measured on the D3Q19
// kernel: +35 % kernel time at porosity 0.65).  Must be reached by whole waves.
// LANES = lanes per 128-byte line: 8 for the 16-byte population pairs, 16 for 8-byte side arrays.
template <int LANES = 16>
__device__ __forceinline__ bool line_has_active(bool active, unsigned lane)
{
    const unsigned long long m = __ballot(active);
    return ((m >> (lane & (64u - LANES))) & ((1ull << LANES) - 1ull)) != 0;
}

}  // namespace lbmpm_dev
