// rk3d.hip -- D3Q19 colour-gradient (perturbation operator) time stepper for gfx950, written
// for z-slab decomposition across the GPUs of a node.
//
// The reference ships only an ini for this model (IniFiles/RKtwophasesetup3D.ini; the module
// RKColorGradientD3Q19 imported by main.py:22 is absent), so the model is the D3Q19 extension
// of the reference's 2-D kernels, operator by operator (list + citations in
// oracle/rk3d_oracle.c and DESIGN.md).  Pinned to the reference by reduction: a y-uniform lattice through
// these kernels reproduces the captures of the reference's real D2Q9 perturbation driver (RKD2Q9.py:978-1223)
// to 3e-13 (tests/test_rk3d_reduction.py), and pinned in full 3-D against oracle/rk3d_oracle.c.  That driver
// does not run as shipped; the captures come from it with four call-site repairs listed in every fixture
// (tests/golden/gen/make_golden_rk_pert.py).  Three pad or drop arguments; R3 MOVES calTotalFluidPDF from
// right after streaming (RKD2Q9.py:1065, where the SRT branch would collide a total that the boundary kernels
// have not touched yet) to after collision 1, which changes the numbers the loop produces: the pin is to that
// repaired loop, not to an executable original.
//
// Storage of the populations (compact layout): rk3dq.h keeps 19 colour-blind populations + k_R + the recolouring
// vector per cell (the default), the kernels below both colour lattices (LBMPM_RK3D_STORAGE=38, the cross-check).
//
// Two storage layouts per rank (zl = 0..nzl+1: owned planes 1..nzl + one halo plane on each side that
// holds the neighbour rank's outermost plane; only the five populations that cross the cut are
// ever filled / read there; x, y periodic, z not: planes 0 and nz-1 of the global lattice are
// boundary ghost planes):
//   compact (default when nx % 64 == 0): fluid cells only, f[zl][q][j] = {red, blue}, see
//       "compact storage" below -> kernels rk3dc_*
//   dense: plane-major SoA f[zl][colour][q][y][x] -> kernels rk3d_*
// and two schedules: fused (default; one z-marching kernel per step that keeps the phase field in
// an LDS ring) and split (variant 1, dense only: phase_field sweep + collide sweep).
// One time step of a slab = [f halo exchange] -> phase field of the face planes -> [phi halo
// exchange] -> collide; lbmpm_rk3d_collide_interior / _boundary overlap the exchanges with the
// planes that do not depend on them.
#include <type_traits>
#include "lbmpm_common.h"
#include "d2q9_device.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

using lbmpm::set_error;
using lbmpm_dev::wrapi;

constexpr int Q = 19;
#define LBMPM_D3Q19_CX {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0}
#define LBMPM_D3Q19_CY {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1}
#define LBMPM_D3Q19_CZ {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1}
#define LBMPM_D3Q19_OPP {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17}
__device__ __forceinline__ constexpr double wq(int i) { return i == 0 ? 1. / 3. : (i < 7 ? 1. / 18. : 1. / 36.); }
__device__ __forceinline__ constexpr double bq(int i) { return i == 0 ? -1. / 3. : (i < 7 ? 1. / 18. : 1. / 36.); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct RK3Dev {
    int nx, ny, nzl, pitch;
    unsigned plane_bytes;        // bytes of one [y][x] plane of doubles
    size_t plane2, vol;
    int z0, nzg;                 // global z of local plane zl is z0 + zl - 1
    const uint8_t *flags;        // [vol]
    const uint32_t *solidnbr;    // [vol], bit (i-1) <=> node + e_i is not fluid
    const double *fin;           // [nzl+2][2][Q][plane2]
    double *fout;
    double *phi;                 // [vol]; non-fluid cells hold solidPhi
    double *diag;                // [5][vol] rhoR, rhoB, vx, vy, vz or nullptr
    double ak, beta, cR, cB, solidPhi, vzR, vzB, rhoOutR, rhoOutB;   // cX = 1 / (2 (tauX - 1/2))
    double rcA, rcD;             // beta w_i / |e_i| of the recolouring term for |e_i| = 1 and sqrt 2 (lbmpm_rk3d_config::recolor_*)
    int first, fill;
    int mrt;                     // 0: BGK, 1: MRT (d'Humieres D3Q19 basis) on the colour-blind populations
    // compact storage (layout 1): only fluid cells are stored, see "compact storage" below
    const u32x4 *seg;                 // [rows][nseg] {fluid mask lo, hi, j of the first fluid cell, fluid(x0-1) | fluid(x0+w) << 1 | pad << 2 | (64-w) << 8}, rows = (nzl+2)*ny
    const u32x4 *seg2;                // [rows][nseg] {j of cell x0-1, j of cell x0+w (periodic), -, -}; w = cells of the segment (make_row)
    const unsigned long long *pstart; // [nzl+3] fluid cells before plane zl
    int nseg;
    const uint32_t *pur_in;           // [rows][nseg] row flags of the q23 storage (rk3dq.h), ping-pong with fin / fout
    uint32_t *pur_out;
#ifdef LBMPM_DEV                       // development builds only (tools/dev/devlib.py); the product library has neither
    unsigned long long *trace;        // LBMPM_RK3D_TRACE: four words per workgroup of rk3dq_fused (start, prologue done, end, planes)
#endif
    int halo_lo;                      // planes zl <= halo_lo are not this launch's to compute: their phase field is read from `phi`, their stored state
                                      // pulled from as it stands -- 0: the halo plane below the slab; 3 on the lattice's bottom rank with the
                                      // convective outlet (the planes z = 0, 1, 2 take the streamed populations of plane 3: rk3d_state.h / conv kernels)
    int conv;                         // [BoundaryCondition] BoundaryTypeOutlet = 'Convective' (A:700-784 as z planes): the planes z = 2, 1, 0 take the
                                      // streamed populations of plane 3 and re-sum their densities, no pressure rule on plane 1 (source_plane_x)
    int inletP;                       // 0 velocity inlet: vzR, vzB are velocities; 1 pressure inlet per colour: vzR, vzB hold densityRH, densityBH
                                      // (one pair of kernel arguments for both, and this word behind everything else: the marching kernel's
                                      // argument loads and registers stay what they were -- it has no scalar register to spare)
};

// ---- addressing.  Populations are stored plane-major, f[zl][colour][q][y][x]: everything a node
// touches in one step (38 populations x 3 planes) lies within 114 consecutive planes, so every
// access is <uniform 64-bit base in SGPRs> + <32-bit byte offset in one VGPR>; the nine in-plane
// offsets of a node's 3 x 3 neighbourhood are computed once.
struct Cell { unsigned o[3][3]; };     // o[1 + dy][1 + dx] = byte offset of (y + dy, x + dx) inside a plane, periodic

__device__ __forceinline__ Cell make_cell(const RK3Dev &p, int x, int y)
{
    Cell c;
    const int xs[3] = {wrapi(x - 1, p.nx), x, wrapi(x + 1, p.nx)}, ys[3] = {wrapi(y - 1, p.ny), y, wrapi(y + 1, p.ny)};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) c.o[a][b] = (unsigned)(ys[a] * p.pitch + xs[b]) * 8u;
    return c;
}

// a per-iteration copy the optimiser cannot prove loop-invariant: keeps the 2 x 18 derived load
// offsets from being hoisted out of the march loop into (and held in) ~70 registers
__device__ __forceinline__ Cell fresh(const Cell &c)
{
    Cell r = c;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) asm volatile("" : "+v"(r.o[a][b]));
    return r;
}

__device__ __forceinline__ const char *plane_ptr(const double *f, const RK3Dev &p, int zl)
{
    return reinterpret_cast<const char *>(f) + (size_t)zl * (2 * Q) * p.plane_bytes;
}
__device__ __forceinline__ double ldg(const char *uniform_base, unsigned off) { return *reinterpret_cast<const double *>(uniform_base + off); }
__device__ __forceinline__ void stg(char *uniform_base, unsigned off, double v) { *reinterpret_cast<double *>(uniform_base + off) = v; }

// Pull streaming with half-way bounce-back folded into the address: where the upstream cell is
// not fluid the lane reads the opposite population of its own cell instead (one load per
// direction either way, no divergent second pass).  sn = solidnbr word of the node.
template <bool FIRST>   // FIRST: the state given by set_density has not been streamed yet, "pull" in place
__device__ __forceinline__ void pull3(const RK3Dev &p, const Cell &c, int zl, unsigned sn, double fR[Q], double fB[Q])
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ, OPP[Q] = LBMPM_D3Q19_OPP;
    const unsigned pb = p.plane_bytes, own = c.o[1][1];
    const char *red = plane_ptr(p.fin, p, zl - 1), *blue = red + (size_t)Q * pb;   // slots 0..113 = planes zl-1, zl, zl+1
    fR[0] = ldg(red, (unsigned)(2 * Q) * pb + own);
    fB[0] = ldg(blue, (unsigned)(2 * Q) * pb + own);
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const unsigned up = (unsigned)((1 - CZ[i]) * 2 * Q + i) * pb + c.o[1 - CY[i]][1 - CX[i]];
        const unsigned back = (unsigned)(2 * Q + OPP[i]) * pb + own;
        const unsigned s = FIRST ? (unsigned)(2 * Q + i) * pb + own : (((sn >> (OPP[i] - 1)) & 1u) ? back : up);
        fR[i] = ldg(red, s);
        fB[i] = ldg(blue, s);
    }
}

// solidnbr word (bit 31 = the cell itself is fluid) of the cell with in-plane byte offset own
__device__ __forceinline__ unsigned load_meta(const RK3Dev &p, int zl, unsigned own)
{
    return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(p.solidnbr + (size_t)zl * p.plane2) + (own >> 1));
}

// plane the populations of plane zl are pulled around: the ghost planes copy their neighbour
__device__ __forceinline__ int source_plane(const RK3Dev &p, int zl)
{
    const int zg = p.z0 + zl - 1;
    return zg == p.nzg - 1 ? zl - 1 : (zg == 0 ? zl + 1 : zl);
}

// ... with the convective outlet (the 38-value dense kernels, the diagnostics and set-up kernels; NOT the marching kernel of the
// 23-value storage, which never computes those planes: rk3dq_conv_*)
__device__ __forceinline__ int source_plane_x(const RK3Dev &p, int zl)
{
    const int zg = p.z0 + zl - 1;
    if (p.conv && zg <= 2) return zl + (3 - zg);
    return source_plane(p, zl);
}

__device__ __forceinline__ double sum19(const double f[Q])
{
    double r = 0.;
#pragma unroll
    for (int i = 0; i < Q; ++i) r += f[i];
    return r;
}

// Zou-He velocity inlet, top plane, unknown e_z = -1 (Hecht & Harting 2010; 2-D analogue
// AcceleratedRKGPU2D.py:657-695)
__device__ __forceinline__ double zouhe_inlet(double uz, double f[Q])
{
    const double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    const double sp = f[5] + f[11] + f[14] + f[15] + f[18];
    const double rho = (s0 + 2. * sp) / (1. + uz);
    const double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    const double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[6] = f[5] - 1. / 3. * rho * uz;
    f[12] = f[11] - 1. / 6. * rho * uz + Nx;
    f[13] = f[14] - 1. / 6. * rho * uz - Nx;
    f[16] = f[15] - 1. / 6. * rho * uz + Ny;
    f[17] = f[18] - 1. / 6. * rho * uz - Ny;
    return rho;
}

// Zou-He pressure inlet, top plane, unknown e_z = -1 (2-D analogue A:925-962): u_z = -1 + (S0 + 2 S+) / rho, then the closure above
__device__ __forceinline__ void zouhe_inlet_pressure(double rho, double f[Q])
{
    const double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    const double sp = f[5] + f[11] + f[14] + f[15] + f[18];
    const double uz = -1. + (s0 + 2. * sp) / rho;
    const double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    const double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[6] = f[5] - 1. / 3. * rho * uz;
    f[12] = f[11] - 1. / 6. * rho * uz + Nx;
    f[13] = f[14] - 1. / 6. * rho * uz - Nx;
    f[16] = f[15] - 1. / 6. * rho * uz + Ny;
    f[17] = f[18] - 1. / 6. * rho * uz - Ny;
}

// Zou-He pressure outlet, bottom plane, unknown e_z = +1 (2-D analogue A:1008-1039)
__device__ __forceinline__ void zouhe_outlet(double rho, double f[Q])
{
    const double s0 = f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10];
    const double sm = f[6] + f[12] + f[13] + f[16] + f[17];
    const double uz = 1. - 1. / rho * (s0 + 2. * sm);
    const double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
    const double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
    f[5] = f[6] + 1. / 3. * rho * uz;
    f[11] = f[12] + 1. / 6. * rho * uz - Nx;
    f[14] = f[13] + 1. / 6. * rho * uz + Nx;
    f[15] = f[16] + 1. / 6. * rho * uz - Ny;
    f[18] = f[17] + 1. / 6. * rho * uz + Ny;
}

// boundary planes: Zou-He per colour on the pulled populations of plane zl (pulled around source_plane(zl))
__device__ __forceinline__ void finish_state3(const RK3Dev &p, int zl, double fR[Q], double fB[Q], double &rR, double &rB)
{
    const int zg = p.z0 + zl - 1, zsg = p.z0 + source_plane_x(p, zl) - 1;
    rR = sum19(fR);
    rB = sum19(fB);
    if (zsg == p.nzg - 2) {
        if (p.inletP) {                // the ghost plane copies populations and densities (A:968-1002)
            zouhe_inlet_pressure(p.vzR, fR); rR = p.vzR;
            zouhe_inlet_pressure(p.vzB, fB); rB = p.vzB;
        } else {
            rR = zouhe_inlet(p.vzR, fR);
            rB = zouhe_inlet(p.vzB, fB);
            if (zg == p.nzg - 1) { rR = sum19(fR); rB = sum19(fB); }
        }
    }
    if (zsg == 1) {
        zouhe_outlet(p.rhoOutR, fR); rR = p.rhoOutR;
        zouhe_outlet(p.rhoOutB, fB); rB = p.rhoOutB;
    }
}

// post-streaming, post-boundary state of the node with in-plane neighbourhood c on plane zl
__device__ __forceinline__ void node_state3(const RK3Dev &p, const Cell &c, int zl, double fR[Q], double fB[Q], double &rR,
                                            double &rB)
{
    const int zs = source_plane_x(p, zl);
    if (p.first) pull3<true>(p, c, zs, 0u, fR, fB);
    else pull3<false>(p, c, zs, load_meta(p, zs, c.o[1][1]), fR, fB);
    finish_state3(p, zl, fR, fB, rR, rB);
}


// ======================================================================= compact storage
// Only fluid cells are stored.  Cells are numbered plane by plane, inside a plane tile by tile
// (64 x 8 cells, the footprint of a block of the marching kernel), inside a tile row by row in x
// order (j = index inside the plane); every tile's run starts on a 128-byte line and is padded to
// a whole number of lines, so a block writes complete lines only (partial lines cost a
// read-modify-write, see collide_store).  Populations lie plane-major with the two colours side by
// side, f[zl][q][j] = {red, blue}: as in the dense layout everything a node touches sits within 57
// population blocks of three consecutive planes and is addressed as <uniform base> + <32-bit
// offset>, and one 16-byte access per direction moves both colours.  No neighbour table: per row
// segment of 64 cells there is one 64-bit fluid mask and the j of its first fluid cell, and
// j(x) = first + popcount(mask below x).  A wave owns one row segment, so the words are
// wave-uniform: the popcount is v_mbcnt and "is the upstream cell fluid" is the mask itself used
// as a lane mask.  HBM moves fluid cells only.
constexpr int DIRT[27] = {-1, 16, -1, 12, 6, 13, -1, 17, -1, 8, 4, 9, 2, 0, 1, 10, 3, 7, -1, 18, -1, 14, 5, 11, -1, 15, -1};   // [(cz+1)*9 + (cy+1)*3 + cx+1]
constexpr int TILE_ROWS = 8;        // rows per tile of the cell numbering

struct RowTab {
    unsigned long long m;           // fluid bits of the segment
    unsigned first;                 // j of the segment's first fluid cell
    unsigned lbit, rbit, jl, jr;    // fluid bit and j of the cells x0-1 and x0+64 (periodic in x)
    unsigned pad;                   // cells of padding behind this segment's run (last row of a tile only)
    unsigned last;                  // bit of the segment's last lattice cell: 63, or width - 1 where nx is not a multiple of 64 (seg_x0)
};

// Row segments of a lattice whose nx is not a multiple of 64 ("ragged"): nseg = ceil(nx / 64) segments of nx / nseg cells, rounded
// (segment s starts at x = s nx / nseg: widths differ by one cell at most, every one holds >= 32 cells, or all nx < 64 of them); the lanes
// behind a segment's last cell are non-fluid bits of its mask.  With nx a multiple of 64 this is x0 = 64 s, width 64.
__device__ __host__ __forceinline__ int seg_x0(int sg, int nx, int nseg) { return (int)(((long long)sg * nx) / nseg); }

// lattice x of lane `lane` of row segment sg, or -1 for a lane behind the segment's last cell
#define LBMPM_SEG_LANE_X(p, sg, lane) seg_lane_x((p).nx, (p).nseg, sg, lane)
__device__ __forceinline__ int seg_lane_x(int nx, int nseg, int sg, int lane)
{
    const int x0 = seg_x0(sg, nx, nseg), w = seg_x0(sg + 1, nx, nseg) - x0;
    return lane < w ? x0 + lane : -1;
}

// segment records {mask lo, mask hi, first, lbit | rbit << 1 | pad << 2 | (64 - width) << 8} and {jl, jr, -, -}
// RAGGED = false: the caller knows every segment to be 64 cells wide (the words then hold no width field: compile-time 63)
template <bool UNI, bool RAGGED = false>
__device__ __forceinline__ RowTab make_row(u32x4 sg, u32x4 nb)
{
    if (UNI) {
        sg.x = __builtin_amdgcn_readfirstlane(sg.x); sg.y = __builtin_amdgcn_readfirstlane(sg.y);
        sg.z = __builtin_amdgcn_readfirstlane(sg.z); sg.w = __builtin_amdgcn_readfirstlane(sg.w);
        nb.x = __builtin_amdgcn_readfirstlane(nb.x); nb.y = __builtin_amdgcn_readfirstlane(nb.y);
    }
    RowTab t;
    t.m = ((unsigned long long)sg.y << 32) | sg.x;
    t.first = sg.z; t.lbit = sg.w & 1u; t.rbit = (sg.w >> 1) & 1u;
    t.pad = RAGGED ? (sg.w >> 2) & 63u : sg.w >> 2;
    t.last = RAGGED ? 63u - ((sg.w >> 8) & 63u) : 63u;
    t.jl = nb.x; t.jr = nb.y;
    return t;
}

// rows straight from the tables in global memory (set-up, boundary-plane and diagnostics kernels)
struct GlobalRows {
    const u32x4 *seg, *seg2;     // (copies of the fields, not a reference to the kernel argument: a reference makes hipcc keep a private
    int ny, nseg;                //  copy of all of RK3Dev in scratch memory)
    int sg, y;                   // the node's row segment and row
    __device__ __forceinline__ GlobalRows(const RK3Dev &p, int sg_, int y_) : seg(p.seg), seg2(p.seg2), ny(p.ny), nseg(p.nseg), sg(sg_), y(y_) {}
    __device__ __forceinline__ RowTab operator()(int zl, int ry) const
    {
        const size_t r = ((size_t)zl * ny + wrapi(y + ry, ny)) * nseg + sg;
        return make_row<false, true>(seg[r], seg2[r]);
    }
};

// number of set bits of M below bit b (UNI: b is the lane id, M wave-uniform)
template <bool UNI>
__device__ __forceinline__ unsigned bits_below(unsigned long long M, unsigned b)
{
    if (UNI) return __builtin_amdgcn_mbcnt_hi((unsigned)(M >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)M, 0u));
    return (unsigned)__popcll(M & ((1ull << b) - 1ull));
}
template <bool UNI>
__device__ __forceinline__ bool bit_of(unsigned long long M, unsigned b)
{
    if (UNI) return __builtin_amdgcn_inverse_ballot_w64(M);
    return (M >> b) & 1ull;
}

// j and fluid bit of the cell dx to the right of bit b of the segment (periodic in x)
template <bool UNI>
__device__ __forceinline__ void row_cell(const RowTab &t, int dx, unsigned b, unsigned &j, bool &fl)
{
    if (dx == 0) {
        fl = bit_of<UNI>(t.m, b);
        j = t.first + bits_below<UNI>(t.m, b);
    } else if (dx < 0) {
        const unsigned long long M = (t.m << 1) | (unsigned long long)t.lbit;
        fl = bit_of<UNI>(M, b);
        j = b == 0u ? t.jl : t.first - t.lbit + bits_below<UNI>(M, b);
    } else {
        const unsigned long long M = (t.m >> 1) | ((unsigned long long)t.rbit << t.last);
        fl = bit_of<UNI>(M, b);
        j = b == t.last ? t.jr : t.first + bits_below<UNI>(t.m, b) + (bit_of<UNI>(t.m, b) ? 1u : 0u);
    }
}

struct PlaneAddr {               // of the three planes around the plane that is pulled
    const char *base;            // population 0 of the plane below
    unsigned off[3], cnt[3];     // byte offset of the three planes' blocks from base, fluid cells per plane
};

__device__ __forceinline__ PlaneAddr plane_addr(const RK3Dev &p, const double *f, int zl)
{
    PlaneAddr a;
    const unsigned long long p0 = p.pstart[zl - 1], p1 = p.pstart[zl], p2 = p.pstart[zl + 1], p3 = p.pstart[zl + 2];
    a.base = reinterpret_cast<const char *>(f) + (size_t)p0 * (Q * 16);
    a.cnt[0] = (unsigned)(p1 - p0); a.cnt[1] = (unsigned)(p2 - p1); a.cnt[2] = (unsigned)(p3 - p2);
    a.off[0] = 0u; a.off[1] = a.cnt[0] * (unsigned)(Q * 16); a.off[2] = a.off[1] + a.cnt[1] * (unsigned)(Q * 16);
    return a;
}
__device__ __forceinline__ double2 ldg2(const char *uniform_base, unsigned off) { return *reinterpret_cast<const double2 *>(uniform_base + off); }

// pull of the node at bit b (cell x) of the row that `rows(zl, 0)` describes, which must be fluid
// there; also returns the node's own j.  Bounce-back as in pull3: a non-fluid upstream cell
// redirects the load to the opposite population of the node itself.
template <bool FIRST, bool UNI, typename Rows>
__device__ __forceinline__ void pull3c(const RK3Dev &p, const Rows &rows, int x, int zl, unsigned b, double fR[Q], double fB[Q], unsigned &own_j)
{
    constexpr int OPP[Q] = LBMPM_D3Q19_OPP;
    const PlaneAddr a = plane_addr(p, p.fin, zl);
    {
        const RowTab t = rows(zl, 0);
        own_j = t.first + bits_below<UNI>(t.m, b);
    }
    const unsigned own16 = own_j * 16u;
    if (FIRST) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const double2 v = ldg2(a.base, a.off[1] + (unsigned)i * a.cnt[1] * 16u + own16);
            fR[i] = v.x;
            fB[i] = v.y;
        }
        return;
    }
#pragma unroll
    for (int rz = -1; rz <= 1; ++rz)
#pragma unroll
        for (int ry = -1; ry <= 1; ++ry) {
            const RowTab t = rows(zl + rz, ry);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int i = DIRT[(1 - rz) * 9 + (1 - ry) * 3 + (1 - dx)];      // the direction that arrives from (dx, ry, rz)
                if (i < 0) continue;
                unsigned off;
                if (i == 0) off = a.off[1] + own16;
                else {
                    unsigned j;
                    bool fl;
                    row_cell<UNI>(t, dx, b, j, fl);
                    const unsigned up = a.off[1 + rz] + (unsigned)i * a.cnt[1 + rz] * 16u + j * 16u;
                    const unsigned back = a.off[1] + (unsigned)OPP[i] * a.cnt[1] * 16u + own16;
                    off = fl ? up : back;
                }
                const double2 v = ldg2(a.base, off);
                fR[i] = v.x;
                fB[i] = v.y;
            }
        }
}

constexpr int BX3 = 64, BY3 = 4;

// K1: phase field of the streamed, boundary-corrected lattice on the planes zl0 .. zl0+gridDim.z-1
// (all owned planes for the split variant and for diagnostics; only the planes next to a
// neighbour rank for the fused variant)
__global__ __launch_bounds__(BX3 *BY3) void rk3d_phase_field(RK3Dev p, int zl0)
{
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + zl0;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const Cell c = make_cell(p, x, y);
    double fR[Q], fB[Q], rR, rB;
    node_state3(p, c, zl, fR, fB, rR, rB);
    p.phi[idx] = (rR - rB) / (rR + rB);
    if (p.diag) {
        constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
        double mx = 0., my = 0., mz = 0.;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const double t = fR[i] + fB[i];
            mx += (double)CX[i] * t; my += (double)CY[i] * t; mz += (double)CZ[i] * t;
        }
        const double rho = rR + rB;
        p.diag[idx] = rR; p.diag[p.vol + idx] = rB;
        p.diag[2 * p.vol + idx] = mx / rho; p.diag[3 * p.vol + idx] = my / rho; p.diag[4 * p.vol + idx] = mz / rho;
    }
}

// rows of the D3Q19 moment basis (d'Humieres et al. 2002) that relax at their own rates, as functions
// of the direction; all fold to constants in the unrolled loops
__device__ __forceinline__ constexpr double mrt_c2(int i) { return i == 0 ? 0. : (i < 7 ? 1. : 2.); }
__device__ __forceinline__ constexpr double mrt_e(int i) { return 19. * mrt_c2(i) - 30.; }
__device__ __forceinline__ constexpr double mrt_eps(int i) { return (21. * mrt_c2(i) * mrt_c2(i) - 53. * mrt_c2(i) + 24.) / 2.; }
__device__ __forceinline__ constexpr double mrt_pixx(int i)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX;
    return (3. * mrt_c2(i) - 5.) * (3. * (double)(CX[i] * CX[i]) - mrt_c2(i));
}
__device__ __forceinline__ constexpr double mrt_piww(int i)
{
    constexpr int CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    return (3. * mrt_c2(i) - 5.) * (double)(CY[i] * CY[i] - CZ[i] * CZ[i]);
}
// 1 / x and 1 / sqrt(x) for the collision: the hardware estimate + two Newton steps (5 / 8 instructions; the IEEE sequences hipcc
// emits for `1. / x` and `sqrt(x)` take 11 and ~15 with their scaling and fix-ups).  Arguments here are densities and squared
// gradients of O(1e-36 .. 1e2): no denormals, no infinities.  The result is within an ulp or two of the correctly rounded one; the
// oracle keeps the reference's divisions, the tests bound the difference (1e-10).
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.), r, r);
    return r;
}
__device__ __forceinline__ double rsq_nr(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = __builtin_fma(__builtin_fma(-0.5 * x * r, r, 0.5), r, r);
    r = __builtin_fma(__builtin_fma(-0.5 * x * r, r, 0.5), r, r);
    return r;
}

// pair classes of the D3Q19 directions in this file's numbering (i odd, i + 1 opposite): 1 x, 3 y, 5 z, 7 9 xy, 11 13 xz, 15 17 yz.
// Rows of the moment basis of d'Humieres et al. 2002 that relax at their own rates, on the pair sums:
//   e: rest -30, axis -11, diagonal 8;  eps: 12, -4, 1;  pi_xx: x -4, y z 2, xy xz 1, yz -2;  pi_ww: x 0, y -2, z 2, xy 1, xz -1, yz 0
// (tests/test_lattice.py checks these against the closed forms 19 c^2 - 30 etc.)
__device__ __forceinline__ constexpr int pair_class(int i) { return i < 7 ? (i - 1) / 2 : (i < 11 ? 3 : (i < 15 ? 4 : 5)); }   // 0 x, 1 y, 2 z, 3 xy, 4 xz, 5 yz

// BGK + perturbation + recolouring of one node from the colour-blind populations ft = fR + fB,
// the colour densities and the colour gradient; stores the post-collision populations of plane zl.
// Lanes of non-fluid cells whose 128-byte line holds fluid store zeros: partially written
// lines cost the memory system a read-modify-write (measured: +35 % kernel time at porosity 0.65).
// STORE: 0 dense (two 8-byte stores per direction), 1 compact {red, blue} pairs (16 bytes per node and direction), 2 the
// colour-blind population alone + one record {k_R, A} per node (rk3dq.h; every calling lane is a fluid cell there: the lanes that
// write line padding go through pad_store_q); MRT: [RelaxationType] Type
// e_i . v for a lattice direction whose components are 0 or +-1 (they fold in the unrolled loops): the non-zero terms only
__device__ __forceinline__ double edot(int cx, int cy, int cz, double x, double y, double z)
{
    double r = 0.;
    bool any = false;
    if (cx != 0) { r = cx > 0 ? x : -x; any = true; }
    if (cy != 0) { r = any ? (cy > 0 ? r + y : r - y) : (cy > 0 ? y : -y); any = true; }
    if (cz != 0) { r = any ? (cz > 0 ? r + z : r - z) : (cz > 0 ? z : -z); }
    return r;
}

template <int STORE, bool MRT, bool NT = false>
__device__ __forceinline__ void collide_store(const RK3Dev &p, char *red, unsigned stride, unsigned own, bool fluid_in,
                                              const double ft_in[Q], double rR, double rB, double gx, double gy, double gz,
                                              uint32_t *rowflag = nullptr)
{
#pragma clang fp contract(fast)      // fused multiply-adds here (the 2-D kernels stay uncontracted for bit parity with the reference)
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const bool fluid = STORE == 2 ? true : fluid_in;
    // momentum from the differences of opposite pairs (i odd, i + 1 opposite): 9 subtractions + 12 additions, no products by the
    // zero components of e_i (hipcc may not drop `0. * t`: no fast-math here)
    double mx = 0., my = 0., mz = 0.;
    // momentum: the populations that move along +e and along -e summed separately, no products by the zero components of e_i
    // (hipcc may not drop `0. * t`: no fast-math here).  (Not via the differences of opposite pairs: hipcc then keeps all nine alive
    // for the odd part of the MRT relaxation further down, and the kernel spills.)
    {
        double px = 0., py = 0., pz = 0., nx = 0., ny = 0., nz = 0.;
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            const double t = ft_in[i];
            if (CX[i] > 0) px += t;
            if (CX[i] < 0) nx += t;
            if (CY[i] > 0) py += t;
            if (CY[i] < 0) ny += t;
            if (CZ[i] > 0) pz += t;
            if (CZ[i] < 0) nz += t;
        }
        mx = px - nx; my = py - ny; mz = pz - nz;
    }
    // Arithmetic organised for the hardware, not after the reference's statement order (the oracle
    // keeps that; the tests bound the rounding difference): reciprocals by Newton steps, and the
    // directions in opposite pairs, whose equilibrium, perturbation and recolouring terms differ
    // in sign only.
    const double rho = rR + rB, irho = rcp_nr(rho);
    const double ux = mx * irho, uy = my * irho, uz = mz * irho, usq = ux * ux + uy * uy + uz * uz;
    const double phi = (rR - rB) * irho;
    const double omega = rcp_nr(0.5 + rcp_nr((1. + phi) * p.cR + (1. - phi) * p.cB));
    const double g2 = gx * gx + gy * gy + gz * gz;
    const double ign = g2 != 0. ? rsq_nr(g2) : 0.;          // 1 / |G|
    const double gn = g2 * ign, ig2 = ign * ign;
    // STORE 2 keeps k_R alone (k_B = 1 - k_R).  The rule that makes "single colour" a crisp property of a cell (the row flags of
    // rk3dq.h rest on it): a colour whose density is within 2 ulp of the total's rounding (|rho_c| <= 2^-51 rho: absent, or the
    // residue of rho - rho_R at the far end of the other colour's tail) is absent -- k_R exactly 1 / 0 and no recolouring vector;
    // a colour that is present keeps k_R off the end point even where the product with the Newton reciprocal rounds onto it.
    // (Negative densities beyond rounding -- the scheme's undershoot next to a sharp front, 1e-8 -- stay what they are.)
    double kR = rR * irho, arc = rR * rB * irho * irho;
    if (STORE == 2) {
        constexpr double below_one = 1. - 0x1p-53, above_one = 1. + 0x1p-52;
        const double tiny = 0x1p-51 * rho;
        const bool noB = fabs(rB) <= tiny, noR = fabs(rR) <= tiny;
        if (kR == 1.) kR = rB > 0. ? below_one : above_one;
        if (kR == 0.) kR = rR > 0. ? 0x1p-1022 : -0x1p-1022;
        kR = noB ? 1. : (noR ? 0. : kR);
        arc = (noB || noR) ? 0. : arc;
    }
    const double kB = rB * irho, akgn = p.ak * gn;
    const double arcA = (arc * p.rcA) * ign, arcD = (arc * p.rcD) * ign;
    const double c0 = 1. - 1.5 * usq;
    // MRT ([RelaxationType] Type = 'MRT'): f -= M^-1 S M (f - feq) in the D3Q19 basis of d'Humieres et al. 2002
    // with s_e = 1.19, s_eps = s_pi = 1.4, the third-order moments q and m at 1.2 and the stress
    // moments at omega.  (d'Humieres' s_m = 1.98 is unstable next to the Zou-He planes; see the oracle.)
    // d = f - feq has no mass and no momentum, so
    //   even part: Delta = omega d + sum_k (s_k - omega) M_k (M_k . d) / |M_k|^2 over k = e, eps, pi_xx, pi_ww
    //              (the moments see only the sums of opposite pairs),
    //   odd part:  everything that is not momentum relaxes at 1.2, i.e. the pair differences do.
    // The four projections by pair class (the rows are constant on a class): sums over the axis and the diagonal pairs for e and
    // eps, signed sums for pi_xx and pi_ww; back: six distinct corrections, one per class.
    double evc[6] = {0., 0., 0., 0., 0., 0.}, ev0 = 0.;
    const double hodd = MRT ? 0.5 * (1.2 - omega) : 0.;
    if (MRT) {
        // private copies of u: the equilibrium terms of this pass must not be kept (18 registers) for the next
        double vx = ux, vy = uy, vz = uz, v0 = c0;
        asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(v0));
        const double d0 = ft_in[0] - (rho * wq(0)) * v0;
        double sc[6] = {0., 0., 0., 0., 0., 0.};        // sum of the pair sums of f - feq, per class
#pragma unroll
        for (int i = 1; i < Q; i += 2) {
            const double eu = edot(CX[i], CY[i], CZ[i], vx, vy, vz);
            const double sp = (ft_in[i] + ft_in[i + 1]) - 2. * ((rho * wq(i)) * (v0 + 4.5 * eu * eu));
            constexpr int dummy = 0; (void)dummy;
            sc[pair_class(i)] += sp;
        }
        const double Sa = (sc[0] + sc[1]) + sc[2], Sd = (sc[3] + sc[4]) + sc[5];
        double k0 = (-30. * d0 - 11. * Sa) + 8. * Sd, k1 = (12. * d0 - 4. * Sa) + Sd;
        double k2 = ((-4. * sc[0] + 2. * (sc[1] + sc[2])) + (sc[3] + sc[4])) - 2. * sc[5];
        double k3 = (2. * (sc[2] - sc[1]) + sc[3]) - sc[4];
        k0 *= (1.19 - omega) * (1. / 2394.); k1 *= (1.4 - omega) * (1. / 252.);
        k2 *= (1.4 - omega) * (1. / 72.);    k3 *= (1.4 - omega) * (1. / 24.);
        const double ea = -11. * k0 - 4. * k1, ed = 8. * k0 + k1;
        ev0 = -30. * k0 + 12. * k1;
        evc[0] = ea - 4. * k2;
        evc[1] = (ea + 2. * k2) - 2. * k3;
        evc[2] = (ea + 2. * k2) + 2. * k3;
        evc[3] = (ed + k2) + k3;
        evc[4] = (ed + k2) - k3;
        evc[5] = ed - 2. * k2;
    }
    char *blue = red + (size_t)Q * stride;       // red = population 0 of the node's plane, stride = bytes between populations
    auto put = [&](int i, double g, double a) {
        if (STORE == 2) {
            // NT (rk3dq_fused, rows where both colours meet): streaming stores, so that the written lines do not push the rim cells'
            // and halo records' lines -- which neighbouring workgroups read within a march step or two -- out of the XCD's L2
            if (NT) __builtin_nontemporal_store(g, reinterpret_cast<double *>(red + (size_t)i * stride + own));
            else stg(red + (size_t)i * stride, own, g);
        }
        else if (STORE == 1) {
            double2 v;
            v.x = fluid ? kR * g + a : 0.; v.y = fluid ? kB * g - a : 0.;
            *reinterpret_cast<double2 *>(red + (size_t)i * stride + own) = v;
        } else {
            stg(red + (size_t)i * stride, own, fluid ? kR * g + a : 0.);
            stg(blue + (size_t)i * stride, own, fluid ? kB * g - a : 0.);
        }
    };
    // rk3dq_fused: the pulls of the plane after next, issued before the barrier, have had the arithmetic above to arrive; from
    // here on stores are outstanding, and a later wait for the pulls would wait for the stores too (vmcnt retires in order)
    if (STORE == 2) __builtin_amdgcn_s_waitcnt(0x0F70);
    // relaxation as f - (f - feq) omega: a node at equilibrium stays there bit for bit
    put(0, ((ft_in[0] - (ft_in[0] - (rho * wq(0)) * c0) * omega) - (MRT ? ev0 : 0.)) - akgn * bq(0), 0.);
#pragma unroll
    for (int i = 1; i < Q; i += 2) {             // i and i + 1 are opposite
        const double w = wq(i);
        const double eu = edot(CX[i], CY[i], CZ[i], ux, uy, uz);
        const double eg = edot(CX[i], CY[i], CZ[i], gx, gy, gz);
        const double sym = (rho * w) * (c0 + 4.5 * eu * eu), odd = (3. * rho * w) * eu;
        const double pert = (akgn * w * ig2) * (eg * eg) - akgn * bq(i);
        const double a = (i < 7 ? arcA : arcD) * eg;
        double ev = 0., od = 0.;
        if (MRT) {
            ev = evc[pair_class(i)];
            od = hodd * ((ft_in[i] - ft_in[i + 1]) - 2. * odd);
        }
        put(i, ((ft_in[i] - (ft_in[i] - (sym + odd)) * omega) - (ev + od)) + pert, a);
        put(i + 1, ((ft_in[i + 1] - (ft_in[i + 1] - (sym - odd)) * omega) - (ev - od)) + pert, -a);
        __builtin_amdgcn_sched_barrier(0);      // one pair's temporaries at a time: registers are the scarce resource here
    }
    if (STORE == 2) {                            // the record the next step's pulls rebuild the colours from: f_R,i = k_R g_i + c_i e_i.A
        const double ai = arc * ign;
        double2 v, w;
        v.x = kR; v.y = ai * gx; w.x = ai * gy; w.y = ai * gz;
        // row flag (rk3dq.h): every fluid cell of this wave's row segment pure red / pure blue -> the records are not written
        const bool noA = v.y == 0. && w.x == 0. && w.y == 0.;
        const bool notred = !(v.x == 1. && noA), notblue = !(v.x == 0. && noA);
        const unsigned code = (__ballot(notred) == 0ull ? 1u : 0u) | (__ballot(notblue) == 0ull ? 2u : 0u);
        *rowflag = code;
        if (code == 0u) {
            char *s = red + (size_t)Q * stride + (size_t)own * 4u;
            if (NT) {
                double *sd = reinterpret_cast<double *>(s);
                __builtin_nontemporal_store(v.x, sd); __builtin_nontemporal_store(v.y, sd + 1);
                __builtin_nontemporal_store(w.x, sd + 2); __builtin_nontemporal_store(w.y, sd + 3);
            } else {
                *reinterpret_cast<double2 *>(s) = v;
                *reinterpret_cast<double2 *>(s + 16) = w;
            }
        }
    }
}

// true when the 128-byte line (16 lanes) this lane stores into holds at least one fluid node
__device__ __forceinline__ bool line_has_fluid(bool fluid, unsigned lane, int fill)
{
    if (!fill) return fluid;
    const unsigned long long m = __ballot(fluid);
    const unsigned g = (unsigned)fill;
    return ((m >> (lane & (64u - g))) & ((g == 64u) ? ~0ull : ((1ull << g) - 1ull))) != 0;
}

// K2 of the split variant: stream + boundaries again, colour gradient from the global phase field,
// collision, store
template <bool MRT>
__global__ __launch_bounds__(BX3 *BY3) void rk3d_collide(RK3Dev p)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    const bool fluid = p.flags[idx] & 1;
    if (!line_has_fluid(fluid, threadIdx.x, p.fill)) return;
    const Cell c = make_cell(p, x, y);
    double fR[Q], fB[Q], rR, rB;
    node_state3(p, c, zl, fR, fB, rR, rB);
    const char *ph0 = reinterpret_cast<const char *>(p.phi + (size_t)(zl - 1) * p.plane2);
    double gx = 0., gy = 0., gz = 0.;
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const double ph = ldg(ph0, (unsigned)(1 + CZ[i]) * p.plane_bytes + c.o[1 + CY[i]][1 + CX[i]]);   // non-fluid cells hold solidPhi
        gx += 3. * wq(i) * (double)CX[i] * ph;
        gy += 3. * wq(i) * (double)CY[i] * ph;
        gz += 3. * wq(i) * (double)CZ[i] * ph;
    }
    double ft[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) ft[i] = fR[i] + fB[i];
    collide_store<0, MRT>(p, const_cast<char *>(plane_ptr(p.fout, p, zl)), p.plane_bytes, c.o[1][1], fluid, ft, rR, rB, gx, gy, gz);
}

// Fused time step (default): one block owns a TX x TY column of nodes and marches along z.
// At march step z every thread pulls the populations of its node of plane z+1 ONCE, reduces them to
// what the collision needs (19 colour-blind sums + 2 densities), parks that in LDS and puts the
// node's phase field into a four-plane LDS ring (the ring also covers the one-cell rim around the
// tile, whose phase field the first waves recompute; those pulls mostly hit L2, the cells belong
// to neighbouring tiles of the same XCD band); after one barrier the thread collides its node of
// plane z from the parked state of the previous march step and the ring.  Per node and step the
// HBM sees 38 population reads + 38 writes: the separate phase-field sweep and the global
// phase-field round trip of the split variant are gone.  Registers are kept under 168 so that
// three blocks (12 waves) share a CU and cover each other's load latency.
template <int TX, int TY>
struct March {
    static constexpr int FX = TX + 2, FY = TY + 2, NT = TX * TY, NH = 2 * TX + 2 * FY, RING = 4;
    static_assert(NT % 64 == 0 && (TX == 32 || TX == 64) && NH <= NT, "tile rows must fill 128-byte lines");
};

// lattice coordinate of tile-frame coordinate g (periodic), or -1 when nobody needs it
__device__ __forceinline__ int ring_coord(int g, int n) { return g < -1 || g > n ? -1 : (g == -1 ? n - 1 : (g == n ? 0 : g)); }

// meta word of plane zl: for ghost planes the word of the source plane decides the bounce-back
// addresses, the fluid bit is the same (identical geometry by construction); 0 outside the owned planes
__device__ __forceinline__ unsigned plane_meta(const RK3Dev &p, int zl, unsigned own)
{
    if (zl < 1 || zl > p.nzl) return 0u;
    return load_meta(p, source_plane_x(p, zl), own);
}

// phase field of a cell of plane zl for the ring; leaves the pulled, boundary-corrected populations
// and densities of a fluid node of an owned plane in fR/fB/rR/rB
template <bool FIRST>
__device__ __forceinline__ double ring_phi(const RK3Dev &p, const Cell &c, int zl, unsigned meta, double fR[Q], double fB[Q],
                                           double &rR, double &rB)
{
    if (zl == 0 || zl == p.nzl + 1) return (p.phi + (size_t)zl * p.plane2)[c.o[1][1] >> 3];   // neighbour rank's plane (or outside: solidPhi)
    if (!(meta >> 31)) return p.solidPhi;
    pull3<FIRST>(p, c, source_plane_x(p, zl), meta, fR, fB);
    finish_state3(p, zl, fR, fB, rR, rB);
    return (rR - rB) / (rR + rB);
}

template <int TX, int TY, bool FIRST, bool MRT>
__global__ __launch_bounds__(TX *TY, 768 / (TX * TY)) void rk3d_fused(RK3Dev p, int tilesX, int tilesY, int rows_per_xcd, int chunk_len, int z_first, int z_last)
{
    using M = March<TX, TY>;
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    __shared__ double sphi[M::RING][M::FY][M::FX];
    __shared__ double park[Q][M::NT];
    // workgroup b runs on XCD b % 8: every XCD owns a band of tile rows, so rim re-reads hit its L2
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int tx = slot % tilesX, r = slot / tilesX, ty = xcd * rows_per_xcd + r % rows_per_xcd, chunk = r / rows_per_xcd;
    if (ty >= tilesY) return;
    const int tid = threadIdx.x, lx = tid % TX, ly = tid / TX;
    const int x = tx * TX + lx, y = ty * TY + ly;
    const bool own = x < p.nx && y < p.ny;
    const int xo = ring_coord(x, p.nx), yo = ring_coord(y, p.ny);           // own cell (may be the wrapped rim of a cut tile)
    const bool has_own = xo >= 0 && yo >= 0;
    const Cell co = make_cell(p, has_own ? xo : 0, has_own ? yo : 0);
    const unsigned own_off = co.o[1][1];
    // rim cell of this thread: bottom row, top row, then the two columns (corners included)
    int hlx = 0, hly = 0;
    bool has_rim = false;
    Cell ch = co;
    if (tid < M::NH) {
        if (tid < 2 * TX) { hlx = 1 + tid % TX; hly = tid < TX ? 0 : M::FY - 1; }
        else { const int k = tid - 2 * TX; hlx = k < M::FY ? 0 : M::FX - 1; hly = k % M::FY; }
        const int hx = ring_coord(tx * TX + hlx - 1, p.nx), hy = ring_coord(ty * TY + hly - 1, p.ny);
        has_rim = hx >= 0 && hy >= 0;
        if (has_rim) ch = make_cell(p, hx, hy);
    }
    const int za = z_first + chunk * chunk_len, zb = min(za + chunk_len - 1, z_last);      // owned planes of this block
    unsigned meta_o = has_own ? plane_meta(p, za - 1, own_off) : 0u;        // meta words one plane ahead of their use
    unsigned meta_h = has_rim ? plane_meta(p, za - 1, ch.o[1][1]) : 0u;
    bool fluid = false;
    double rR = 1., rB = 1.;                    // densities of the parked plane
#pragma unroll
    for (int i = 0; i < Q; ++i) park[i][tid] = 1.;

    for (int z = za - 2; z <= zb; ++z) {
        const unsigned mo = meta_o, mh = meta_h;
        meta_o = has_own ? plane_meta(p, z + 2, own_off) : 0u;
        meta_h = has_rim ? plane_meta(p, z + 2, ch.o[1][1]) : 0u;
        // ---- plane z + 1, rim cells: phase field only
        if (has_rim) {
            double fR[Q], fB[Q], a, c;
            sphi[(z + 1) & (M::RING - 1)][hly][hlx] = ring_phi<FIRST>(p, fresh(ch), z + 1, mh, fR, fB, a, c);
        }
        // ---- plane z + 1, own cell: pull once, phase field into the ring, reduced state into the park
        //      (the park still holds plane z: swap)
        double ft[Q];
        const double rRz = rR, rBz = rB;
        {
            double fR[Q], fB[Q], rRn = 1., rBn = 1.;
#pragma unroll
            for (int i = 0; i < Q; ++i) { fR[i] = 0.; fB[i] = 0.; }
            double ph = p.solidPhi;
            if (has_own) {
                ph = ring_phi<FIRST>(p, fresh(co), z + 1, mo, fR, fB, rRn, rBn);
                sphi[(z + 1) & (M::RING - 1)][ly + 1][lx + 1] = ph;
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                ft[i] = park[i][tid];
                park[i][tid] = fR[i] + fB[i];
            }
            rR = rRn; rB = rBn;
        }
        __syncthreads();
        // ---- plane z: collide
        if (z >= za) {
            const bool isfl = fluid && own;
            if (line_has_fluid(isfl, tid & 63, p.fill) && own) {
                double gx = 0., gy = 0., gz = 0.;
#pragma unroll
                for (int i = 1; i < Q; ++i) {
                    const double ph = sphi[(z + CZ[i]) & (M::RING - 1)][ly + 1 + CY[i]][lx + 1 + CX[i]];
                    gx += 3. * wq(i) * (double)CX[i] * ph;
                    gy += 3. * wq(i) * (double)CY[i] * ph;
                    gz += 3. * wq(i) * (double)CZ[i] * ph;
                }
                collide_store<0, MRT>(p, const_cast<char *>(plane_ptr(p.fout, p, z)), p.plane_bytes, own_off, isfl, ft, rRz, rBz, gx, gy, gz);
            }
        }
        fluid = (mo >> 31) && z + 1 >= 1 && z + 1 <= p.nzl;
    }
}


// ---------------------------------------------------------------- compact-storage kernels
// K1 on compact storage (boundary planes of a slab, diagnostics)
__global__ __launch_bounds__(BX3 *BY3) void rk3dc_phase_field(RK3Dev p, int zl0)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + zl0;
    if (y >= p.ny || x >= p.nx) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[Q], fB[Q], rR, rB;
    unsigned j;
    const GlobalRows rows{p, x >> 6, y};
    if (p.first) pull3c<true, false>(p, rows, x, source_plane(p, zl), (unsigned)(x & 63), fR, fB, j);
    else pull3c<false, false>(p, rows, x, source_plane(p, zl), (unsigned)(x & 63), fR, fB, j);
    finish_state3(p, zl, fR, fB, rR, rB);
    p.phi[idx] = (rR - rB) / (rR + rB);
    if (p.diag) {
        constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
        double mx = 0., my = 0., mz = 0.;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const double t = fR[i] + fB[i];
            mx += (double)CX[i] * t; my += (double)CY[i] * t; mz += (double)CZ[i] * t;
        }
        const double rho = rR + rB;
        p.diag[idx] = rR; p.diag[p.vol + idx] = rB;
        p.diag[2 * p.vol + idx] = mx / rho; p.diag[3 * p.vol + idx] = my / rho; p.diag[4 * p.vol + idx] = mz / rho;
    }
}

// the rows around a tile, staged in LDS ahead of their use: [ring slot][tile row - 2 .. TY + 1][segment tx-1, tx, tx+1][record 0, 1]
template <int TY>
struct TileRows {
    static constexpr int ROWS = TY + 4, SLOTS = 8;
    const u32x4 (*ring)[ROWS][6];
    int lrow, k;                 // this lane's row inside the staged rows, its segment slot
    template <bool UNI, bool RAGGED>
    __device__ __forceinline__ RowTab get(int zl, int ry) const
    {
        const u32x4 *r = ring[zl & (SLOTS - 1)][lrow + ry];
        return make_row<UNI, RAGGED>(r[2 * k], r[2 * k + 1]);
    }
};
template <int TY, bool UNI, bool RAGGED = false>
struct TileRowsU {
    TileRows<TY> t;
    __device__ __forceinline__ RowTab operator()(int zl, int ry) const { return t.template get<UNI, RAGGED>(zl, ry); }
};

// phase field of a cell of plane zl for the ring (compact storage); for a fluid node of an owned
// plane also the pulled, boundary-corrected populations, densities and the node's j
template <bool FIRST, bool UNI, typename Rows>
__device__ __forceinline__ double ring_phi_c(const RK3Dev &p, const Rows &rows, int x, int y, int zl, unsigned b, bool &fluid,
                                             double fR[Q], double fB[Q], double &rR, double &rB, unsigned &j)
{
    fluid = false;
    if (zl == 0 || zl == p.nzl + 1) return (p.phi + (size_t)zl * p.plane2)[(size_t)y * p.pitch + x];   // neighbour rank's plane (or outside: solidPhi)
    const RowTab t = rows(zl, 0);
    fluid = bit_of<UNI>(t.m, b);
    if (!fluid) return p.solidPhi;
    j = t.first + bits_below<UNI>(t.m, b);
    unsigned js;
    pull3c<FIRST, UNI>(p, rows, x, source_plane(p, zl), b, fR, fB, js);
    finish_state3(p, zl, fR, fB, rR, rB);
    return (rR - rB) / (rR + rB);
}

// the marching kernel of rk3d_fused on compact storage (TX = 64: a wave owns one row segment)
template <int TY, bool FIRST, bool MRT>
__global__ __launch_bounds__(64 * TY, 512 / (64 * TY)) void rk3dc_fused(RK3Dev p, int tilesX, int tilesY, int rows_per_xcd, int chunk_len, int z_first, int z_last)
{
    constexpr int TX = 64;
    using M = March<TX, TY>;
    using TR = TileRows<TY>;
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    __shared__ double sphi[M::RING][M::FY][M::FX];
    __shared__ double park[Q][M::NT];
    __shared__ u32x4 srow[TR::SLOTS][TR::ROWS][6];
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int tx = slot % tilesX, r = slot / tilesX, ty = xcd * rows_per_xcd + r % rows_per_xcd, chunk = r / rows_per_xcd;
    if (ty >= tilesY) return;
    const int tid = threadIdx.x, lx = tid % TX, ly = tid / TX;
    const int x = tx * TX + lx, y = ty * TY + ly;
    const bool own = y < p.ny;                                     // nx is a multiple of 64 here
    const int yo = ring_coord(y, p.ny);                            // row ny of a cut tile = row 0, rim of row ny-1
    const bool has_own = yo >= 0;
    // rim cell of this thread: bottom row, top row (one wave each), then the two columns (corners included)
    int hlx = 0, hly = 0, hx = 0, hy = 0;
    bool has_rim = false;
    if (tid < M::NH) {
        if (tid < 2 * TX) { hlx = 1 + tid % TX; hly = tid < TX ? 0 : M::FY - 1; }
        else { const int k = tid - 2 * TX; hlx = k < M::FY ? 0 : M::FX - 1; hly = k % M::FY; }
        hx = ring_coord(tx * TX + hlx - 1, p.nx);
        hy = ring_coord(ty * TY + hly - 1, p.ny);
        has_rim = hx >= 0 && hy >= 0;
    }
    const TileRowsU<TY, true> rows_own{{srow, ly + 2, 1}};
    const TileRowsU<TY, true> rows_rimrow{{srow, hly + 1, 1}};
    const TileRowsU<TY, false> rows_rimcol{{srow, hly + 1, hlx == 0 ? 0 : 2}};
    const int za = z_first + chunk * chunk_len, zb = min(za + chunk_len - 1, z_last);
    // Row records are staged two march steps ahead of their first use: fetched into a register
    // during one step, written to LDS at the top of the next (so nobody waits for that fetch),
    // read from the step after.  72 lanes, one 16-byte record each.
    auto fetch_rows = [&](int zl) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (tid < TR::ROWS * 6 && zl >= 0 && zl <= p.nzl + 1) {
            const int row = tid / 6, k = tid % 6;
            int yy = (ty * TY - 2 + row) % p.ny;
            if (yy < 0) yy += p.ny;
            int sg = tx - 1 + (k >> 1);
            sg = sg < 0 ? sg + p.nseg : (sg >= p.nseg ? sg - p.nseg : sg);
            const size_t rr = ((size_t)zl * p.ny + yy) * p.nseg + sg;
            v = (k & 1) ? p.seg2[rr] : p.seg[rr];
        }
        return v;
    };
    auto put_rows = [&](int zl, u32x4 v) {
        if (tid < TR::ROWS * 6) const_cast<u32x4 &>(srow[zl & (TR::SLOTS - 1)][tid / 6][tid % 6]) = v;
    };
    for (int zl = za - 3; zl <= za + 2; ++zl) put_rows(zl, fetch_rows(zl));
    u32x4 staged = fetch_rows(za + 3);
    // Software pipeline of the own cells: the pulls of plane z + 2 are issued before the barrier and
    // the collision of plane z, and consumed at the top of the next march step (the reduced state
    // sits in the LDS park meanwhile, so the pulls in flight are all the registers carry).
    bool fluid = false, fl_raw = false;         // node of the parked plane / of the plane in flight is fluid
    bool padz = false, pad_raw = false;         // idle lane that writes line padding for that plane
    unsigned jz = 0, j_raw = 0;                 // their j
    double rR = 1., rB = 1.;                    // densities of the parked plane
    double rawR[Q], rawB[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) { park[i][tid] = 1.; rawR[i] = 0.; rawB[i] = 0.; }
    __syncthreads();
    auto issue = [&](int zl) {                  // pulls of the own cell of plane zl
        fl_raw = false; pad_raw = false;
        if (!has_own || zl < 1 || zl > p.nzl) return;
        const RowTab t = rows_own(zl, 0);
        fl_raw = bit_of<true>(t.m, (unsigned)lx);
        if (!fl_raw) {
            // the first `pad` idle lanes of the last row of a tile write the zeros that complete the tile's last line
            const unsigned rank = bits_below<true>(~t.m, (unsigned)lx);
            pad_raw = own && rank < t.pad;
            j_raw = t.first + (unsigned)__popcll(t.m) + rank;
            return;
        }
        j_raw = t.first + bits_below<true>(t.m, (unsigned)lx);
        unsigned js;
        pull3c<FIRST, true>(p, rows_own, x, source_plane(p, zl), (unsigned)lx, rawR, rawB, js);
    };
    issue(za - 1);

    for (int z = za - 2; z <= zb; ++z) {
        put_rows(z + 5, staged);                // read from the next march step on
        staged = fetch_rows(z + 6);
        // ---- plane z + 1, rim cells: phase field only
        if (tid < 2 * TX) {
            if (has_rim) {
                double fR[Q], fB[Q], a, c;
                bool f;
                unsigned j;
                sphi[(z + 1) & (M::RING - 1)][hly][hlx] = ring_phi_c<FIRST, true>(p, rows_rimrow, hx, hy, z + 1, (unsigned)(tid % TX), f, fR, fB, a, c, j);
            }
        } else if (has_rim) {
            double fR[Q], fB[Q], a, c;
            bool f;
            unsigned j;
            sphi[(z + 1) & (M::RING - 1)][hly][hlx] = ring_phi_c<FIRST, false>(p, rows_rimcol, hx, hy, z + 1, (unsigned)(hx & 63), f, fR, fB, a, c, j);
        }
        // ---- plane z + 1, own cell (pulled during the previous march step): boundary rules, phase
        //      field into the ring, reduced state into the park (which still holds plane z: swap)
        double ft[Q];
        const double rRz = rR, rBz = rB;
        const unsigned jzz = jz;
        const bool fluidn = fl_raw;
        {
            double rRn = 1., rBn = 1., ph = p.solidPhi;
            if (z + 1 == 0 || z + 1 == p.nzl + 1) { if (has_own) ph = (p.phi + (size_t)(z + 1) * p.plane2)[(size_t)yo * p.pitch + x]; }
            else if (fluidn) {
                finish_state3(p, z + 1, rawR, rawB, rRn, rBn);
                ph = (rRn - rBn) / (rRn + rBn);
            }
            if (has_own) sphi[(z + 1) & (M::RING - 1)][ly + 1][lx + 1] = ph;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                ft[i] = park[i][tid];
                double t = rawR[i] + rawB[i];
                asm volatile("" : "+v"(t));      // here, not later: frees the pull registers for the next plane
                park[i][tid] = t;
            }
            rR = rRn; rB = rBn; jz = j_raw;
        }
        const bool padzz = padz;
        padz = pad_raw;
        // ---- pulls of plane z + 2 into flight
        if (z + 2 <= zb + 1) issue(z + 2);
        else { fl_raw = false; pad_raw = false; }
        __syncthreads();
        // ---- plane z: collide
        if (z >= za && ((fluid && own) || padzz)) {
            double gx = 0., gy = 0., gz = 0.;
#pragma unroll
            for (int i = 1; i < Q; ++i) {
                const double ph = sphi[(z + CZ[i]) & (M::RING - 1)][ly + 1 + CY[i]][lx + 1 + CX[i]];
                gx += 3. * wq(i) * (double)CX[i] * ph;
                gy += 3. * wq(i) * (double)CY[i] * ph;
                gz += 3. * wq(i) * (double)CZ[i] * ph;
            }
            const unsigned long long p0 = p.pstart[z], p1 = p.pstart[z + 1];
            const unsigned cnt = (unsigned)(p1 - p0);
            collide_store<1, MRT>(p, reinterpret_cast<char *>(p.fout) + (size_t)p0 * (Q * 16), cnt * 16u, jzz * 16u, fluid, ft, rRz, rBz, gx, gy, gz);
        }
        fluid = fluidn;
    }
}

// halo packing on compact storage: the five populations (both colours) that cross each cut, as
// contiguous runs of the outermost owned planes
__global__ void rk3dc_pack(RK3Dev p, const double *f, double *send_up, double *send_dn)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long pt = p.pstart[p.nzl], pb = p.pstart[1];
    const size_t ct = (size_t)(p.pstart[p.nzl + 1] - pt), cb = (size_t)(p.pstart[2] - pb);
    const double2 *f2 = reinterpret_cast<const double2 *>(f);
    double2 *up = reinterpret_cast<double2 *>(send_up), *dn = reinterpret_cast<double2 *>(send_dn);
    for (int j = 0; j < 5; ++j) {
        if (k < ct) up[(size_t)j * ct + k] = f2[(size_t)pt * Q + (size_t)UP[j] * ct + k];
        if (k < cb) dn[(size_t)j * cb + k] = f2[(size_t)pb * Q + (size_t)DN[j] * cb + k];
    }
}

__global__ void rk3dc_unpack(RK3Dev p, double *f, const double *recv_from_below, const double *recv_from_above, int have_below,
                             int have_above)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long pl = p.pstart[0], ph = p.pstart[p.nzl + 1];
    const size_t cl = (size_t)(p.pstart[1] - pl), ch = (size_t)(p.pstart[p.nzl + 2] - ph);
    double2 *f2 = reinterpret_cast<double2 *>(f);
    const double2 *lo = reinterpret_cast<const double2 *>(recv_from_below), *hi = reinterpret_cast<const double2 *>(recv_from_above);
    for (int j = 0; j < 5; ++j) {
        if (have_below && k < cl) f2[(size_t)pl * Q + (size_t)UP[j] * cl + k] = lo[(size_t)j * cl + k];
        if (have_above && k < ch) f2[(size_t)ph * Q + (size_t)DN[j] * ch + k] = hi[(size_t)j * ch + k];
    }
}

__global__ void rk3dc_init_rest(RK3Dev p, const double *rho_r, const double *rho_b, double *f)
{
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    if (!(p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1)) return;
    const GlobalRows rows{p, x >> 6, y};
    const RowTab t = rows(zl, 0);
    const unsigned j = t.first + bits_below<false>(t.m, (unsigned)(x & 63));
    const size_t sd = ((size_t)(zl - 1) * p.ny + y) * p.nx + x;
    const double a = rho_r[sd], b = rho_b[sd];
    const unsigned long long p0 = p.pstart[zl];
    const size_t cnt = (size_t)(p.pstart[zl + 1] - p0);
    double2 *pl = reinterpret_cast<double2 *>(f) + (size_t)p0 * Q;
    for (int i = 0; i < Q; ++i) {
        double2 v;
        v.x = wq(i) * a; v.y = wq(i) * b;
        pl[(size_t)i * cnt + j] = v;
    }
}

// halo packing: the five populations per colour that cross each cut
__global__ void rk3d_pack(RK3Dev p, const double *f, double *send_up, double *send_dn)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.plane2) return;
    const double *top = f + (size_t)p.nzl * 2 * Q * p.plane2, *bot = f + (size_t)2 * Q * p.plane2;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 5; ++j) {
            send_up[(size_t)(c * 5 + j) * p.plane2 + k] = top[(size_t)(c * Q + UP[j]) * p.plane2 + k];
            send_dn[(size_t)(c * 5 + j) * p.plane2 + k] = bot[(size_t)(c * Q + DN[j]) * p.plane2 + k];
        }
}

__global__ void rk3d_unpack(RK3Dev p, double *f, const double *recv_from_below, const double *recv_from_above,
                            int have_below, int have_above)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.plane2) return;
    double *lo = f, *hi = f + (size_t)(p.nzl + 1) * 2 * Q * p.plane2;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 5; ++j) {
            if (have_below) lo[(size_t)(c * Q + UP[j]) * p.plane2 + k] = recv_from_below[(size_t)(c * 5 + j) * p.plane2 + k];
            if (have_above) hi[(size_t)(c * Q + DN[j]) * p.plane2 + k] = recv_from_above[(size_t)(c * 5 + j) * p.plane2 + k];
        }
}

// f = w rho at rest on the owned planes (3-D analogue of RKD2Q9.py:577-601); rho arrays are dense
// [nzl][ny][nx] on the device
__global__ void rk3d_init_rest(RK3Dev p, const double *rho_r, const double *rho_b, double *f)
{
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    const size_t cell = (size_t)y * p.pitch + x;
    const bool fluid = p.flags[(size_t)zl * p.plane2 + cell] & 1;
    const size_t s = ((size_t)(zl - 1) * p.ny + y) * p.nx + x;
    const double a = fluid ? rho_r[s] : 0., b = fluid ? rho_b[s] : 0.;
    double *pl = f + (size_t)zl * 2 * Q * p.plane2;
    for (int i = 0; i < Q; ++i) {
        pl[(size_t)i * p.plane2 + cell] = wq(i) * a;
        pl[(size_t)(Q + i) * p.plane2 + cell] = wq(i) * b;
    }
}

// phi of every non-fluid cell (halo planes outside the lattice included) = the wetting value
__global__ void rk3d_init_phi(RK3Dev p, double *phi)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < p.vol && !(p.flags[k] & 1)) phi[k] = p.solidPhi;
}

__global__ void rk3d_setup_solidnbr(RK3Dev p, uint32_t *solidnbr)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z;
    if (x >= p.nx || y >= p.ny) return;
    unsigned b = 0;
    for (int i = 1; i < Q; ++i) {
        const int zn = zl + CZ[i];
        bool fluid = false;
        if (zn >= 0 && zn <= p.nzl + 1)
            fluid = p.flags[(size_t)zn * p.plane2 + (size_t)wrapi(y + CY[i], p.ny) * p.pitch + wrapi(x + CX[i], p.nx)] & 1;
        if (!fluid) b |= 1u << (i - 1);
    }
    if (p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1) b |= 1u << 31;
    solidnbr[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] = b;
}

#include "rk3dq.h"
#include "rk3d_state.h"

}  // namespace

// ====================================================================== host side
#include "rk3d_transport.h"

struct lbmpm_rk3d {
    lbmpm_rk3d_config cfg;
    int nx, ny, nzl, pitch;
    size_t plane2, vol;
    int64_t nfluid = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t *flags = nullptr;
    uint32_t *solidnbr = nullptr;
    double *fA = nullptr, *fB = nullptr, *phi = nullptr, *diag = nullptr;
    uint32_t *purA = nullptr, *purB = nullptr;       // row flags of the q23 storage, swapped with fA / fB
    double *send_up = nullptr, *send_dn = nullptr, *recv_below = nullptr, *recv_above = nullptr;
    std::vector<uint8_t> h_domain;   // owned planes only, [nzl][ny][nx]
    // compact storage (fluid cells only): default whenever nx is a multiple of 64; LBMPM_RK3D_LAYOUT=dense overrides
    // [BoundaryCondition] BoundaryTypeOutlet = 'Convective'; conv_lo: on the slab that holds the lattice's bottom, the local index of the last
    // of the three planes that copy plane 3 (3), else 0
    bool conv = false;
    int conv_lo = 0;
    bool compact = false;
    // q23: compact storage of 19 colour-blind populations + {k_R, A} per cell instead of 2 x 19 (rk3dq.h); default on compact
    // storage, LBMPM_RK3D_STORAGE=38 keeps the 38-value kernels (the cross-check)
    bool q23 = false;
    unsigned long long *trace = nullptr;      // dev tool, see RK3Dev
    unsigned *slotq = nullptr;                // tile counters of the rk3dq_fused launches (launch_q23); null: tiles by block index
    unsigned slot_launches = 0;
    unsigned long long *probe_bad = nullptr;    // mismatch counter of lbmpm_rk3d_transport_probe
    slabtx::Transport tx;            // the exchange's transport when the library drives it itself (lbmpm_rk3d_ipc_* / lbmpm_rk3d_rccl_connect)
    bool halo_valid = false;         // q23 slabs: the halo planes (populations, records, flags, phase field) belong to the current state
    int dbg = 0;
    int nseg = 0;
    size_t ncells = 0;               // stored cells, halo planes included
    unsigned long long *pstart = nullptr;
    uint32_t *seg = nullptr, *seg2 = nullptr;        // 4 words per record
    std::vector<unsigned long long> h_pstart;
    std::vector<uint8_t> h_rowpop;   // fluid cells per row segment [rows][nseg] (lbmpm_rk3d_storage_info)
    bool streamed = false;
    int variant = 0, tile = 0, chunk_len = 32, fill = 16;   // tuning: LBMPM_RK3D_VARIANT / _TILE / _CHUNK / _FILL
    bool chunk_auto = false;
    int ncu = 256;
    // planes next to each face that wait for the halo exchange (LBMPM_RK3D_BOUNDARY): plane 1 needs the neighbour's populations
    // and phase field, plane 2 the phase field of plane 1; from plane 3 on nothing of the neighbour is read.  Measured with k = 8
    // virtual ranks on one GPU (tools/slabbench.py, 512^3): depth 2 / 3 / 4 / 8 -> +4.5 / +5.9 / +6.6 / +8.1 % over the single slab
    int boundary = 2;
    hipStream_t aux = nullptr;       // second stream for the interior planes (lbmpm_rk3d_collide_interior)
    hipEvent_t ev_dep = nullptr, ev_done = nullptr;
    bool interior_pending = false;
    int64_t steps = 0, bytes = 0;
    lbmpm::EventPool slab_pool;      // lbmpm_rk3d_step_slab(timed): 4 event pairs per step {step, interior, exchange chain, boundary}
    int64_t slab_timed_steps = 0;
    // steady-state watchdog (lbmpm_rk3d_sync_deadline): a word in pinned host memory that the exchange chain of every slab step writes
    // its step number into -- the host's evidence of progress -- and a private stream for the watchdog's own copies
    unsigned long long *beat_host = nullptr, *beat_dev = nullptr;
    hipStream_t wd_stream = nullptr;
    int64_t observed_at = -1;        // value of `steps` when lbmpm_rk3d_phase_field(ctx, 1) last filled phi / diag for all owned planes
    lbmpm::EventPool pool;
};

namespace {

RK3Dev make_dev(const lbmpm_rk3d *c)
{
    RK3Dev p{};
    p.nx = c->nx; p.ny = c->ny; p.nzl = c->nzl; p.pitch = c->pitch; p.plane2 = c->plane2; p.vol = c->vol;
    p.plane_bytes = (unsigned)(c->plane2 * sizeof(double));
    p.seg = reinterpret_cast<const u32x4 *>(c->seg); p.seg2 = reinterpret_cast<const u32x4 *>(c->seg2); p.pstart = c->pstart; p.nseg = c->nseg;
    p.z0 = (int)c->cfg.z_offset; p.nzg = (int)c->cfg.nz_global;
    p.flags = c->flags; p.solidnbr = c->solidnbr; p.fin = c->fA; p.fout = c->fB; p.phi = c->phi; p.diag = nullptr;
    p.ak = (c->cfg.ak_r + c->cfg.ak_b) * 0.5; p.beta = c->cfg.beta; p.cR = 1. / (2. * (c->cfg.tau_r - 0.5)); p.cB = 1. / (2. * (c->cfg.tau_b - 0.5));
    p.solidPhi = c->cfg.solid_phi; p.vzR = c->cfg.inlet_vz_r; p.vzB = c->cfg.inlet_vz_b;
    p.rhoOutR = c->cfg.outlet_rho_r; p.rhoOutB = c->cfg.outlet_rho_b;
    p.conv = c->conv ? 1 : 0;
    p.halo_lo = 0;
    p.inletP = c->cfg.inlet_type == LBMPM_INLET_PRESSURE ? 1 : 0;
    if (p.inletP) { p.vzR = c->cfg.inlet_rho_r; p.vzB = c->cfg.inlet_rho_b; }
    p.rcA = c->cfg.beta * (c->cfg.recolor_axis > 0. ? c->cfg.recolor_axis : 1. / 18.);
    p.rcD = c->cfg.beta * (c->cfg.recolor_diag > 0. ? c->cfg.recolor_diag : (1. / 36.) * 0.70710678118654752440);
    p.first = c->streamed ? 0 : 1;
    p.fill = c->fill;
    p.mrt = c->cfg.relaxation;
#ifdef LBMPM_DEV
    p.trace = c->trace;
#endif
    p.pur_in = c->purA; p.pur_out = c->purB;
    return p;
}

template <typename T>
int dev_alloc(lbmpm_rk3d *c, T **ptr, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, count * sizeof(T));
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); return LBMPM_ERR_NOMEM; }
    e = hipMemsetAsync(v, 0, count * sizeof(T), c->stream);
    if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return LBMPM_ERR_HIP; }
    *ptr = static_cast<T *>(v);
    c->bytes += (int64_t)(count * sizeof(T));
    return LBMPM_OK;
}

dim3 grid3(const lbmpm_rk3d *c, int planes) { return dim3((c->nx + BX3 - 1) / BX3, (c->ny + BY3 - 1) / BY3, planes); }

}  // namespace

extern "C" int lbmpm_rk3d_create(const lbmpm_rk3d_config *cfg, const uint8_t *is_domain_with_halo, lbmpm_rk3d **out)
{
    LBMPM_REQUIRE(cfg && is_domain_with_halo && out, "lbmpm_rk3d_create: null argument");
    LBMPM_REQUIRE(cfg->nx >= 4 && cfg->ny >= 4 && cfg->nz_local >= 2 && cfg->nz_global >= 8,
                  "lbmpm_rk3d_create: domain %lld x %lld x %lld(local) out of range", (long long)cfg->nx,
                  (long long)cfg->ny, (long long)cfg->nz_local);
    LBMPM_REQUIRE(cfg->z_offset >= 0 && cfg->z_offset + cfg->nz_local <= cfg->nz_global, "slab [%lld, %lld) outside 0..%lld",
                  (long long)cfg->z_offset, (long long)(cfg->z_offset + cfg->nz_local), (long long)cfg->nz_global);
    LBMPM_REQUIRE(cfg->tau_r > 0.5 && cfg->tau_b > 0.5, "TauR/TauB must exceed 0.5");
    LBMPM_REQUIRE(cfg->outlet_rho_r > 0. && cfg->outlet_rho_b > 0., "densityRL/densityBL must be positive (the Zou-He outlet divides by them; the shipped ini uses 1e-8 for the absent colour)");
    LBMPM_REQUIRE((double)cfg->nx * cfg->ny * (cfg->nz_local + 2) < 2.0e9, "slab too large for 32-bit plane indices");
    LBMPM_REQUIRE((double)(cfg->nx + 31) * cfg->ny * 8.0 * 6 * Q < 4.0e9, "xy plane too large: 114 planes must fit 32-bit byte offsets (about 4.3M cells per plane)");
    int variant = cfg->variant, tile = 0, chunk_len = 32, fill = 16;
    // Environment switches of the PRODUCT library: the cross-checks LBMPM_RK3D_VARIANT (split sweeps), _LAYOUT (dense), _STORAGE (38 values
    // per cell) -- other statements of the same step, which the tests hold against each other.  The tuning knobs (_TILE, _CHUNK, _FILL,
    // _BOUNDARY, _XCC, _SLAB_SCHEDULE) exist in development builds only (-DLBMPM_DEV, openlbmpm_amd/build.py::build_dev).
    bool chunk_fixed = false;
    if (const char *e = getenv("LBMPM_RK3D_VARIANT")) variant = atoi(e);
#ifdef LBMPM_DEV
    if (const char *e = getenv("LBMPM_RK3D_TILE")) tile = atoi(e);
    if (const char *e = getenv("LBMPM_RK3D_CHUNK")) { chunk_len = atoi(e) > 0 ? atoi(e) : 32; chunk_fixed = true; }
    if (const char *e = getenv("LBMPM_RK3D_FILL")) fill = atoi(e);
#endif
    LBMPM_REQUIRE(variant == 0 || variant == 1, "lbmpm_rk3d_create: variant must be 0 (fused) or 1 (split)");
    LBMPM_REQUIRE(cfg->relaxation == 0 || cfg->relaxation == 1, "lbmpm_rk3d_create: relaxation must be 0 (SRT) or 1 (MRT)");
    LBMPM_REQUIRE(cfg->outlet_type == LBMPM_OUTLET_PRESSURE || cfg->outlet_type == LBMPM_OUTLET_CONVECTIVE, "lbmpm_rk3d_create: outlet_type must be LBMPM_OUTLET_PRESSURE or LBMPM_OUTLET_CONVECTIVE");
    if (cfg->outlet_type == LBMPM_OUTLET_CONVECTIVE) {
        LBMPM_REQUIRE(cfg->nz_global >= 8 && (cfg->z_offset == 0 ? cfg->nz_local >= 6 : cfg->z_offset >= 6),
                      "lbmpm_rk3d_create: the convective outlet needs nz >= 8 and the lattice's bottom slab to own the planes 0 .. 5 at least (slab [%lld, %lld))",
                      (long long)cfg->z_offset, (long long)(cfg->z_offset + cfg->nz_local));
        LBMPM_REQUIRE(cfg->variant == 0 && !getenv("LBMPM_RK3D_VARIANT"), "lbmpm_rk3d_create: the convective outlet runs the fused schedule only");
        if (cfg->z_offset == 0) {
            const size_t hp = (size_t)cfg->nx * cfg->ny;
            for (int z = 2; z <= 4; ++z)        // (with-halo array: global plane g is plane g + 1)
                LBMPM_REQUIRE(memcmp(is_domain_with_halo + hp, is_domain_with_halo + (size_t)z * hp, hp) == 0,
                              "lbmpm_rk3d_create: the convective outlet copies plane 3 onto the planes 2, 1, 0: their masks must coincide (plane %d differs from plane 0)", z - 1);
        }
    }
    LBMPM_REQUIRE(cfg->inlet_type == LBMPM_INLET_VELOCITY || (cfg->inlet_type == LBMPM_INLET_PRESSURE && cfg->inlet_rho_r > 0. && cfg->inlet_rho_b > 0.),
                  "lbmpm_rk3d_create: inlet_type must be LBMPM_INLET_VELOCITY or LBMPM_INLET_PRESSURE with positive densityRH / densityBH (the Zou-He pressure plane divides by them)");
    LBMPM_REQUIRE(cfg->recolor_axis >= 0. && cfg->recolor_diag >= 0. && cfg->recolor_axis < 1. && cfg->recolor_diag < 1.,
                  "lbmpm_rk3d_create: recolor_axis / recolor_diag must be 0 (the model's weights) or a weight in (0, 1)");
    LBMPM_REQUIRE(fill == 0 || fill == 4 || fill == 8 || fill == 16 || fill == 32 || fill == 64,
                  "LBMPM_RK3D_FILL must be 0 or a power of two <= 64");
    LBMPM_HIP_TRY(hipSetDevice(cfg->device));
    lbmpm_rk3d *c = new (std::nothrow) lbmpm_rk3d();
    if (!c) { set_error("out of host memory"); return LBMPM_ERR_NOMEM; }
    c->cfg = *cfg;
    c->nx = (int)cfg->nx; c->ny = (int)cfg->ny; c->nzl = (int)cfg->nz_local;
    c->variant = variant; c->tile = tile; c->chunk_len = chunk_len; c->fill = fill;
#ifdef LBMPM_DEV
    if (const char *e = getenv("LBMPM_RK3D_BOUNDARY")) c->boundary = atoi(e) >= 2 ? atoi(e) : 2;
#endif
    c->conv = cfg->outlet_type == LBMPM_OUTLET_CONVECTIVE;
    c->conv_lo = c->conv && cfg->z_offset == 0 ? 3 : 0;
    // compact storage: the 23-value form for any nx (row segments of <= 64 cells, seg_x0); the 38-value cross-check keeps whole 64-cell segments
    c->compact = variant == 0;
    if (const char *e = getenv("LBMPM_RK3D_LAYOUT")) if (!strcmp(e, "dense")) c->compact = false;
    c->q23 = c->compact && c->tile == 0;
    if (const char *e = getenv("LBMPM_RK3D_STORAGE")) if (atoi(e) == 38) c->q23 = false;
    if (!c->q23 && c->nx % 64 != 0) c->compact = false;
    if (c->conv && !c->q23) c->compact = false;      // (the 38-value compact kernels stage their row records too few planes ahead for the copied planes)
    // (the marching kernel packs a thread's lattice coordinates into 16-bit fields: rk3dq.h::sgeo)
    if (c->q23 && (c->nx > 32767 || c->ny > 32767)) {
        set_error("lbmpm_rk3d_create: nx and ny must not exceed 32767 (%d x %d)", c->nx, c->ny);
        delete c;
        return LBMPM_ERR_INVALID;
    }
#ifdef LBMPM_DEV      // timing knock-outs of the slab step (results wrong): development builds only
    if (const char *e = getenv("LBMPM_RK3D_DBG")) c->dbg = atoi(e);
#endif
    // q23 storage: the chunk length is chosen per launch (chunk_planes below) unless LBMPM_RK3D_CHUNK fixes it
    c->chunk_auto = c->q23 && !chunk_fixed;
    if (c->chunk_auto) { c->chunk_len = 64; (void)hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, cfg->device); if (c->ncu <= 0) c->ncu = 256; }
#ifdef LBMPM_DEV
    if (c->q23 && getenv("LBMPM_RK3D_TRACE")) { (void)hipMalloc(reinterpret_cast<void **>(&c->trace), (size_t)1 << 22); (void)hipMemset(c->trace, 0, (size_t)1 << 22); }
#endif
    c->pitch = (c->nx + 31) / 32 * 32;
    c->plane2 = (size_t)c->pitch * c->ny;
    c->vol = c->plane2 * (size_t)(c->nzl + 2);
    const size_t hp = (size_t)c->nx * c->ny;
    c->h_domain.assign(is_domain_with_halo + hp, is_domain_with_halo + hp * (size_t)(c->nzl + 1));
    std::vector<uint8_t> hflags(c->vol, 0);
    for (int z = 0; z < c->nzl + 2; ++z)
        for (int y = 0; y < c->ny; ++y)
            for (int x = 0; x < c->nx; ++x) {
                const uint8_t v = is_domain_with_halo[(size_t)z * hp + (size_t)y * c->nx + x] == 1 ? 1 : 0;
                hflags[(size_t)z * c->plane2 + (size_t)y * c->pitch + x] = v;
                if (z >= 1 && z <= c->nzl) c->nfluid += v;
            }
    if (c->nfluid == 0) { set_error("lbmpm_rk3d_create: the slab has no fluid node (is_domain == 1 marks fluid); cut the lattice elsewhere"); delete c; return LBMPM_ERR_INVALID; }
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return LBMPM_ERR_HIP; }
        c->own_stream = true;
    }
    int rc = LBMPM_OK;
#define TRY_RC(e) do { rc = (e); if (rc != LBMPM_OK) { lbmpm_rk3d_destroy(c); return rc; } } while (0)
    TRY_RC(dev_alloc(c, &c->flags, c->vol));
    TRY_RC(dev_alloc(c, &c->solidnbr, c->vol));
    std::vector<uint32_t> hseg, hseg2;
    if (c->compact) {
        // segment records of the compact storage; numbering: plane, tile (64 x TILE_ROWS), row, x
        c->nseg = (c->nx + 63) / 64;
        const int ns = c->nseg;
        auto sx0 = [&](int sg) { return seg_x0(sg, c->nx, ns); };
        const size_t rows = (size_t)(c->nzl + 2) * c->ny;
        hseg.assign(rows * ns * 4, 0u);
        hseg2.assign(rows * ns * 4, 0u);
        c->h_pstart.assign(c->nzl + 3, 0ull);
        c->h_rowpop.assign(rows * ns, 0);
        unsigned long long total = 0;
        std::vector<unsigned long long> m((size_t)c->ny * ns);
        std::vector<unsigned> first((size_t)c->ny * ns);
        for (int z = 0; z < c->nzl + 2; ++z) {
            c->h_pstart[z] = total;
            for (int y = 0; y < c->ny; ++y)
                for (int sg = 0; sg < ns; ++sg) {
                    const uint8_t *fl = hflags.data() + (size_t)z * c->plane2 + (size_t)y * c->pitch + sx0(sg);
                    unsigned long long w = 0;
                    for (int b = 0; b < sx0(sg + 1) - sx0(sg); ++b)
                        if (fl[b]) w |= 1ull << b;
                    m[(size_t)y * ns + sg] = w;
                }
            unsigned j = 0;
            for (int y0 = 0; y0 < c->ny; y0 += TILE_ROWS)
                for (int sg = 0; sg < ns; ++sg) {
                    int last = -1;
                    for (int y = y0; y < y0 + TILE_ROWS && y < c->ny; ++y) {
                        first[(size_t)y * ns + sg] = j;
                        j += (unsigned)__builtin_popcountll(m[(size_t)y * ns + sg]);
                        last = y;
                    }
                    const unsigned lc = c->q23 ? (unsigned)QLINE : 8u;   // cells per 128-byte line
                    const unsigned pad = (lc - (j % lc)) % lc;          // whole 128-byte lines per tile run
                    hseg[(((size_t)z * c->ny + last) * ns + sg) * 4 + 3] = pad << 2;
                    j += pad;
                }
            // "odd stride": the 19 direction regions of a plane lie j stored cells apart.  Where that is a multiple of 16 KiB (regular lattices:
            // an all-fluid 512^2 plane puts them exactly 2 MiB apart) a wave's 19 pulls fall on the same channels and banks -- measured on the
            // kernel's access shape: 15.5 -> 16.3 G cells/s with 9 KB of padding per region (profiles/r06_march3d_layouts.txt).  Porous planes
            // never meet it; regular ones get 73 x 16 unused cells behind their last tile.
            if (j > 0 && ((size_t)j * (c->q23 ? 8u : 16u)) % 16384u == 0) j += 73u * 16u;
            for (int y = 0; y < c->ny; ++y)
                for (int sg = 0; sg < ns; ++sg) {
                    const size_t r = ((size_t)z * c->ny + y) * ns + sg;
                    const int sl = sg > 0 ? sg - 1 : ns - 1, sr = sg + 1 < ns ? sg + 1 : 0;
                    const unsigned long long ml = m[(size_t)y * ns + sl], mr = m[(size_t)y * ns + sr];
                    c->h_rowpop[r] = (uint8_t)__builtin_popcountll(m[(size_t)y * ns + sg]);
                    hseg[r * 4 + 0] = (uint32_t)m[(size_t)y * ns + sg];
                    hseg[r * 4 + 1] = (uint32_t)(m[(size_t)y * ns + sg] >> 32);
                    hseg[r * 4 + 2] = first[(size_t)y * ns + sg];
                    const int wl = sx0(sl + 1) - sx0(sl), wd = sx0(sg + 1) - sx0(sg);       // cells of the segment to the left, of this one
                    hseg[r * 4 + 3] |= (uint32_t)((ml >> (wl - 1)) & 1ull) | ((uint32_t)(mr & 1ull) << 1) | ((uint32_t)(64 - wd) << 8);
                    hseg2[r * 4 + 0] = first[(size_t)y * ns + sl] + (unsigned)__builtin_popcountll(ml) - 1u;   // j of cell x0-1 (if fluid)
                    hseg2[r * 4 + 1] = first[(size_t)y * ns + sr];                                             // j of cell x0+64 (if fluid)
                }
            total += j;
        }
        c->h_pstart[c->nzl + 2] = total;
        c->ncells = (size_t)total;
        TRY_RC(dev_alloc(c, &c->seg, hseg.size()));
        TRY_RC(dev_alloc(c, &c->seg2, hseg2.size()));
        TRY_RC(dev_alloc(c, &c->pstart, c->h_pstart.size()));
    }
    const size_t fcount = c->q23 ? (size_t)QS * (c->ncells + 2) : (c->compact ? 2 * Q * (c->ncells + 1) : 2 * Q * c->vol);
    TRY_RC(dev_alloc(c, &c->fA, fcount));
    TRY_RC(dev_alloc(c, &c->fB, fcount));
    if (c->q23) {
        bool by_xcc = true;       // tiles handed out by the XCD a workgroup runs on (rk3dq_fused); LBMPM_RK3D_XCC=0 (development builds): by block index
#ifdef LBMPM_DEV
        if (getenv("LBMPM_RK3D_XCC") && atoi(getenv("LBMPM_RK3D_XCC")) == 0) by_xcc = false;
#endif
        if (by_xcc) TRY_RC(dev_alloc(c, &c->slotq, 2 * 4096 * 8));     // zeroed
        // (row segments of a slab are indexed with 32 bits in the kernels: rk3dq.h::row_index)
        LBMPM_REQUIRE((size_t)(c->nzl + 2) * c->ny * c->nseg < ((size_t)1 << 31), "rk3d: more than 2^31 row segments in one slab");
        TRY_RC(dev_alloc(c, &c->purA, (size_t)(c->nzl + 2) * c->ny * c->nseg));
        TRY_RC(dev_alloc(c, &c->purB, (size_t)(c->nzl + 2) * c->ny * c->nseg));
    }
    TRY_RC(dev_alloc(c, &c->phi, c->vol));
    // q23: 13 per cell + the row flags (rk3dq.h).  Per STORED cell of a plane: the tile padding and the odd-stride padding count (a regular
    // plane's cnt exceeds its nx * ny cells), so the buffers follow the largest plane
    size_t max_cnt = c->plane2;
    if (c->compact) for (int z = 0; z < c->nzl + 2; ++z) max_cnt = std::max(max_cnt, (size_t)(c->h_pstart[z + 1] - c->h_pstart[z]));
    const size_t face_doubles = (c->q23 ? FACE_DOUBLES + 1 : 10) * max_cnt;
    TRY_RC(dev_alloc(c, &c->send_up, face_doubles));
    TRY_RC(dev_alloc(c, &c->send_dn, face_doubles));
    TRY_RC(dev_alloc(c, &c->recv_below, face_doubles));
    TRY_RC(dev_alloc(c, &c->recv_above, face_doubles));
#undef TRY_RC
    hipError_t e = hipMemcpyAsync(c->flags, hflags.data(), c->vol, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && c->compact) e = hipMemcpyAsync(c->seg, hseg.data(), hseg.size() * sizeof(hseg[0]), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && c->compact) e = hipMemcpyAsync(c->seg2, hseg2.data(), hseg2.size() * sizeof(hseg2[0]), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && c->compact) e = hipMemcpyAsync(c->pstart, c->h_pstart.data(), c->h_pstart.size() * sizeof(c->h_pstart[0]), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_error("flags upload failed: %s", hipGetErrorString(e)); lbmpm_rk3d_destroy(c); return LBMPM_ERR_HIP; }
    RK3Dev p = make_dev(c);
    rk3d_setup_solidnbr<<<grid3(c, c->nzl + 2), dim3(BX3, BY3), 0, c->stream>>>(p, c->solidnbr);
    rk3d_init_phi<<<dim3((unsigned)((c->vol + 255) / 256)), dim3(256), 0, c->stream>>>(p, c->phi);
    e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("set-up kernel failed: %s", hipGetErrorString(e)); lbmpm_rk3d_destroy(c); return LBMPM_ERR_HIP; }
    *out = c;
    return LBMPM_OK;
}

extern "C" void lbmpm_rk3d_destroy(lbmpm_rk3d *c)
{
    if (!c) return;
    // first the device and the streams (unpack kernels or peer copies may still be queued on them), then the transport's memory
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    c->tx.disconnect();
    if (c->probe_bad) (void)hipFree(c->probe_bad);
    if (c->beat_host) (void)hipHostFree(c->beat_host);
    if (c->wd_stream) (void)hipStreamDestroy(c->wd_stream);
    for (void *ptr : {(void *)c->seg, (void *)c->seg2, (void *)c->pstart, (void *)c->flags, (void *)c->solidnbr, (void *)c->fA, (void *)c->fB, (void *)c->purA, (void *)c->purB, (void *)c->trace, (void *)c->slotq, (void *)c->phi, (void *)c->diag,
                      (void *)c->send_up, (void *)c->send_dn, (void *)c->recv_below, (void *)c->recv_above})
        if (ptr) (void)hipFree(ptr);
    c->pool.destroy();
    c->slab_pool.destroy();
    if (c->aux) { (void)hipStreamDestroy(c->aux); (void)hipEventDestroy(c->ev_dep); (void)hipEventDestroy(c->ev_done); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int lbmpm_rk3d_set_stream(lbmpm_rk3d *c, void *hip_stream)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (hip_stream == nullptr) {
        if (!c->own_stream) { LBMPM_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        return LBMPM_OK;
    }
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = static_cast<hipStream_t>(hip_stream);
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_set_density(lbmpm_rk3d *c, const double *rho_r, const double *rho_b)
{
    LBMPM_REQUIRE(c && rho_r && rho_b, "lbmpm_rk3d_set_density: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    // stage the two density fields on the device, expand there
    const size_t n = (size_t)c->nx * c->ny * c->nzl;
    const size_t fbytes = (c->q23 ? (size_t)QS * (c->ncells + 2) : (c->compact ? 2 * Q * (c->ncells + 1) : 2 * Q * c->vol)) * sizeof(double);
    double *stage = nullptr;
    LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&stage), 2 * n * sizeof(double)));
    hipError_t e = hipMemcpyAsync(stage, rho_r, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(stage + n, rho_b, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->fA, 0, fbytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->fB, 0, fbytes, c->stream);
    if (e == hipSuccess) {
        RK3Dev p = make_dev(c);
        if (c->q23) {
            // both buffers start from the same image: the ghost planes are never written again (rk3dq.h), and a segment without a
            // fluid cell keeps the flag "any colour" (all bits) for good
            const size_t pbytes = (size_t)(c->nzl + 2) * c->ny * c->nseg * sizeof(uint32_t);
            e = hipMemsetAsync(c->purA, 0xff, pbytes, c->stream);
            rk3dq_init_rest<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p, stage, stage + n, c->fA, c->purA);
            if (e == hipSuccess) e = hipMemcpyAsync(c->fB, c->fA, fbytes, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(c->purB, c->purA, pbytes, hipMemcpyDeviceToDevice, c->stream);
        }
        else if (c->compact) rk3dc_init_rest<<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p, stage, stage + n, c->fA);
        else rk3d_init_rest<<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p, stage, stage + n, c->fA);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) { set_error("lbmpm_rk3d_set_density: %s", hipGetErrorString(e)); return LBMPM_ERR_HIP; }
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    c->streamed = false;
    c->halo_valid = false;
    c->steps = 0;
    c->observed_at = -1;
    return LBMPM_OK;
}

// ---- state in and out (rk3d_state.h): macroscopic start, populations per colour, the stored values themselves
namespace {
template <int MODE>
void launch_state_io(lbmpm_rk3d *c, const RK3Dev &p, double *f, int zl0, int planes, double *a, double *b)
{
    const dim3 block(BX3, BY3);
    const unsigned gy = (unsigned)((c->ny + BY3 - 1) / BY3);
    if (c->q23) rk3d_state_io<ST_Q23, MODE><<<dim3(c->nseg, gy, planes), block, 0, c->stream>>>(p, f, zl0, a, b);
    else if (c->compact) rk3d_state_io<ST_C38, MODE><<<dim3(c->nseg, gy, planes), block, 0, c->stream>>>(p, f, zl0, a, b);
    else rk3d_state_io<ST_DENSE, MODE><<<dim3((c->nx + 63) / 64, gy, planes), block, 0, c->stream>>>(p, f, zl0, a, b);
}

int state_width(const lbmpm_rk3d *c) { return c->q23 ? QS : 2 * Q; }
size_t state_fbytes(const lbmpm_rk3d *c) { return (c->q23 ? (size_t)QS * (c->ncells + 2) : (c->compact ? 2 * Q * (c->ncells + 1) : 2 * Q * c->vol)) * sizeof(double); }

// The owned planes in batches through a staging buffer on the device.  wa / wb: doubles per cell of the host arrays ha / hb (MACRO: the
// five host arrays m[0..4], one double per cell each; a NULL velocity component is zero).
template <int MODE>
int state_transfer(lbmpm_rk3d *c, double *ha, int wa, double *hb, int wb, const double *const *m)
{
    constexpr bool GET = MODE == IO_GET_STATE || MODE == IO_GET_PDF;
    LBMPM_REQUIRE(!c->interior_pending, "state transfer inside a step: finish it with lbmpm_rk3d_collide_boundary first");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->aux) LBMPM_HIP_TRY(hipStreamSynchronize(c->aux));
    const size_t pcells = (size_t)c->nx * c->ny;
    const int per = MODE == IO_SET_MACRO ? 5 : wa + wb;
    int batch = (int)(((size_t)256 << 20) / (pcells * (size_t)per * sizeof(double)));
    batch = batch < 1 ? 1 : (batch > c->nzl ? c->nzl : batch);
    double *stage = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&stage), (size_t)batch * pcells * per * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        set_error("state transfer: hipMalloc of the %d-plane staging buffer failed", batch);
        return LBMPM_ERR_NOMEM;
    }
    hipError_t e = hipSuccess;
    if (!GET) e = hipMemsetAsync(c->fA, 0, state_fbytes(c), c->stream);
    const RK3Dev p = make_dev(c);
    for (int z0 = 0; z0 < c->nzl && e == hipSuccess; z0 += batch) {
        const int n = c->nzl - z0 < batch ? c->nzl - z0 : batch;
        const size_t cells = (size_t)n * pcells;
        double *a = stage, *b = stage + cells * (size_t)wa;
        if (MODE == IO_SET_MACRO) {
            for (int k = 0; k < 5 && e == hipSuccess; ++k)
                e = m[k] ? hipMemcpyAsync(stage + (size_t)k * cells, m[k] + (size_t)z0 * pcells, cells * sizeof(double), hipMemcpyHostToDevice, c->stream)
                         : hipMemsetAsync(stage + (size_t)k * cells, 0, cells * sizeof(double), c->stream);
        } else if (!GET) {
            e = hipMemcpyAsync(a, ha + (size_t)z0 * pcells * wa, cells * wa * sizeof(double), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && wb) e = hipMemcpyAsync(b, hb + (size_t)z0 * pcells * wb, cells * wb * sizeof(double), hipMemcpyHostToDevice, c->stream);
        }
        if (e != hipSuccess) break;
        launch_state_io<MODE>(c, p, c->fA, z0 + 1, n, a, b);
        e = hipGetLastError();
        if (e == hipSuccess && GET) {
            e = hipMemcpyAsync(ha + (size_t)z0 * pcells * wa, a, cells * wa * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && wb) e = hipMemcpyAsync(hb + (size_t)z0 * pcells * wb, b, cells * wb * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);       // the staging buffer is reused by the next batch
    }
    if (e == hipSuccess && !GET) {
        if (c->q23) {
            // row flags from the records; the ghost planes' rows and segments without a fluid cell keep "any colour" for good (as set_density)
            const size_t pbytes = (size_t)(c->nzl + 2) * c->ny * c->nseg * sizeof(uint32_t);
            e = hipMemsetAsync(c->purA, 0xff, pbytes, c->stream);
            rk3dq_flags_from_records<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p, c->fA, c->purA, 1);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(c->purB, c->purA, pbytes, hipMemcpyDeviceToDevice, c->stream);
        }
        // both buffers start from the same image (the ghost planes of the q23 storage are never written again)
        if (e == hipSuccess) e = hipMemcpyAsync(c->fB, c->fA, state_fbytes(c), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(stage);
    if (e != hipSuccess) { set_error("state transfer: %s", hipGetErrorString(e)); return LBMPM_ERR_HIP; }
    return LBMPM_OK;
}

void state_was_set(lbmpm_rk3d *c, int64_t steps, bool streamed)
{
    c->streamed = streamed;
    c->halo_valid = false;
    c->steps = steps;
    c->observed_at = -1;
}
}  // namespace

extern "C" int lbmpm_rk3d_set_macro(lbmpm_rk3d *c, const double *rho_r, const double *rho_b, const double *vx, const double *vy, const double *vz)
{
    LBMPM_REQUIRE(c && rho_r && rho_b, "lbmpm_rk3d_set_macro: null argument");
    const double *m[5] = {rho_r, rho_b, vx, vy, vz};
    const int rc = state_transfer<IO_SET_MACRO>(c, nullptr, 0, nullptr, 0, m);
    if (rc == LBMPM_OK) state_was_set(c, 0, false);
    return rc;
}

extern "C" int lbmpm_rk3d_set_pdf(lbmpm_rk3d *c, const double *pdf_r, const double *pdf_b, int post_collision)
{
    LBMPM_REQUIRE(c && pdf_r && pdf_b, "lbmpm_rk3d_set_pdf: null argument");
    const int rc = state_transfer<IO_SET_PDF>(c, const_cast<double *>(pdf_r), Q, const_cast<double *>(pdf_b), Q, nullptr);
    if (rc == LBMPM_OK) state_was_set(c, 0, post_collision != 0);
    return rc;
}

extern "C" int lbmpm_rk3d_get_pdf(lbmpm_rk3d *c, double *pdf_r, double *pdf_b)
{
    LBMPM_REQUIRE(c && pdf_r && pdf_b, "lbmpm_rk3d_get_pdf: null argument");
    return state_transfer<IO_GET_PDF>(c, pdf_r, Q, pdf_b, Q, nullptr);
}

extern "C" int lbmpm_rk3d_state_info(const lbmpm_rk3d *c, int64_t *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk3d_state_info: null argument");
    out[0] = state_width(c); out[1] = c->steps; out[2] = c->streamed ? 1 : 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_get_state(lbmpm_rk3d *c, double *state)
{
    LBMPM_REQUIRE(c && state, "lbmpm_rk3d_get_state: null argument");
    return state_transfer<IO_GET_STATE>(c, state, state_width(c), nullptr, 0, nullptr);
}

extern "C" int lbmpm_rk3d_set_state(lbmpm_rk3d *c, const double *state, int64_t doubles_per_cell, int64_t steps_done, int post_collision)
{
    LBMPM_REQUIRE(c && state && steps_done >= 0, "lbmpm_rk3d_set_state: bad argument");
    LBMPM_REQUIRE(doubles_per_cell == state_width(c), "lbmpm_rk3d_set_state: the state holds %lld doubles per cell, this context stores %d (lbmpm_rk3d_state_info; "
                  "a state of the other storage goes through lbmpm_rk3d_get_pdf / set_pdf)", (long long)doubles_per_cell, state_width(c));
    const int rc = state_transfer<IO_SET_STATE>(c, const_cast<double *>(state), state_width(c), nullptr, 0, nullptr);
    if (rc == LBMPM_OK) state_was_set(c, steps_done, post_collision != 0);
    return rc;
}

// bytes of the face message that describes plane zl (compact storages move the fluid cells of the plane only)
static int64_t face_bytes(const lbmpm_rk3d *c, int zl)
{
    if (c->q23) return (int64_t)(((c->h_pstart[zl + 1] - c->h_pstart[zl]) * FACE_DOUBLES + ((size_t)c->ny * c->nseg + 1) / 2) * sizeof(double));
    return c->compact ? (int64_t)((c->h_pstart[zl + 1] - c->h_pstart[zl]) * 10 * sizeof(double)) : (int64_t)(10 * c->plane2 * sizeof(double));
}

extern "C" int lbmpm_rk3d_pack_halo(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    if (c->q23) {       // the whole face message of the current state (rk3dq.h): populations, records, class sums, row flags
        const int below = c->cfg.z_offset > 0, above = c->cfg.z_offset + c->cfg.nz_local < c->cfg.nz_global;
        if (!below && !above) return LBMPM_OK;
        RK3Dev q = make_dev(c);
        rk3dq_face_pack<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3, 2), dim3(BX3, BY3), 0, c->stream>>>(q, c->send_up, c->send_dn, below, above);
        LBMPM_HIP_TRY(hipGetLastError());
        return LBMPM_OK;
    }
    RK3Dev p = make_dev(c);
    const int threads = 256;
    const dim3 grid((unsigned)((c->plane2 + threads - 1) / threads));
    if (c->compact) rk3dc_pack<<<grid, dim3(threads), 0, c->stream>>>(p, c->fA, c->send_up, c->send_dn);
    else rk3d_pack<<<grid, dim3(threads), 0, c->stream>>>(p, c->fA, c->send_up, c->send_dn);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_unpack_halo(lbmpm_rk3d *c, int have_below, int have_above)
{
    LBMPM_REQUIRE(c, "null context");
    if (c->q23) {
        if (!have_below && !have_above) return LBMPM_OK;
        RK3Dev q = make_dev(c);
        const dim3 grid(c->nseg, (c->ny + BY3 - 1) / BY3, 2), block(BX3, BY3);
        rk3dq_face_unpack<<<grid, block, 0, c->stream>>>(q, c->fA, c->purA, c->recv_below, c->recv_above, have_below, have_above);
        rk3dq_halo_phi<<<grid, block, 0, c->stream>>>(q, c->recv_below, c->recv_above, have_below, have_above);
        LBMPM_HIP_TRY(hipGetLastError());
        c->halo_valid = true;
        return LBMPM_OK;
    }
    RK3Dev p = make_dev(c);
    const int threads = 256;
    const dim3 grid((unsigned)((c->plane2 + threads - 1) / threads));
    if (c->compact) rk3dc_unpack<<<grid, dim3(threads), 0, c->stream>>>(p, c->fA, c->recv_below, c->recv_above, have_below, have_above);
    else rk3d_unpack<<<grid, dim3(threads), 0, c->stream>>>(p, c->fA, c->recv_below, c->recv_above, have_below, have_above);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_phase_field(lbmpm_rk3d *c, int with_diagnostics)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (with_diagnostics && !c->diag) { const int rc = dev_alloc(c, &c->diag, 5 * c->vol); if (rc) return rc; }
    RK3Dev p = make_dev(c);
    p.diag = with_diagnostics ? c->diag : nullptr;
    auto k1 = [&](int planes, int zl0) {
        if (c->q23) rk3dq_phase_field<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3, planes), dim3(BX3, BY3), 0, c->stream>>>(p, zl0);
        else if (c->compact) rk3dc_phase_field<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3, planes), dim3(BX3, BY3), 0, c->stream>>>(p, zl0);
        else rk3d_phase_field<<<grid3(c, planes), dim3(BX3, BY3), 0, c->stream>>>(p, zl0);
    };
    if (with_diagnostics) c->observed_at = c->steps;
    if (c->variant == 1 || with_diagnostics) k1(c->nzl, 1);
    else if (c->q23) { /* the halo planes' phase field comes with the face exchange (rk3dq_halo_phi) */ }
    else {
        // fused variant: the marching kernel computes the phase field itself; only the planes a
        // neighbour rank needs are produced here
        if (c->cfg.z_offset > 0) k1(1, 1);
        if (c->cfg.z_offset + c->cfg.nz_local < c->cfg.nz_global && (c->nzl > 1 || c->cfg.z_offset == 0)) k1(1, c->nzl);
    }
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

namespace {
// run f(bool_constant<a>, bool_constant<b>): two run-time switches -> template arguments
template <typename F>
void dispatch2(bool a, bool b, F &&f)
{
    if (a) { if (b) f(std::true_type{}, std::true_type{}); else f(std::true_type{}, std::false_type{}); }
    else   { if (b) f(std::false_type{}, std::true_type{}); else f(std::false_type{}, std::false_type{}); }
}

template <int TX, int TY>
void launch_fused(lbmpm_rk3d *c, const RK3Dev &p, hipStream_t st, int z_first, int z_last)
{
    const int tilesX = (c->nx + TX - 1) / TX, tilesY = (c->ny + TY - 1) / TY, rpx = (tilesY + 7) / 8;
    const int nchunks = (z_last - z_first + 1 + c->chunk_len - 1) / c->chunk_len;
    const dim3 grid((unsigned)(8 * tilesX * rpx * nchunks)), block(TX * TY);
    auto go = [&](auto first, auto mrt) {
        rk3d_fused<TX, TY, decltype(first)::value, decltype(mrt)::value><<<grid, block, 0, st>>>(p, tilesX, tilesY, rpx, c->chunk_len, z_first, z_last);
    };
    dispatch2(p.first != 0, p.mrt != 0, go);
}

template <int TY>
void launch_fused_c(lbmpm_rk3d *c, const RK3Dev &p, hipStream_t st, int z_first, int z_last)
{
    const int tilesX = c->nseg, tilesY = (c->ny + TY - 1) / TY, rpx = (tilesY + 7) / 8;
    const int nchunks = (z_last - z_first + 1 + c->chunk_len - 1) / c->chunk_len;
    const dim3 grid((unsigned)(8 * tilesX * rpx * nchunks)), block(64 * TY);
    auto go = [&](auto first, auto mrt) {
        rk3dc_fused<TY, decltype(first)::value, decltype(mrt)::value><<<grid, block, 0, st>>>(p, tilesX, tilesY, rpx, c->chunk_len, z_first, z_last);
    };
    dispatch2(p.first != 0, p.mrt != 0, go);
}

// q23 storage: planes z_first..z_last and (if z_last2 >= z_first2) z_first2..z_last2 in ONE launch
// Planes per workgroup of a launch over n planes.  A march costs two fill steps per chunk, so chunks should be long; but the launch
// should still hand every CU several workgroups.  Measured at 512^3 (512 tiles, MRT): chunks of 16 / 32 / 64 / 128 / 256 / 512 planes
// 7.87 / 7.32 / 7.08 / 6.91 / 6.72 / 6.81 ms per step -> as few chunks as keep ~4 workgroups per CU in the launch, none shorter than
// 64 planes unless the range is (or the lattice is small, below), all of equal length.
int chunk_planes(const lbmpm_rk3d *c, int n)
{
    if (!c->chunk_auto) return c->chunk_len;
    const int tiles = c->nseg * ((c->ny + 7) / 8);
    const int want = (4 * c->ncu + tiles - 1) / tiles;              // chunks for ~4 workgroups per CU
    int len = (n + want - 1) / want;
    // none shorter than 64 planes -- unless the lattice is so small that chunks of 64 leave CUs without a workgroup (the reference's own
    // 32 x 32 x 96 ini: four tiles): then rather shorter chunks, down to 8 planes, than idle CUs (100^3 porous: 0.32 -> 0.16 ms per step)
    int minlen = 64;
    while (minlen > 8 && tiles * ((n + minlen - 1) / minlen) < c->ncu) minlen /= 2;
    if (len < minlen) len = minlen;
    const int nch = (n + len - 1) / len;
    return nch > 0 ? (n + nch - 1) / nch : 64;
}

void launch_q23(lbmpm_rk3d *c, const RK3Dev &p, hipStream_t st, int z_first, int z_last, int z_first2, int z_last2)
{
    const int tilesX = c->nseg, tilesY = (c->ny + 7) / 8, rpx = (tilesY + 7) / 8;
    const int n1 = z_last - z_first + 1, n2 = z_last2 >= z_first2 ? z_last2 - z_first2 + 1 : 0;
    const int chunk_len = chunk_planes(c, n1 > n2 ? n1 : n2);
    const int nchunks1 = (n1 + chunk_len - 1) / chunk_len;
    const int nchunks2 = n2 > 0 ? (n2 + chunk_len - 1) / chunk_len : 0;
    const dim3 grid((unsigned)(8 * tilesX * rpx * (nchunks1 + nchunks2))), block(512);
    // tile counters of this launch: slice after slice of two rings of 4096 x 8 words that are zeroed wholesale, a ring while the other
    // one is half used up (no memset per launch: that is a kernel of its own)
    unsigned *q = nullptr;
    if (c->slotq) {
        const unsigned n = c->slot_launches++, ring = (n / 4096u) & 1u, i = n % 4096u;
        if (i == 2048u) (void)hipMemsetAsync(c->slotq + (size_t)(ring ^ 1u) * 4096u * 8u, 0, 4096u * 8u * sizeof(unsigned), st);
        q = c->slotq + ((size_t)ring * 4096u + i) * 8u;
    }
    auto go = [&](auto first, auto mrt) {
        constexpr bool F = decltype(first)::value, M = decltype(mrt)::value;
        auto launch = [&](auto ragged, auto pin) {
            rk3dq_fused<F, M, decltype(ragged)::value, decltype(pin)::value><<<grid, block, 0, st>>>(p, tilesX, tilesY, rpx, chunk_len, z_first, z_last,
                                                                                                     nchunks1, z_first2, z_last2, q);
        };
        dispatch2(c->nx % 64 != 0, p.inletP != 0, launch);
    };
    dispatch2(p.first != 0, p.mrt != 0, go);
}

// planes z_first..z_last of the time step on stream st
void launch_step_range(lbmpm_rk3d *c, const RK3Dev &p, hipStream_t st, int z_first, int z_last)
{
    if (z_last < z_first) return;
    if (c->q23 && c->conv_lo > 0) {
        // convective outlet (rk3dq.h, "convective outlet"): the launch that owns plane zs = conv_lo + 1 also does the planes below it
        const int zs = c->conv_lo + 1;
        if (z_last < zs) return;                           // (a range of copied planes only: done with plane zs)
        if (z_first <= zs) {
            const dim3 block(BX3, BY3);
            const unsigned gy = (unsigned)((c->ny + BY3 - 1) / BY3);
            RK3Dev q = p;
            q.diag = nullptr;
            rk3dq_phase_field<<<dim3(c->nseg, gy, 1), block, 0, st>>>(q, zs);
            rk3dq_conv_phi<<<dim3((unsigned)((c->plane2 + 255) / 256)), dim3(256), 0, st>>>(q, zs);
            RK3Dev m = p;
            m.halo_lo = c->conv_lo;
            launch_q23(c, m, st, zs, z_last, 1, 0);
            dispatch2(p.first != 0, p.mrt != 0, [&](auto first, auto mrt) {
                rk3dq_conv_collide<decltype(first)::value, decltype(mrt)::value><<<dim3(c->nseg, gy, 2), block, 0, st>>>(p, zs);
            });
            return;
        }
    }
    if (c->q23) { launch_q23(c, p, st, z_first, z_last, 1, 0); return; }
    if (c->compact) {
        if (c->tile == 1) launch_fused_c<4>(c, p, st, z_first, z_last);
        else launch_fused_c<8>(c, p, st, z_first, z_last);
        return;
    }
    if (c->tile == 1) launch_fused<64, 4>(c, p, st, z_first, z_last);        // 3 blocks/CU, 55 % rim
    else if (c->tile == 2) launch_fused<32, 8>(c, p, st, z_first, z_last);
    else launch_fused<64, 8>(c, p, st, z_first, z_last);                     // default: 1 block/CU, 29 % rim
}

void finish_step(lbmpm_rk3d *c)
{
    std::swap(c->fA, c->fB);
    std::swap(c->purA, c->purB);
    c->streamed = true;
    c->steps += 1;
    c->halo_valid = false;      // the halo planes held the previous state's neighbours; the pipelined slab step sets it again after its unpack
}
}  // namespace

extern "C" int lbmpm_rk3d_collide(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_REQUIRE(!c->interior_pending, "lbmpm_rk3d_collide after lbmpm_rk3d_collide_interior: finish the step with lbmpm_rk3d_collide_boundary");
    RK3Dev p = make_dev(c);
    if (c->variant == 1) {
        if (p.mrt) rk3d_collide<true><<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p);
        else rk3d_collide<false><<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p);
    }
    else launch_step_range(c, p, c->stream, 1, c->nzl);
    LBMPM_HIP_TRY(hipGetLastError());
    finish_step(c);
    return LBMPM_OK;
}

// Overlap of the halo exchange with the bulk of the step (fused variant): the planes at least
// `boundary` planes away from both faces of the slab need neither the neighbours' populations nor
// their phase field, so they are collided on a second stream while the caller packs, exchanges
// and unpacks on the context's stream; lbmpm_rk3d_collide_boundary then does the planes next to
// the faces and joins the two streams.
static int collide_interior_ev(lbmpm_rk3d *c, hipEvent_t e0, hipEvent_t e1);
static int collide_boundary_ev(lbmpm_rk3d *c, hipEvent_t e0, hipEvent_t e1);
extern "C" int lbmpm_rk3d_collide_interior(lbmpm_rk3d *c) { return collide_interior_ev(c, nullptr, nullptr); }
extern "C" int lbmpm_rk3d_collide_boundary(lbmpm_rk3d *c) { return collide_boundary_ev(c, nullptr, nullptr); }

static int collide_interior_ev(lbmpm_rk3d *c, hipEvent_t e0, hipEvent_t e1)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_REQUIRE(!c->interior_pending, "lbmpm_rk3d_collide_interior called twice in one step");
    const int cb = c->boundary;
    if (c->variant == 1 || c->nzl < 2 * cb + 1) return LBMPM_OK;       // nothing to overlap: collide_boundary does the whole slab
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->aux) {
        LBMPM_HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        LBMPM_HIP_TRY(hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming));
        LBMPM_HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    }
    LBMPM_HIP_TRY(hipEventRecord(c->ev_dep, c->stream));                // everything issued so far (previous step) first
    LBMPM_HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_dep, 0));
    RK3Dev p = make_dev(c);
    if (e0) LBMPM_HIP_TRY(hipEventRecord(e0, c->aux));
    launch_step_range(c, p, c->aux, cb + 1, c->nzl - cb);
    LBMPM_HIP_TRY(hipGetLastError());
    if (e1) LBMPM_HIP_TRY(hipEventRecord(e1, c->aux));
    LBMPM_HIP_TRY(hipEventRecord(c->ev_done, c->aux));
    c->interior_pending = true;
    return LBMPM_OK;
}

static int collide_boundary_ev(lbmpm_rk3d *c, hipEvent_t e0, hipEvent_t e1)
{
    LBMPM_REQUIRE(c, "null context");
    if (!c->interior_pending) {
        if (e0) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
        const int rc = lbmpm_rk3d_collide(c);
        if (rc == LBMPM_OK && e1) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
        return rc;
    }
    const int cb = c->boundary;
    RK3Dev p = make_dev(c);
    if (e0) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
    launch_step_range(c, p, c->stream, 1, cb);
    launch_step_range(c, p, c->stream, c->nzl - cb + 1, c->nzl);
    LBMPM_HIP_TRY(hipGetLastError());
    if (e1) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    LBMPM_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done, 0));
    c->interior_pending = false;
    finish_step(c);
    return LBMPM_OK;
}

// ---- transports of the slab exchange inside the library (include/lbmpm.h; rk3d_transport.h)
static int tx_shape(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c->q23, "the in-library transports move the one-exchange face message of the compact 23-value storage (LBMPM_RK3D_STORAGE, _LAYOUT, _TILE, _VARIANT unset)");
    const bool below = c->cfg.z_offset > 0, above = c->cfg.z_offset + c->cfg.nz_local < c->cfg.nz_global;
    return c->tx.set_shape(c->cfg.device, below, above, (size_t)face_bytes(c, c->nzl), (size_t)face_bytes(c, 1), (size_t)face_bytes(c, 0),
                           (size_t)face_bytes(c, c->nzl + 1));
}

extern "C" int lbmpm_rk3d_ipc_init(lbmpm_rk3d *c, void *blob_out)
{
    LBMPM_REQUIRE(c && blob_out, "lbmpm_rk3d_ipc_init: null argument");
    int rc = tx_shape(c);
    slabtx::IpcBlob b;
    if (rc == LBMPM_OK) rc = c->tx.ipc_alloc(&b);
    if (rc != LBMPM_OK) { c->tx.disconnect(); return rc; }
    memset(blob_out, 0, LBMPM_IPC_BLOB_BYTES);
    memcpy(blob_out, &b, sizeof b);
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_ipc_connect(lbmpm_rk3d *c, const void *blob_below, const void *blob_above)
{
    LBMPM_REQUIRE(c && c->tx.kind == LBMPM_TRANSPORT_IPC && !c->tx.connected, "lbmpm_rk3d_ipc_connect: call lbmpm_rk3d_ipc_init first (once)");
    LBMPM_REQUIRE((blob_below != nullptr) == c->tx.has_below && (blob_above != nullptr) == c->tx.has_above,
                  "lbmpm_rk3d_ipc_connect: a blob for exactly the neighbours this slab has (below: %d, above: %d)", (int)c->tx.has_below, (int)c->tx.has_above);
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    slabtx::IpcBlob b;
    int rc = LBMPM_OK;
    if (blob_below) { memcpy(&b, blob_below, sizeof b); rc = c->tx.ipc_open(0, &b, c->tx.bytes_dn); }
    if (rc == LBMPM_OK && blob_above) { memcpy(&b, blob_above, sizeof b); rc = c->tx.ipc_open(1, &b, c->tx.bytes_up); }
    if (rc != LBMPM_OK) return rc;
    c->tx.connected = true;
    c->halo_valid = false;
    return LBMPM_OK;
}

extern "C" int lbmpm_rccl_unique_id(void *id_out, const char *librccl_path)
{
    LBMPM_REQUIRE(id_out, "lbmpm_rccl_unique_id: null argument");
    slabtx::Rccl r;
    int rc = r.open(librccl_path);
    if (rc != LBMPM_OK) return rc;
    slabtx::Rccl::UniqueId id;
    const int e = r.GetUniqueId(&id);
    if (e != 0) { set_error("ncclGetUniqueId: %s", r.GetErrorString(e)); r.close(); return LBMPM_ERR_HIP; }
    memcpy(id_out, &id, LBMPM_RCCL_ID_BYTES);
    // (the handle stays open: unloading librccl here would unload what the id refers to in some builds)
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_rccl_connect(lbmpm_rk3d *c, const void *id, int rank, int nranks, const char *librccl_path)
{
    LBMPM_REQUIRE(c && id && nranks >= 1 && rank >= 0 && rank < nranks, "lbmpm_rk3d_rccl_connect: bad argument");
    int rc = tx_shape(c);
    if (rc != LBMPM_OK) { c->tx.disconnect(); return rc; }
    if (!((c->tx.has_below ? rank > 0 : true) && (c->tx.has_above ? rank + 1 < nranks : true))) {
        set_error("lbmpm_rk3d_rccl_connect: rank %d of %d has no rank %s it, but the slab has a neighbour there", rank, nranks, c->tx.has_below && rank == 0 ? "below" : "above");
        c->tx.disconnect();           // (the landing area of tx_shape: every error path behind it gives it back)
        return LBMPM_ERR_INVALID;
    }
    rc = c->tx.rccl.open(librccl_path);
    if (rc != LBMPM_OK) { c->tx.disconnect(); return rc; }
    slabtx::Rccl::UniqueId uid;
    memcpy(&uid, id, sizeof uid);
    if (hipSetDevice(c->cfg.device) != hipSuccess) { set_error("hipSetDevice(%d) failed", c->cfg.device); c->tx.disconnect(); return LBMPM_ERR_HIP; }
    const int e = c->tx.rccl.CommInitRank(&c->tx.comm, nranks, uid, rank);
    if (e != 0) { set_error("ncclCommInitRank(rank %d of %d): %s", rank, nranks, c->tx.rccl.GetErrorString(e)); c->tx.disconnect(); return LBMPM_ERR_HIP; }
    c->tx.rank = rank; c->tx.nranks = nranks; c->tx.peer_up = rank + 1; c->tx.peer_dn = rank - 1;
    c->tx.kind = LBMPM_TRANSPORT_RCCL; c->tx.connected = true; c->tx.seq = 0;
    c->halo_valid = false;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_transport_disconnect(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    c->tx.disconnect();
    c->halo_valid = false;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_transport_kind(lbmpm_rk3d *c, int *value_ops)
{
    if (!c) return LBMPM_TRANSPORT_NONE;
    if (value_ops) *value_ops = c->tx.kind == LBMPM_TRANSPORT_IPC && c->tx.value_ops ? 1 : 0;
    return c->tx.connected ? c->tx.kind : LBMPM_TRANSPORT_NONE;
}

static int release_ipc_waits(lbmpm_rk3d *c);
extern "C" int lbmpm_rk3d_ipc_release_waits(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c && c->tx.kind == LBMPM_TRANSPORT_IPC && c->tx.flags, "lbmpm_rk3d_ipc_release_waits: no IPC transport");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    return release_ipc_waits(c);
}

namespace { __global__ void tx_fill(double *p, size_t n, double v) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v + (double)i; } }
namespace { __global__ void tx_check(const double *p, size_t n, double v, unsigned long long *bad)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && p[i] != v + (double)i) atomicAdd(bad, 1ull);
} }

// Probe of the CONNECTED transport between the real neighbours, enqueued on the context's stream (the caller polls the stream under a
// deadline, then reads the verdict): `rounds` patterned messages each way -- both slot parities several times over, so that a reader
// that served a landing slot from a stale cache line would be caught -- written into the (still unused) send buffers, exchanged,
// and compared on the receiving side by a kernel launched behind the transport's waits.  Call before set_density.
extern "C" int lbmpm_rk3d_transport_probe(lbmpm_rk3d *c, int rounds)
{
    LBMPM_REQUIRE(c && c->tx.connected && rounds >= 1, "lbmpm_rk3d_transport_probe: no transport connected");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->probe_bad) LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->probe_bad), sizeof(unsigned long long)));
    LBMPM_HIP_TRY(hipMemsetAsync(c->probe_bad, 0, sizeof(unsigned long long), c->stream));
    const size_t nu = c->tx.bytes_up / 8, nd = c->tx.bytes_dn / 8, nb = c->tx.bytes_from_below / 8, na = c->tx.bytes_from_above / 8;
    auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    for (int r = 1; r <= rounds; ++r) {
        const double up = 1000. * r + 1., dn = -(1000. * r + 1.);
        if (c->tx.has_above) tx_fill<<<grid(nu), dim3(256), 0, c->stream>>>(c->send_up, nu, up);
        if (c->tx.has_below) tx_fill<<<grid(nd), dim3(256), 0, c->stream>>>(c->send_dn, nd, dn);
        const double *fb = nullptr, *fa = nullptr;
        const int rc = c->tx.exchange(c->stream, c->send_up, c->send_dn, &fb, &fa);
        if (rc != LBMPM_OK) return rc;
        if (c->tx.has_below) tx_check<<<grid(nb), dim3(256), 0, c->stream>>>(fb, nb, up, c->probe_bad);       // what the rank below sent up
        if (c->tx.has_above) tx_check<<<grid(na), dim3(256), 0, c->stream>>>(fa, na, dn, c->probe_bad);       // what the rank above sent down
        LBMPM_HIP_TRY(hipGetLastError());
    }
    c->halo_valid = false;
    return LBMPM_OK;
}

// verdict of the last probe: doubles that arrived different from what the neighbour sent (synchronises the context's stream)
extern "C" int lbmpm_rk3d_transport_probe_result(lbmpm_rk3d *c, int64_t *mismatches)
{
    LBMPM_REQUIRE(c && mismatches && c->probe_bad, "lbmpm_rk3d_transport_probe_result: no probe was run");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    unsigned long long v = 0;
    LBMPM_HIP_TRY(hipMemcpyAsync(&v, c->probe_bad, sizeof v, hipMemcpyDeviceToHost, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    *mismatches = (int64_t)v;
    return LBMPM_OK;
}

extern "C" int lbmpm_transport_selftest(int kind, int device, int64_t bytes, const char *librccl_path)
{
    LBMPM_REQUIRE((kind == LBMPM_TRANSPORT_IPC || kind == LBMPM_TRANSPORT_RCCL) && bytes >= 8 && bytes % 8 == 0, "lbmpm_transport_selftest: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(device));
    slabtx::Transport t;
    const size_t n = (size_t)bytes / 8;
    double *up = nullptr, *dn = nullptr;
    hipStream_t st = nullptr;
    int rc = t.set_shape(device, true, true, (size_t)bytes, (size_t)bytes, (size_t)bytes, (size_t)bytes);
    auto done = [&](int code) { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } t.disconnect(); (void)hipFree(up); (void)hipFree(dn); return code; };
    if (rc != LBMPM_OK) return done(rc);
    if (hipMalloc(reinterpret_cast<void **>(&up), (size_t)bytes) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&dn), (size_t)bytes) != hipSuccess ||
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_error("lbmpm_transport_selftest: allocation failed"); return done(LBMPM_ERR_NOMEM); }
    if (kind == LBMPM_TRANSPORT_IPC) {
        slabtx::IpcBlob b;
        rc = t.ipc_alloc(&b);
        if (rc == LBMPM_OK) rc = t.ipc_open(0, &b, (size_t)bytes);       // both neighbours are this very landing area (same process: by pointer)
        if (rc == LBMPM_OK) rc = t.ipc_open(1, &b, (size_t)bytes);
        if (rc != LBMPM_OK) return done(rc);
        t.connected = true;
    } else {
        rc = t.rccl.open(librccl_path);
        if (rc != LBMPM_OK) return done(rc);
        slabtx::Rccl::UniqueId id;
        int e = t.rccl.GetUniqueId(&id);
        if (e == 0) e = t.rccl.CommInitRank(&t.comm, 1, id, 0);
        if (e != 0) { set_error("RCCL self-test: %s", t.rccl.GetErrorString(e)); return done(LBMPM_ERR_HIP); }
        t.kind = LBMPM_TRANSPORT_RCCL; t.rank = 0; t.nranks = 1; t.peer_up = 0; t.peer_dn = 0; t.connected = true;
    }
    std::vector<double> got(n);
    for (int round = 1; round <= 3; ++round) {          // three messages: both parities of the IPC slots, and the first one again
        tx_fill<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(up, n, 1000. * round);
        tx_fill<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(dn, n, -1000. * round);
        const double *fb = nullptr, *fa = nullptr;
        rc = t.exchange(st, up, dn, &fb, &fa);
        if (rc != LBMPM_OK) return done(rc);
        // IPC: "up" lands in the upper neighbour's from-below slot = our own; RCCL to oneself: sends and receives match in order,
        // the first receive posted is the from-above slot
        const double *exp_up = kind == LBMPM_TRANSPORT_IPC ? fb : fa, *exp_dn = kind == LBMPM_TRANSPORT_IPC ? fa : fb;
        for (int w = 0; w < 2; ++w) {
            if (hipMemcpyAsync(got.data(), w == 0 ? exp_up : exp_dn, (size_t)bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                set_error("lbmpm_transport_selftest: reading the landed message failed: %s", hipGetErrorString(hipGetLastError())); return done(LBMPM_ERR_HIP);
            }
            const double v = (w == 0 ? 1000. : -1000.) * round;
            for (size_t i = 0; i < n; ++i)
                if (got[i] != v + (double)i) { set_error("lbmpm_transport_selftest: message %d (%s) differs at double %zu", round, w == 0 ? "up" : "down", i); return done(LBMPM_ERR_STATE); }
        }
    }
    return done(LBMPM_OK);
}

// the face message of the state (f, pur) over the connected transport, enqueued on the context's stream: pack is the caller's
static int tx_exchange_unpack(lbmpm_rk3d *c, const RK3Dev &q, double *f, uint32_t *pur, bool unpack)
{
    const double *fb = nullptr, *fa = nullptr;
    const int rc = c->tx.exchange(c->stream, c->send_up, c->send_dn, &fb, &fa);
    if (rc != LBMPM_OK) return rc;
    if (unpack) {
        const dim3 grid(c->nseg, (c->ny + BY3 - 1) / BY3, 2), block(BX3, BY3);
        rk3dq_face_unpack<<<grid, block, 0, c->stream>>>(q, f, pur, fb, fa, c->tx.has_below, c->tx.has_above);
        rk3dq_halo_phi<<<grid, block, 0, c->stream>>>(q, fb, fa, c->tx.has_below, c->tx.has_above);
        LBMPM_HIP_TRY(hipGetLastError());
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_halo_exchange(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c && c->tx.connected, "lbmpm_rk3d_halo_exchange: no transport connected (lbmpm_rk3d_ipc_connect / lbmpm_rk3d_rccl_connect)");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->tx.has_below && !c->tx.has_above) return LBMPM_OK;
    int rc = lbmpm_rk3d_pack_halo(c);
    if (rc == LBMPM_OK) rc = tx_exchange_unpack(c, make_dev(c), c->fA, c->purA, true);
    if (rc == LBMPM_OK) c->halo_valid = true;
    return rc;
}

// The whole time step of a slab behind ONE call: interior planes on the second stream, then on the context's
// stream pack -> exchange(populations) -> unpack -> phase field of the face planes -> exchange(phase field) ->
// boundary planes -> join.  `exchange(user, what)` (what = 0 populations, 1 phase field) is the caller's transport:
// it must enqueue the transfer of the LBMPM_RK3D_BUF_* send buffers into the neighbours' receive buffers on the
// context's stream (torch.distributed P2P under torch.cuda.stream(...) does exactly that; so would ncclSend /
// ncclRecv) and return 0.  The library never blocks the host here.
extern "C" int lbmpm_rk3d_step_slab(lbmpm_rk3d *c, int64_t nsteps, int has_below, int has_above, lbmpm_rk3d_exchange_fn exchange,
                                    void *user, int timed)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk3d_step_slab: bad argument");
    const bool nb = has_below || has_above;
    const bool own_tx = nb && !exchange && c->tx.connected;
    LBMPM_REQUIRE(!nb || exchange || own_tx, "lbmpm_rk3d_step_slab: a slab with neighbours needs an exchange callback or a connected transport");
    LBMPM_REQUIRE(!own_tx || c->q23, "the in-library transports serve the compact 23-value storage");
    LBMPM_REQUIRE((has_below != 0) == (c->cfg.z_offset > 0) && (has_above != 0) == (c->cfg.z_offset + c->cfg.nz_local < c->cfg.nz_global),
                  "lbmpm_rk3d_step_slab: has_below / has_above contradict the slab's position in the lattice");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const int64_t tsteps = timed ? (nsteps < 256 ? nsteps : 256) : 0;
    if (timed) {
        if (c->slab_pool.reserve((size_t)(4 * tsteps)) != LBMPM_OK) { set_error("hipEventCreate failed"); return LBMPM_ERR_HIP; }
        c->slab_pool.reset();
        c->slab_timed_steps = tsteps;
    }
    if (c->q23 && nb) {
        // q23 storage: ONE exchange per step, software-pipelined.  The halo planes of the current state are valid on entry (primed below
        // after set_density).  Step: boundary planes first (context's stream) -> face message of the NEW state packed, exchanged
        // and unpacked (same stream) while the interior planes run on the second stream -> join.  The next step's boundary planes find
        // their halo ready; the transfer hides behind the interior planes of THIS step.
        LBMPM_REQUIRE(!c->interior_pending, "lbmpm_rk3d_step_slab: finish the step begun with lbmpm_rk3d_collide_interior first");
        auto fail = [&](int code) {         // leave the context consistent: nothing pending on the second stream
            if (c->aux) (void)hipStreamSynchronize(c->aux);
            (void)hipStreamSynchronize(c->stream);
            c->halo_valid = false;
            return code;
        };
        // every error exit of this block goes through fail(): the lattice launches live on the second stream (advisor, round 5)
#define SLAB_HIP_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return fail(LBMPM_ERR_HIP); } } while (0)
        if (!c->beat_host) {
            SLAB_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->beat_host), 64, hipHostMallocMapped));
            *c->beat_host = 0ull;
            SLAB_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->beat_dev), c->beat_host, 0));
        }
        if (!c->halo_valid) {
            int rc;
            if (own_tx) rc = lbmpm_rk3d_halo_exchange(c);
            else {
                rc = lbmpm_rk3d_pack_halo(c);
                if (rc == LBMPM_OK && exchange(user, 0) != 0) { set_error("lbmpm_rk3d_step_slab: the exchange callback failed (priming the halo planes)"); rc = LBMPM_ERR_STATE; }
                if (rc == LBMPM_OK) rc = lbmpm_rk3d_unpack_halo(c, has_below, has_above);
            }
            if (rc != LBMPM_OK) return fail(rc);
        }
        if (!c->aux) {
#ifdef LBMPM_DEV
            // rk3dq_fused holds a CU's whole register file and most of its LDS: while the interior launch runs, the kernels of the exchange
            // (pack, RCCL's send / recv, unpack) find room only where one of its workgroups retires, so on one GPU the pack .. unpack chain
            // ends with the interior launch (tools/slab_rank_cost.py).  LBMPM_RK3D_COMM_CUS = k (default 0 = off) runs the interior on a
            // stream whose CU mask leaves k CUs alone.  Measured at 512^3 / 8 ranks: any mask slows the interior launch from 1.3 to 1.8 ms
            // (k = 8, 16, 32 alike), the chain drops to 0.55 ms only at k = 32 -- a net loss, hence off.
            int reserve = 0, ncu = 0;
            if (const char *e = getenv("LBMPM_RK3D_COMM_CUS")) reserve = atoi(e);
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->cfg.device);
            if (reserve > 0 && ncu >= 64 && reserve < ncu / 2) {
                std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
                // Bit i of the mask is CU (i / 8) of XCD (i % 8) (measured: leaving out every 32nd / 16th bit slowed the interior by 25 % / 100 %,
                // i.e. took 8 / 16 CUs from ONE of the eight XCDs, whose share of the workgroups stays 1/8).  The highest `reserve` bits are
                // reserve / 8 CUs of every XCD: the exchange's workgroups, dealt round-robin to the XCDs like all others, find room in each.
                for (int i = 0; i < ncu - reserve; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
                if (hipExtStreamCreateWithCUMask(&c->aux, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); c->aux = nullptr; }
            }
#endif
            if (!c->aux) SLAB_HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
            SLAB_HIP_TRY(hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming));
            SLAB_HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
        }
        const int cb = c->boundary;
        const bool has_interior = c->nzl >= 2 * cb + 1;
        const dim3 fgrid(c->nseg, (c->ny + BY3 - 1) / BY3, 2), fblock(BX3, BY3);
        // Two schedules.  "lattice in one stream" (default; LBMPM_RK3D_SLAB_SCHEDULE=split for the other): boundary launch -> pack ->
        // interior launch back to back on the SECOND stream, the chain  wait(pack) -> exchange -> unpack -> halo phase field  on the
        // context's stream beside the interior launch; the next step's boundary launch waits for the chain -- which has long finished.
        // No cross-stream event sits between two lattice launches.  The split schedule (rounds 3 - 4: boundary + pack on the context's
        // stream, the interior behind an event on the second, the step ends with the context's stream waiting for it) pays two such
        // waits per step on the critical path (~ 10 - 15 us each) but enqueues the exchange ahead of the interior launch: it stays the
        // choice for the RCCL transport, whose send / recv kernels need CUs that the interior launch would otherwise take first.
#ifdef LBMPM_DEV
        const char *sched = getenv("LBMPM_RK3D_SLAB_SCHEDULE");
#else
        const char *sched = nullptr;
#endif
        const bool one_stream = has_interior && !(sched && !strcmp(sched, "split")) && ((sched && !strcmp(sched, "one")) || !own_tx || c->tx.kind != LBMPM_TRANSPORT_RCCL);
        if (one_stream) {
            SLAB_HIP_TRY(hipEventRecord(c->ev_dep, c->stream));            // everything enqueued so far (the primed halo planes included)
            for (int64_t k = 0; k < nsteps; ++k) {
                hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                if (k < tsteps) for (int i = 0; i < 4; ++i) c->slab_pool.take(&ev[2 * i], &ev[2 * i + 1]);
                const RK3Dev p = make_dev(c);
                if (ev[0]) SLAB_HIP_TRY(hipEventRecord(ev[0], c->aux));     // (before the wait: a chain that ended late counts into this step)
                SLAB_HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_dep, 0));     // the previous step's chain: its halo planes
                if (ev[6]) SLAB_HIP_TRY(hipEventRecord(ev[6], c->aux));
                const int zi0 = has_below ? cb + 1 : 1, zi1 = has_above ? c->nzl - cb : c->nzl;
                if (has_below && has_above) launch_q23(c, p, c->aux, 1, cb, c->nzl - cb + 1, c->nzl);
                else if (has_below) launch_q23(c, p, c->aux, 1, cb, 1, 0);
                else launch_q23(c, p, c->aux, c->nzl - cb + 1, c->nzl, 1, 0);
                if (ev[7]) SLAB_HIP_TRY(hipEventRecord(ev[7], c->aux));
                RK3Dev q = p;                                                 // the state this step writes
                q.fin = c->fB; q.pur_in = c->purB; q.first = 0;
                rk3dq_face_pack<<<fgrid, fblock, 0, c->aux>>>(q, c->send_up, c->send_dn, has_below, has_above);
                SLAB_HIP_TRY(hipEventRecord(c->ev_done, c->aux));           // (here: "the face message is packed")
                if (hipGetLastError() != hipSuccess) { set_error("lbmpm_rk3d_step_slab: kernel launch failed"); return fail(LBMPM_ERR_HIP); }
                // the interior planes, straight behind the pack in the lattice stream
                if (ev[2]) SLAB_HIP_TRY(hipEventRecord(ev[2], c->aux));
                launch_step_range(c, p, c->aux, zi0, zi1);
                if (ev[3]) SLAB_HIP_TRY(hipEventRecord(ev[3], c->aux));
                // the chain.  (Enqueued BEHIND the interior launch: its dispatch is then in the lattice stream's queue when the pack retires, and the
                // interior's workgroups take the CUs first -- with the chain first, a rank's copy / unpack kernels sometimes won that race and the
                // interior started 0.05 ms late (one rank in eight on one GPU).  A copy engine needs no CU; RCCL keeps the split schedule.)
                SLAB_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done, 0));
                if (ev[4]) SLAB_HIP_TRY(hipEventRecord(ev[4], c->stream));
                const double *from_below = c->recv_below, *from_above = c->recv_above;
                if (own_tx) { const int rc = c->tx.exchange(c->stream, c->send_up, c->send_dn, &from_below, &from_above); if (rc != LBMPM_OK) return fail(rc); }
                else if (exchange(user, 0) != 0) { set_error("lbmpm_rk3d_step_slab: the exchange callback failed"); return fail(LBMPM_ERR_STATE); }
                rk3dq_face_unpack<<<fgrid, fblock, 0, c->stream>>>(q, c->fB, c->purB, from_below, from_above, has_below, has_above);
                rk3dq_halo_phi<<<fgrid, fblock, 0, c->stream>>>(q, from_below, from_above, has_below, has_above);
                slabtx::flag_store<<<1, 1, 0, c->stream>>>(c->beat_dev, (unsigned long long)(c->steps + 1));      // "the exchange of this step is through"
                if (ev[5]) SLAB_HIP_TRY(hipEventRecord(ev[5], c->stream));
                SLAB_HIP_TRY(hipEventRecord(c->ev_dep, c->stream));
                if (hipGetLastError() != hipSuccess) { set_error("lbmpm_rk3d_step_slab: kernel launch failed"); return fail(LBMPM_ERR_HIP); }
                finish_step(c);
                c->halo_valid = true;
                if (ev[1]) SLAB_HIP_TRY(hipEventRecord(ev[1], c->aux));
            }
            // join: whoever uses the context's stream next finds the lattice launches done
            SLAB_HIP_TRY(hipEventRecord(c->ev_done, c->aux));
            SLAB_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done, 0));
            return LBMPM_OK;
        }
        for (int64_t k = 0; k < nsteps; ++k) {
            hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (k < tsteps) for (int i = 0; i < 4; ++i) c->slab_pool.take(&ev[2 * i], &ev[2 * i + 1]);
            if (ev[0]) SLAB_HIP_TRY(hipEventRecord(ev[0], c->stream));
            const RK3Dev p = make_dev(c);
            SLAB_HIP_TRY(hipEventRecord(c->ev_dep, c->stream));            // the previous step, its unpack included
            if (ev[6]) SLAB_HIP_TRY(hipEventRecord(ev[6], c->stream));
            // the boundary ranges that feed a face message, in one launch; a face without a neighbour (the lattice's inlet / outlet end)
            // has no message to hurry for: its planes march with the interior (one range, one prologue and two fill steps less on the end ranks)
            const int zi0 = has_below ? cb + 1 : 1, zi1 = has_above ? c->nzl - cb : c->nzl;
            if (has_interior) {
                if (has_below && has_above) launch_q23(c, p, c->stream, 1, cb, c->nzl - cb + 1, c->nzl);
                else if (has_below) launch_q23(c, p, c->stream, 1, cb, 1, 0);
                else launch_q23(c, p, c->stream, c->nzl - cb + 1, c->nzl, 1, 0);
            }
            else launch_step_range(c, p, c->stream, 1, c->nzl);
            if (ev[7]) SLAB_HIP_TRY(hipEventRecord(ev[7], c->stream));
            if (hipGetLastError() != hipSuccess) { set_error("lbmpm_rk3d_step_slab: kernel launch failed"); return fail(LBMPM_ERR_HIP); }
            if (ev[4]) SLAB_HIP_TRY(hipEventRecord(ev[4], c->stream));
            RK3Dev q = p;                                                     // the state this step writes
            q.fin = c->fB; q.pur_in = c->purB; q.first = 0;
#ifdef LBMPM_DEV
            const int skip = c->dbg;        // timing knock-outs (LBMPM_RK3D_DBG: 1 no pack, 2 no exchange, 4 no unpack / halo phase field)
#else
            constexpr int skip = 0;         // (the knock-outs exist in development builds only)
#endif
            if (!(skip & 1)) rk3dq_face_pack<<<fgrid, fblock, 0, c->stream>>>(q, c->send_up, c->send_dn, has_below, has_above);
            if (has_interior) {
                // Order (per-workgroup time stamps, 512^3 on 8 ranks, tools/dev/k3trace_slab.py): rk3dq_fused fills every CU, so whatever
                // is launched beside a running interior launch trickles through where its workgroups retire and slows them (7.5 instead
                // of 6.3 us per march step, second-round workgroups start late).  Hence: boundary launch -> pack -> THEN the interior
                // launch, enqueued at the same moment as the exchange: the transport's few workgroups (RCCL send / recv, or copies) are
                // placed first on an empty GPU, the interior's workgroups take the rest, the transfer runs beside them.
                SLAB_HIP_TRY(hipEventRecord(c->ev_dep, c->stream));
                SLAB_HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_dep, 0));
            }
            const double *from_below = c->recv_below, *from_above = c->recv_above;
            if (own_tx) {          // copies / ncclSend + ncclRecv and the waits for the neighbours' messages, enqueued here
                if (!(skip & 2)) { const int rc = c->tx.exchange(c->stream, c->send_up, c->send_dn, &from_below, &from_above); if (rc != LBMPM_OK) return fail(rc); }
            }
            else if (!(skip & 2) && exchange(user, 0) != 0) { set_error("lbmpm_rk3d_step_slab: the exchange callback failed"); return fail(LBMPM_ERR_STATE); }
            if (has_interior) {
                hipStream_t ist = (skip & 8) ? c->stream : c->aux;       // (knock-out 8: the interior on the context's own stream)
                if (ev[2]) SLAB_HIP_TRY(hipEventRecord(ev[2], ist));
                launch_step_range(c, p, ist, zi0, zi1);
                if (ev[3]) SLAB_HIP_TRY(hipEventRecord(ev[3], ist));
                SLAB_HIP_TRY(hipEventRecord(c->ev_done, ist));
            }
            if (!(skip & 4)) {
                rk3dq_face_unpack<<<fgrid, fblock, 0, c->stream>>>(q, c->fB, c->purB, from_below, from_above, has_below, has_above);
                rk3dq_halo_phi<<<fgrid, fblock, 0, c->stream>>>(q, from_below, from_above, has_below, has_above);
                slabtx::flag_store<<<1, 1, 0, c->stream>>>(c->beat_dev, (unsigned long long)(c->steps + 1));
            }
            if (hipGetLastError() != hipSuccess) { set_error("lbmpm_rk3d_step_slab: face kernel launch failed"); return fail(LBMPM_ERR_HIP); }
            if (ev[5]) SLAB_HIP_TRY(hipEventRecord(ev[5], c->stream));
            if (has_interior) SLAB_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done, 0));
            finish_step(c);
            c->halo_valid = !(skip & 6);       // the face message of the state just written has been exchanged and unpacked above
            if (ev[1]) SLAB_HIP_TRY(hipEventRecord(ev[1], c->stream));
        }
        return LBMPM_OK;
    }
#undef SLAB_HIP_TRY
    for (int64_t k = 0; k < nsteps; ++k) {
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (k < tsteps) for (int i = 0; i < 4; ++i) c->slab_pool.take(&ev[2 * i], &ev[2 * i + 1]);
        if (ev[0]) LBMPM_HIP_TRY(hipEventRecord(ev[0], c->stream));
        int rc = LBMPM_OK;
        if (nb) rc = collide_interior_ev(c, ev[2], ev[3]);
        if (rc != LBMPM_OK) return rc;
        if (ev[4]) LBMPM_HIP_TRY(hipEventRecord(ev[4], c->stream));
        // a failed exchange must not leave the interior launch pending: the context would refuse every later call
        auto abandon = [&](int code) {
            if (c->interior_pending) { (void)hipStreamSynchronize(c->aux); c->interior_pending = false; }
            return code;        // fA / fB are half-updated: the caller has to set_density (or restart) before stepping again
        };
        if (nb && c->streamed) {
            rc = lbmpm_rk3d_pack_halo(c);
            if (rc == LBMPM_OK && exchange(user, 0) != 0) { set_error("lbmpm_rk3d_step_slab: the exchange callback failed (populations)"); rc = LBMPM_ERR_STATE; }
            if (rc == LBMPM_OK) rc = lbmpm_rk3d_unpack_halo(c, has_below, has_above);
            if (rc != LBMPM_OK) return abandon(rc);
        }
        rc = lbmpm_rk3d_phase_field(c, 0);
        if (rc != LBMPM_OK) return abandon(rc);
        if (nb && exchange(user, 1) != 0) { set_error("lbmpm_rk3d_step_slab: the exchange callback failed (phase field)"); return abandon(LBMPM_ERR_STATE); }
        if (ev[5]) LBMPM_HIP_TRY(hipEventRecord(ev[5], c->stream));
        rc = collide_boundary_ev(c, ev[6], ev[7]);
        if (rc != LBMPM_OK) return rc;
        if (ev[1]) LBMPM_HIP_TRY(hipEventRecord(ev[1], c->stream));
    }
    return LBMPM_OK;
}

// averages [ms] over the timed steps of the last lbmpm_rk3d_step_slab(..., timed = 1): out[0] whole step,
// out[1] interior planes (second stream), out[2] pack .. phase-field exchange on the context's stream (transfers
// included), out[3] boundary planes; out[4] = number of steps averaged.  Synchronises the context's streams.
extern "C" int lbmpm_rk3d_slab_timing(lbmpm_rk3d *c, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk3d_slab_timing: null argument");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->aux) LBMPM_HIP_TRY(hipStreamSynchronize(c->aux));
    for (int i = 0; i < 5; ++i) out[i] = 0.;
    const int64_t n = c->slab_timed_steps;
    if (n <= 0) return LBMPM_OK;
    const bool has_interior = c->aux && !(c->variant == 1 || c->nzl < 2 * c->boundary + 1) &&
                              (c->cfg.z_offset > 0 || c->cfg.z_offset + c->cfg.nz_local < c->cfg.nz_global);
    for (int64_t k = 0; k < n; ++k)
        for (int i = 0; i < 4; ++i) {
            if (i == 1 && !has_interior) continue;
            float ms = 0.f;
            LBMPM_HIP_TRY(hipEventElapsedTime(&ms, c->slab_pool.ev[(size_t)(8 * k + 2 * i)], c->slab_pool.ev[(size_t)(8 * k + 2 * i + 1)]));
            out[i] += ms;
        }
    for (int i = 0; i < 4; ++i) out[i] /= (double)n;
    out[4] = (double)n;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_step(lbmpm_rk3d *c, int64_t nsteps)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk3d_step: bad argument");
    LBMPM_REQUIRE(c->cfg.z_offset == 0 && c->cfg.nz_local == c->cfg.nz_global,
                  "lbmpm_rk3d_step is the single-slab convenience; slabs drive the phases and exchange halos themselves");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    for (int64_t k = 0; k < nsteps; ++k) {
        int rc = lbmpm_rk3d_phase_field(c, 0);
        if (rc == LBMPM_OK) rc = lbmpm_rk3d_collide(c);
        if (rc != LBMPM_OK) return rc;
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_step_timed(lbmpm_rk3d *c, int64_t nsteps, double *ms_total, double *ms_dominant)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk3d_step_timed: bad argument");
    LBMPM_REQUIRE(c->cfg.z_offset == 0 && c->cfg.nz_local == c->cfg.nz_global, "single-slab only");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t pairs = (size_t)(nsteps < 4096 ? nsteps : 4096);
    if (c->pool.reserve(pairs + 1) != LBMPM_OK) { set_error("hipEventCreate failed"); return LBMPM_ERR_HIP; }
    c->pool.reset();
    hipEvent_t t0, t1;
    c->pool.take(&t0, &t1);
    LBMPM_HIP_TRY(hipEventRecord(t0, c->stream));
    for (int64_t k = 0; k < nsteps; ++k) {
        int rc = lbmpm_rk3d_phase_field(c, 0);
        if (rc != LBMPM_OK) return rc;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool ev = c->pool.take(&e0, &e1);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
        rc = lbmpm_rk3d_collide(c);
        if (rc != LBMPM_OK) return rc;
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    }
    LBMPM_HIP_TRY(hipEventRecord(t1, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    LBMPM_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    if (ms_total) *ms_total = ms;
    if (ms_dominant) {
        const size_t timed_launches = c->pool.used / 2 - 1;
        double s = 0.0;
        for (size_t k = 2; k + 1 < c->pool.used; k += 2) {
            float m = 0.f;
            LBMPM_HIP_TRY(hipEventElapsedTime(&m, c->pool.ev[k], c->pool.ev[k + 1]));
            s += m;
        }
        *ms_dominant = timed_launches ? s * (double)nsteps / (double)timed_launches : 0.0;
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_sync(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

// The steady-state watchdog (include/lbmpm.h): host-side polling, so that a stream stuck in hipStreamWaitValue64 / flag_wait / an
// ncclRecv on a neighbour that died does not hang this process for good.
static int release_ipc_waits(lbmpm_rk3d *c)
{
    // from a private non-blocking stream: a copy on the legacy null stream would queue behind the very wait it is to release when the
    // context runs on a blocking stream (advisor, round 5)
    static const unsigned long long big[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    if (!c->wd_stream) LBMPM_HIP_TRY(hipStreamCreateWithFlags(&c->wd_stream, hipStreamNonBlocking));
    LBMPM_HIP_TRY(hipMemcpyAsync(c->tx.flags, big, sizeof big, hipMemcpyHostToDevice, c->wd_stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->wd_stream));
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_sync_deadline(lbmpm_rk3d *c, double seconds)
{
    LBMPM_REQUIRE(c && seconds > 0., "lbmpm_rk3d_sync_deadline: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto idle = [&]() -> int {          // 1 idle, 0 busy, -1 error
        hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess && c->aux) e = hipStreamQuery(c->aux);
        if (e == hipSuccess) return 1;
        if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
        set_error("lbmpm_rk3d_sync_deadline: %s", hipGetErrorString(e));
        return -1;
    };
    // The deadline counts from the last PROGRESS, not from the call: every slab step's exchange chain ends by writing its step number
    // into a pinned host word (lbmpm_rk3d_step_slab); while that word moves, the neighbours answer and the queued steps drain, however
    // many there are and however slow a neighbour is (advisor, round 5: an absolute deadline voided healthy long queues).
    volatile unsigned long long *beat = c->beat_host;
    unsigned long long last = beat ? *beat : 0ull;
    unsigned spins = 0;
    for (;;) {
        const int s = idle();
        if (s == 1) return LBMPM_OK;
        if (s < 0) return LBMPM_ERR_HIP;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if (beat && *beat != last) { last = *beat; t0 = t; }
        if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > seconds) break;
        if (++spins > 2000) { struct timespec nap = {0, 200000}; nanosleep(&nap, nullptr); }       // busy for the first moments, then 0.2 ms naps
    }
    const int kind = c->tx.connected ? c->tx.kind : LBMPM_TRANSPORT_NONE;
    if (kind == LBMPM_TRANSPORT_IPC) {
        const int rc = release_ipc_waits(c);       // every wait of this context returns
        if (rc != LBMPM_OK) return rc;
        c->tx.dead = true;
        (void)hipStreamSynchronize(c->stream);
        if (c->aux) (void)hipStreamSynchronize(c->aux);
    } else if (kind == LBMPM_TRANSPORT_RCCL) {
        c->tx.dead = true;
        if (c->tx.comm && c->tx.rccl.CommAbort) { (void)c->tx.rccl.CommAbort(c->tx.comm); c->tx.comm = nullptr; }
        (void)hipStreamSynchronize(c->stream);
        if (c->aux) (void)hipStreamSynchronize(c->aux);
    }
    if (kind != LBMPM_TRANSPORT_NONE) {      // (nothing released: the work is still running, the step protocol's flags stay what they are)
        c->halo_valid = false;
        c->interior_pending = false;
    }
    set_error("lbmpm_rk3d_sync_deadline: the slab's streams were busy and no face exchange completed for %.1f s -- %s (rank with planes %d..%d of %d)", seconds,
              kind == LBMPM_TRANSPORT_IPC ? "a neighbour's face message did not arrive; the waits were released, the lattice state is void" :
              kind == LBMPM_TRANSPORT_RCCL ? "a neighbour did not answer; the communicator was aborted, the lattice state is void" :
                                             "no in-library transport is connected: nothing was released",
              c->cfg.z_offset, c->cfg.z_offset + c->cfg.nz_local - 1, c->cfg.nz_global);
    return LBMPM_ERR_TIMEOUT;
}

extern "C" int lbmpm_rk3d_buffer(lbmpm_rk3d *c, int which, void **ptr, int64_t *bytes)
{
    LBMPM_REQUIRE(c && ptr && bytes, "lbmpm_rk3d_buffer: null argument");
    const int64_t pb = (int64_t)(c->plane2 * sizeof(double));
    // compact storage moves the fluid cells of the plane only (the two sides of a cut hold the same plane)
    auto fb = [&](int zl) { return face_bytes(c, zl); };
    if (c->q23 && which >= LBMPM_RK3D_BUF_PHI_SEND_UP && which <= LBMPM_RK3D_BUF_PHI_RECV_FROM_ABOVE) {
        *ptr = c->phi; *bytes = 0;          // one exchange per step: the phase field of the halo planes travels as class sums
        return LBMPM_OK;
    }
    switch (which) {
        case LBMPM_RK3D_BUF_F_SEND_UP: *ptr = c->send_up; *bytes = fb(c->nzl); return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_SEND_DOWN: *ptr = c->send_dn; *bytes = fb(1); return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_RECV_FROM_BELOW: *ptr = c->recv_below; *bytes = fb(0); return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_RECV_FROM_ABOVE: *ptr = c->recv_above; *bytes = fb(c->nzl + 1); return LBMPM_OK;
        case LBMPM_RK3D_BUF_PHI_SEND_UP: *ptr = c->phi + (size_t)c->nzl * c->plane2; *bytes = pb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_PHI_SEND_DOWN: *ptr = c->phi + c->plane2; *bytes = pb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_PHI_RECV_FROM_BELOW: *ptr = c->phi; *bytes = pb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_PHI_RECV_FROM_ABOVE: *ptr = c->phi + (size_t)(c->nzl + 1) * c->plane2; *bytes = pb; return LBMPM_OK;
        default: break;
    }
    set_error("lbmpm_rk3d_buffer: unknown buffer id %d", which);
    return LBMPM_ERR_INVALID;
}

extern "C" int lbmpm_rk3d_get_field(lbmpm_rk3d *c, int field, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk3d_get_field: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const double *src = nullptr;
    // the fused kernels keep phi in LDS and write no diagnostics: every field is that of the last
    // lbmpm_rk3d_phase_field(ctx, 1); refuse to hand out one that a later step has outdated
    if (c->observed_at != c->steps && field >= LBMPM_RK3D_PHI && field <= LBMPM_RK3D_VZ) {
        set_error("field %d is stale: call lbmpm_rk3d_phase_field(ctx, 1) after the last step (fields are those of the streamed, "
                  "boundary-corrected lattice at that call; observed at step %lld, now %lld)", field, (long long)c->observed_at, (long long)c->steps);
        return LBMPM_ERR_STATE;
    }
    switch (field) {
        case LBMPM_RK3D_PHI: src = c->phi; break;
        case LBMPM_RK3D_RHO_R: case LBMPM_RK3D_RHO_B: case LBMPM_RK3D_VX: case LBMPM_RK3D_VY: case LBMPM_RK3D_VZ:
            if (!c->diag) { set_error("field %d needs lbmpm_rk3d_phase_field(ctx, 1) first", field); return LBMPM_ERR_STATE; }
            src = c->diag + (size_t)(field - LBMPM_RK3D_RHO_R) * c->vol;
            break;
        default:
            set_error("lbmpm_rk3d_get_field: unknown field id %d", field);
            return LBMPM_ERR_INVALID;
    }
    std::vector<double> h(c->vol);
    LBMPM_HIP_TRY(hipMemcpyAsync(h.data(), src, c->vol * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t hp = (size_t)c->nx * c->ny;
    for (int z = 0; z < c->nzl; ++z)
        for (int y = 0; y < c->ny; ++y)
            for (int x = 0; x < c->nx; ++x) {
                const size_t s = (size_t)z * hp + (size_t)y * c->nx + x;
                out[s] = c->h_domain[s] == 1 ? h[(size_t)(z + 1) * c->plane2 + (size_t)y * c->pitch + x] : 0.0;
            }
    return LBMPM_OK;
}

// What the storage moves, by its own count (bench.py's "bytes moved" beside the algorithmic 608 B): out[0] doubles stored per fluid
// cell, out[1] fluid cells of the owned planes, out[2] those of them in row segments that carry a single-colour flag instead of records
// (q23 storage, rk3dq.h; 0 otherwise), out[3] bytes one time step reads + writes for the owned cells (rim / halo re-reads not counted).
extern "C" int lbmpm_rk3d_storage_info(lbmpm_rk3d *c, int64_t *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk3d_storage_info: null argument");
    out[0] = c->q23 ? QS : 2 * Q; out[1] = c->nfluid; out[2] = 0;
    if (c->q23) {
        LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
        LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<uint32_t> f((size_t)(c->nzl + 2) * c->ny * c->nseg);
        LBMPM_HIP_TRY(hipMemcpy(f.data(), c->purA, f.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const size_t per = (size_t)c->ny * c->nseg;
        for (size_t r = per; r < per * (size_t)(c->nzl + 1); ++r)
            if (f[r] & 3u) out[2] += c->h_rowpop[r];
    }
    out[3] = c->q23 ? 2 * Q * 8 * out[1] + 64 * (out[1] - out[2]) : 2 * 2 * Q * 8 * out[1];
    return LBMPM_OK;
}

// dev tool (LBMPM_RK3D_TRACE=1): the time stamps the workgroups of the LAST rk3dq_fused launch wrote, 4 x n words (100 MHz clock)
extern "C" int lbmpm_rk3d_debug_trace(lbmpm_rk3d *c, unsigned long long *out, int64_t nblocks)
{
#ifndef LBMPM_DEV
    LBMPM_REQUIRE(false, "lbmpm_rk3d_debug_trace: this library was built without -DLBMPM_DEV (per-workgroup time stamps exist in development builds only: tools/dev/devlib.py)");
#endif
    LBMPM_REQUIRE(c && out && c->trace && nblocks * 32 <= (1 << 22), "lbmpm_rk3d_debug_trace: tracing is off (LBMPM_RK3D_TRACE) or too many workgroups");
    LBMPM_HIP_TRY(hipDeviceSynchronize());
    LBMPM_HIP_TRY(hipMemcpy(out, c->trace, (size_t)nblocks * 32, hipMemcpyDeviceToHost));
    return LBMPM_OK;
}

// development aid: one stored component of one plane (0 .. nzl+1, halo planes included) of the current state, dense nx x ny:
// comp 0..18 populations (q23 storage: the colour-blind g_i), 19..22 the record {k_R, A} as stored, 23 the phase-field array,
// 24 the row flags (one value per row segment, out[y * nseg + s])
extern "C" int lbmpm_rk3d_debug_plane(lbmpm_rk3d *c, int comp, int zl, double *out)
{
    LBMPM_REQUIRE(c && out && c->q23 && zl >= 0 && zl <= c->nzl + 1 && comp >= 0 && comp <= 24, "lbmpm_rk3d_debug_plane: q23 storage, plane 0..nzl+1, comp 0..24");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->aux) LBMPM_HIP_TRY(hipStreamSynchronize(c->aux));
    const size_t n = (size_t)c->nx * c->ny;
    if (comp == 24) {
        std::vector<uint32_t> f((size_t)c->ny * c->nseg);
        LBMPM_HIP_TRY(hipMemcpy(f.data(), c->purA + (size_t)zl * c->ny * c->nseg, f.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < f.size(); ++i) out[i] = (double)f[i];
        return LBMPM_OK;
    }
    if (comp == 23) {
        std::vector<double> h(c->plane2);
        LBMPM_HIP_TRY(hipMemcpy(h.data(), c->phi + (size_t)zl * c->plane2, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int y = 0; y < c->ny; ++y) for (int x = 0; x < c->nx; ++x) out[(size_t)y * c->nx + x] = h[(size_t)y * c->pitch + x];
        return LBMPM_OK;
    }
    double *d = nullptr;
    LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(double)));
    const RK3Dev p = make_dev(c);
    rk3dq_debug_plane<<<dim3(c->nseg, (c->ny + BY3 - 1) / BY3), dim3(BX3, BY3), 0, c->stream>>>(p, c->fA, zl, comp, d);
    hipError_t e = hipMemcpyAsync(out, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { set_error("lbmpm_rk3d_debug_plane: %s", hipGetErrorString(e)); return LBMPM_ERR_HIP; }
    return LBMPM_OK;
}

extern "C" int64_t lbmpm_rk3d_num_fluid_nodes(const lbmpm_rk3d *c) { return c ? c->nfluid : 0; }
extern "C" int64_t lbmpm_rk3d_steps_done(const lbmpm_rk3d *c) { return c ? c->steps : 0; }
extern "C" int64_t lbmpm_rk3d_device_bytes(const lbmpm_rk3d *c) { return c ? c->bytes : 0; }
extern "C" const char *lbmpm_rk3d_dominant_kernel(const lbmpm_rk3d *c) { return !c ? "" : (c->variant == 1 ? "rk3d_collide" : (c->q23 ? "rk3dq_fused" : (c->compact ? "rk3dc_fused" : "rk3d_fused"))); }
