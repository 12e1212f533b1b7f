// rk3d_csf.hip -- D3Q19 colour gradient with continuum-surface-force tension ([SurfaceTension] SurfaceTensionType = 'CSF' in 3-D).
//
// "extend a3-a11 to D3Q19 (... 3-D isotropic gradient 3 sum w e phi; CSF kappa = -div n in 3-D)".  The
// reference ships no 3-D source; this is its 2-D CSF loop, RKColorGradientLBM.runRKColorGradient2DCSF (RKCG2D/RKD2Q9.py:1295-1490, kernels
// of RKCG2D/AcceleratedRKGPU2D.py = "A:"), carried to three dimensions operator by operator with z as the flow axis:
//   inlet plane nz-2 / ghost nz-1   A:2348-2412 + A:607-650 (velocity on f_tot, ratioB quirk kept) | A:925-962 + A:968-1002 (pressure per colour)
//   outlet plane 1 / ghost 0        A:2560-2590 + A:1045-1081 (pressure on f_tot) | A:700-784 (convective copies of plane 3 onto 2, 1, 0)
//   f_tot, u with the lagged force, phi   A:1414-1424, A:2634-2654, A:1348-1357
//   phi on the wetting solids       A:1560-1581            gradient   A:1584-1634
//   wetting rule 2 (Akai 2018)      A:2430-2492, solid normals from the 3-D E8 stencil (Sbragaglia et al. 2007) whose sums along one axis
//                                   are the reference's 24-point weights (RKD2Q9.py:811-885)
//   curvature, force                A:2499-2551: K = -(I - n n) : grad n, F = -1/2 sigma K G
//   BGK / MRT + Guo source          A:1804-1848 + A:1743-1798 | A:1938-2017 + A:2027-2113 (D3Q19 basis of d'Humieres et al. 2002)
//   recolouring, streaming          A:1857-1899, A:340-417 (as a pull)
// The parity oracle is oracle/rk3d_csf_oracle.c (pinned by reduction to the capture of the real 2-D driver, tests/test_oracle_rk3d_csf.py);
// the arithmetic below keeps that file's evaluation order (no FMA contraction in this file), so the two agree to rounding.
//
// Schedule of one time step (populations q-major SoA for FLUID cells only in lattice order, two buffers, pull; one thread per fluid
// cell, blocks of 256 cells, fixed grids over lists of blocks):
//   csf3d_tile_count / _rank   which blocks are deep inside one colour (deep_colour), the three lists
//   csf3d_phase      pull + boundary planes -> rho_R, rho_B -> phi                                38 reads, 1 write per cell   } blocks that
//   csf3d_solid_phi  phi of the wetting solids (list in lattice order)                                                          } are not deep
//   csf3d_gradient   G (18 cached phi reads), wetting rule on the cells next to solid, n = -G / |G|   6 writes                   }
//   csf3d_collide    pull + boundary planes again (bit-identical to the first pass), curvature from the neighbours' n, force,   }
//                    collision, recolouring -> the other buffer                                   38 reads, 41 writes           }
//   csf3d_collide_deep   the deep blocks: the present colour alone through a table of source cells 19 reads, 19 writes
// The curvature needs n one cell around and n needs phi one cell around that: two global dependencies per step, hence the two passes
// over the populations where colours meet (136 doubles per cell and step).  Measured: DESIGN.md section 4.
#include "lbmpm_common.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

using lbmpm::set_error;

constexpr int Q = 19;
#define CSF_CX {0, 1, -1, 0, 0, 0, 0, 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0}
#define CSF_CY {0, 0, 0, 1, -1, 0, 0, 1, -1, -1, 1, 0, 0, 0, 0, 1, -1, 1, -1}
#define CSF_CZ {0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 1, -1, -1, 1, 1, -1, -1, 1}
#define CSF_OPP {0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 18, 17}
__device__ __host__ __forceinline__ constexpr double wq(int i) { return i == 0 ? 1. / 3. : (i < 7 ? 1. / 18. : 1. / 36.); }

// meta word of a cell: bit 0 fluid, bits 1 .. 18 "the neighbour in direction i is fluid", bits 20 - 21 kind
// (0 solid, 1 fluid, 2 wetting solid: >= 1 fluid among its 18 neighbours, 3 fluid with >= 1 solid among them)
constexpr unsigned KIND_SHIFT = 20;
constexpr unsigned SRC_WALL = 0xFFFFFFFFu;
enum { L_TODO = 0, L_FULL = 1, L_DEEP = 2 };

struct CsfDev {
    int nx, ny, nz;
    int glo, ghi;                // ghost planes at the low / high end of z (0: the undivided lattice; 2: images of the neighbouring slab's edge planes)
    int zoff, nzg;               // plane z of this lattice is plane z + zoff of the undivided lattice of nzg planes (its own planes: 0 <= z + zoff < nzg)
    unsigned N;                  // cells
    size_t NS;                   // stride between the planes of a dense SoA array (N rounded up to 16)
    size_t FS;                   // stride between the 38 population planes: fluid cells only, numbered in lattice order (x fastest)
    const uint32_t *meta;
    const uint32_t *cidx;        // [N] number of a fluid cell among the fluid cells
    const uint32_t *cells;       // [NF] its inverse: the lattice cell of fluid cell j
    unsigned NF;
    const double *fin;
    double *fout;
    double *phi, *G, *nh, *F, *K, *U;
    const double *ns;
    double sigma, cosT, sinT, beta, delta, tauR, tauB, vzIn, pInB, pInR, pOut;
    int tauType, inletP, conv, wetting, nwet;
    double rate[6];
    double eps;                  // a colour below eps * rho is absent (0x1p-51 unless the caller asks for more)
    // bulk skip (see deep_colour): per block of 256 fluid cells
    unsigned nblk;
    int skip;
    uint8_t *pure;               // [nblk] after a step: 1 every cell of the block holds red only (rho_B == 0 exactly), 2 blue only, 0 else
    uint8_t *deep_prev;          // [nblk] deep_colour of the step before
    const uint8_t *bcblk;        // [nblk] the block holds cells of the open planes or their ghosts: always the full path
    const uint32_t *rng;         // [2][nblk] first / last block that holds a cell within two cells of this block's cells
    const uint32_t *pfx;         // [2][nblk + 1] number of blocks before k whose `pure` is not 1 / not 2
    const uint8_t *deep_now;     // [nblk] deep_colour of this step (csf3d_deep_mark)
    const uint32_t *work;        // [3][nblk + 1] lists of blocks in order, each followed by its length: L_TODO the blocks csf3d_phase / csf3d_gradient
                                 // have something to do in (not deep, or deep since this step), L_FULL the blocks that are not deep, L_DEEP those that are
    const uint32_t *src;         // [18][FS] number of the fluid cell x - e_i (the cell direction i is pulled from), SRC_WALL off a solid
};

struct Nb { unsigned xo[3], yo[3], zo[3]; };
__device__ __forceinline__ Nb make_nb(const CsfDev &p, int x, int y, int z)
{
    Nb n;
    n.xo[0] = (unsigned)(x == 0 ? p.nx - 1 : x - 1); n.xo[1] = (unsigned)x; n.xo[2] = (unsigned)(x == p.nx - 1 ? 0 : x + 1);
    const unsigned nx = (unsigned)p.nx, pl = (unsigned)p.nx * (unsigned)p.ny;
    n.yo[0] = (unsigned)(y == 0 ? p.ny - 1 : y - 1) * nx; n.yo[1] = (unsigned)y * nx; n.yo[2] = (unsigned)(y == p.ny - 1 ? 0 : y + 1) * nx;
    n.zo[0] = (unsigned)(z == 0 ? p.nz - 1 : z - 1) * pl; n.zo[1] = (unsigned)z * pl; n.zo[2] = (unsigned)(z == p.nz - 1 ? 0 : z + 1) * pl;
    return n;
}
__device__ __forceinline__ unsigned at(const Nb &n, int dx, int dy, int dz) { return n.zo[dz + 1] + n.yo[dy + 1] + n.xo[dx + 1]; }

__device__ __forceinline__ double sum19(const double f[Q])
{
    double r = 0.;
#pragma unroll
    for (int i = 0; i < Q; ++i) r += f[i];
    return r;
}
// +-v or nothing for a lattice component (the oracle's `c * v` with c in {-1, 0, 1})
__device__ __forceinline__ void addc(double &acc, int c, double v) { if (c > 0) acc += v; else if (c < 0) acc -= v; }
__device__ __forceinline__ double edotv(int cx, int cy, int cz, double x, double y, double z)
{   // CX * x + CY * y + CZ * z as the oracle's left-to-right sum (products by 0 and 1 are exact)
    double r = cx > 0 ? x : (cx < 0 ? -x : 0.);
    r = cy > 0 ? r + y : (cy < 0 ? r - y : r + 0.);
    r = cz > 0 ? r + z : (cz < 0 ? r - z : r + 0.);
    return r;
}
// A:170-176
__device__ __forceinline__ double feq(double rho, int i, int cx, int cy, int cz, double vx, double vy, double vz)
{
    const double eu = edotv(cx, cy, cz, vx, vy, vz);
    return rho * wq(i) * (1 + (3. * eu + 4.5 * eu * eu - 1.5 * (vx * vx + vy * vy + vz * vz)));
}
__device__ __forceinline__ double sum_inplane(const double f[Q]) { return f[0] + f[1] + f[2] + f[3] + f[4] + f[7] + f[8] + f[9] + f[10]; }

// The populations, densities of one fluid cell as the loop's first half leaves them: pulled (FIRST: taken where they stand -- the state
// given by set_macro / set_pdf has been streamed already), then the boundary-plane rule of the cell's plane.  BC = false: the pull alone
// (what the reference's arrays hold after a step).
template <bool FIRST, bool BC>
__device__ __forceinline__ void cell_state(const CsfDev &p, int x, int y, int z, double fR[Q], double fB[Q], double &rR, double &rB, int only = 0)
{   // only = 1 / 2: every source cell holds red / blue alone (deep_colour): the other colour's populations are exact zeros and are not read
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ, OPP[Q] = CSF_OPP;
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    int zs = z;                                   // the plane whose streamed populations this cell takes
    bool top = false, bot = false;                // the inlet's / the pressure outlet's rule applies
    if (BC && z >= p.glo && z < p.nz - p.ghi) {   // (a slab's own planes; the open planes are planes of the undivided lattice)
        const int zg = z + p.zoff;
        int zsg = zg;
        if (zg == p.nzg - 1) zsg = p.nzg - 2;
        else if (p.conv) { if (zg <= 2) zsg = 3; }
        else if (zg == 0) zsg = 1;
        zs = z + (zsg - zg);
        top = zsg == p.nzg - 2; bot = !p.conv && zsg == 1;
    }
    const Nb nb = make_nb(p, x, y, zs);
    const unsigned own = at(nb, 0, 0, 0);
    const uint32_t m = p.meta[own];
    const unsigned oj = p.cidx[own];
    const double *fr = p.fin, *fb = p.fin + (size_t)Q * p.FS;
    // the numbers of the source cells first (one dependent load each), then the 38 population loads
    unsigned sj[Q];
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const bool from_fluid = !FIRST && ((m >> OPP[i]) & 1u);          // the cell the population comes from, x - e_i
        sj[i] = from_fluid ? p.cidx[at(nb, -CX[i], -CY[i], -CZ[i])] : oj;
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        size_t off = (size_t)i * p.FS + oj;
        if (!(FIRST || i == 0)) {
            const bool from_fluid = (m >> OPP[i]) & 1u;
            off = (size_t)(from_fluid ? i : OPP[i]) * p.FS + sj[i];     // off a solid: half-way bounce-back, the cell's own opposite population
        }
        fR[i] = only == 2 ? 0. : fr[off];
        fB[i] = only == 1 ? 0. : fb[off];
    }
    rR = sum19(fR); rB = sum19(fB);
    if (!BC) return;
    if (top) {
        if (!p.inletP) {                         // A:2348-2412 constantTotalVelocityInlet
            double t[Q];
#pragma unroll
            for (int i = 0; i < Q; ++i) t[i] = fR[i] + fB[i];
            const double v = p.vzIn;
            const double rho = (sum_inplane(t) + 2. * (t[5] + t[11] + t[14] + t[15] + t[18])) / (1. + v);
#pragma unroll
            for (int a = 0; a < 5; ++a)
                t[DN[a]] = feq(rho, DN[a], CX[DN[a]], CY[DN[a]], CZ[DN[a]], 0., 0., v) + (t[UP[a]] - feq(rho, UP[a], CX[UP[a]], CY[UP[a]], CZ[UP[a]], 0., 0., v));
            const double ratioR = rR / (rR + rB);
            rR = ratioR * rho;
#pragma unroll
            for (int a = 0; a < 5; ++a) fR[DN[a]] = ratioR * t[DN[a]];
            const double ratioB = rB / (rR + rB);             // with the new rho_R: the reference's quirk
            rB = ratioB * rho;
#pragma unroll
            for (int a = 0; a < 5; ++a) fB[DN[a]] = ratioB * t[DN[a]];
            if (zs != z) { rR = sum19(fR); rB = sum19(fB); }          // A:607-650: the ghost plane re-sums
        } else {                                 // A:925-962 calConstPressureInletGPU; the ghost plane copies the densities too (A:968-1002)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double *f = c == 0 ? fB : fR;
                const double pr = c == 0 ? p.pInB : p.pInR;
                const double v = -1. + (sum_inplane(f) + 2. * (f[5] + f[11] + f[14] + f[15] + f[18])) / pr;
                const double Nx = 0.5 * ((f[1] + f[7] + f[9]) - (f[2] + f[8] + f[10]));
                const double Ny = 0.5 * ((f[3] + f[7] + f[10]) - (f[4] + f[8] + f[9]));
                f[6] = f[5] - 1. / 3. * pr * v;
                f[12] = f[11] + Nx - 1. / 6. * pr * v;
                f[13] = f[14] - Nx - 1. / 6. * pr * v;
                f[16] = f[15] + Ny - 1. / 6. * pr * v;
                f[17] = f[18] - Ny - 1. / 6. * pr * v;
            }
            rB = p.pInB; rR = p.pInR;
        }
    } else if (bot) {                            // A:2560-2590 calConstPressureLowerGPUTotal; ghost plane 0: A:1045-1081
        double t[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) t[i] = fR[i] + fB[i];
        const double pL = p.pOut;
        const double v = 1. - 1. / pL * (sum_inplane(t) + 2. * (t[6] + t[12] + t[13] + t[16] + t[17]));
        const double Nx = 0.5 * ((t[1] + t[7] + t[9]) - (t[2] + t[8] + t[10]));
        const double Ny = 0.5 * ((t[3] + t[7] + t[10]) - (t[4] + t[8] + t[9]));
        t[5] = t[6] + 1. / 3. * (pL * v);
        t[11] = t[12] - Nx + 1. / 6. * pL * v;
        t[14] = t[13] + Nx + 1. / 6. * pL * v;
        t[15] = t[16] - Ny + 1. / 6. * pL * v;
        t[18] = t[17] + Ny + 1. / 6. * pL * v;
        const double ratioR = rR / (rR + rB), ratioB = rB / (rR + rB);
#pragma unroll
        for (int a = 0; a < 5; ++a) { fR[UP[a]] = ratioR * t[UP[a]]; fB[UP[a]] = ratioB * t[UP[a]]; }
    }
}

__device__ __forceinline__ bool cell_of(const CsfDev &p, unsigned n, int &x, int &y, int &z)
{
    if (n >= p.N) return false;
    const unsigned pl = (unsigned)p.nx * (unsigned)p.ny;
    z = (int)(n / pl);
    const unsigned r = n - (unsigned)z * pl;
    y = (int)(r / (unsigned)p.nx);
    x = (int)(r - (unsigned)y * (unsigned)p.nx);
    return true;
}

// The step's kernels run one thread per FLUID cell (no idle lanes in a porous medium; a wave's 64 cells are 512 consecutive bytes of
// every population plane).  Workgroups go to the XCDs round robin (workgroup b to XCD b % 8); numbered like this, every XCD walks ONE
// contiguous eighth of the fluid cells in lattice order: the cells a workgroup's neighbours in y read are then in the same L2.
__device__ __forceinline__ unsigned block_of()
{
    const unsigned per = (gridDim.x + 7u) / 8u;          // gridDim.x is a multiple of 8
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}
// The bulk of a phase.  A cell whose every source cell held one colour only after the last step (rho of the other colour exactly 0:
// the recolouring then hands on exact zeros) holds that colour only now: phi = +-1 exactly; if that is true two cells around, its
// gradient, normal, curvature and force are exact zeros.  Decided per block of 256 fluid cells: the blocks between the first and the
// last one that holds a cell within two cells of this block's (`rng`, a whole stretch of the lattice order: conservative) all `pure` of one
// colour -- a difference of two prefix counts.  Such a block skips the phase-field pull, the gradient and the curvature: bit-equal to
// the full path (tests/test_rk3d_csf_gpu.py), 79 instead of 136 doubles per cell and step.  The open planes always take the full path.
__device__ __forceinline__ int deep_colour(const CsfDev &p, unsigned b)
{
    if (!p.skip || b >= p.nblk || p.bcblk[b]) return 0;
    const unsigned lo = p.rng[b], hi = p.rng[p.nblk + b];
    const uint32_t *pr = p.pfx, *pb = p.pfx + (p.nblk + 1u);
    if (pr[hi + 1u] - pr[lo] == 0u) return 1;
    if (pb[hi + 1u] - pb[lo] == 0u) return 2;
    return 0;
}
__device__ __forceinline__ bool fluid_cell(const CsfDev &p, unsigned blk, unsigned &j, unsigned &n, int &x, int &y, int &z)
{
    j = blk * 256u + threadIdx.x;
    if (j >= p.NF) return false;
    n = p.cells[j];
    const unsigned pl = (unsigned)p.nx * (unsigned)p.ny;
    z = (int)(n / pl);
    const unsigned r = n - (unsigned)z * pl;
    y = (int)(r / (unsigned)p.nx);
    x = (int)(r - (unsigned)y * (unsigned)p.nx);
    return z >= p.glo && z < p.nz - p.ghi;       // (a slab's ghost planes are images of the neighbour's cells: received, not computed)
}

// csf3d_phase and csf3d_gradient run over the blocks that have something to do: all of them without the bulk skip, else the list
// p.work of the blocks that are not deep or have just become so (a launch of one workgroup per block spends ~ 1 ms at 512^3 on
// workgroups that only find out that they may leave).  A fixed grid; the workgroups of an XCD share one contiguous stretch of the list.
__device__ __forceinline__ const uint32_t *list_of(const CsfDev &p, int which) { return p.work + (size_t)which * (p.nblk + 1u); }
__device__ __forceinline__ void work_range(const CsfDev &p, int which, unsigned &k, unsigned &end, unsigned &step)
{
    const unsigned cnt = p.skip ? list_of(p, which)[p.nblk] : p.nblk, x = blockIdx.x & 7u;
    k = (unsigned)(((unsigned long long)cnt * x) >> 3) + (blockIdx.x >> 3);
    end = (unsigned)(((unsigned long long)cnt * (x + 1u)) >> 3);
    step = gridDim.x >> 3;
}

template <bool FIRST>
__global__ __launch_bounds__(256) void csf3d_phase(CsfDev p)
{
    unsigned k, end, step;
    work_range(p, L_TODO, k, end, step);
    for (; k < end; k += step) {
        const unsigned blk = p.skip ? list_of(p, L_TODO)[k] : k;
        unsigned j, n;
        int x, y, z;
        if (!fluid_cell(p, blk, j, n, x, y, z)) continue;
        const int deep = p.skip ? (int)p.deep_now[blk] : 0;
        if (deep) { p.phi[n] = deep == 1 ? 1. : -1.; continue; }       // (rho - 0) / (rho + 0) = 1 exactly; listed: the block has just become deep
        double fR[Q], fB[Q], rR, rB;
        cell_state<FIRST, true>(p, x, y, z, fR, fB, rR, rB);
        p.phi[n] = (rR - rB) / (rR + rB);
    }
}

// A:1560-1581 calColorValueOnSolid over the list of wetting solids (in lattice order: neighbouring solids read neighbouring phi)
// (a wall cell whose first fluid neighbour's block has been deep for two steps keeps its value: deep means that every fluid neighbour of
// the walls next to that block has phi = the block's colour, this step and the last)
__global__ __launch_bounds__(256) void csf3d_solid_phi(CsfDev p, const uint32_t *wetlist, const uint32_t *wethome)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= (unsigned)p.nwet) return;
    const unsigned n = wetlist[k];
    int x, y, z;
    cell_of(p, n, x, y, z);
    if ((p.glo && z < 1) || (p.ghi && z > p.nz - 2)) return;       // the outer ghost plane's walls: their neighbours are not all here, nobody reads them
    if (p.skip && z >= p.glo && z < p.nz - p.ghi) {                // (the phi message overwrites the ghost planes' walls with the sender's values of the step before)
        const unsigned hb = wethome[k];
        const uint8_t d = p.deep_now[hb];
        if (d != 0 && p.deep_prev[hb] == d) return;
    }
    const Nb nb = make_nb(p, x, y, z);
    const uint32_t m = p.meta[n];
    double sum = 0., sw = 0.;
#pragma unroll
    for (int i = 1; i < Q; ++i)
        if ((m >> i) & 1u) { sum += wq(i) * p.phi[at(nb, CX[i], CY[i], CZ[i])]; sw += wq(i); }
    p.phi[n] = sum / sw;
}

// A:1584-1634 gradient, A:2430-2492 wetting rule, and the unit normal n = -G / |G| (threshold 1e-8, A:2512-2520) the curvature reads
__global__ __launch_bounds__(256) void csf3d_gradient(CsfDev p)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    unsigned k, end, step;
    work_range(p, L_TODO, k, end, step);
    for (; k < end; k += step) {
    const unsigned blk = p.skip ? list_of(p, L_TODO)[k] : k;
    unsigned j, n;
    int x, y, z;
    if (!fluid_cell(p, blk, j, n, x, y, z)) continue;
    if (p.skip && p.deep_now[blk]) {             // phi is the same constant one cell around, phi_s of the walls included: sums of +-w that cancel exactly
        p.G[n] = 0.; p.G[p.NS + n] = 0.; p.G[2 * p.NS + n] = 0.;       // (listed: the block has just become deep)
        p.nh[n] = 0.; p.nh[p.NS + n] = 0.; p.nh[2 * p.NS + n] = 0.;
        continue;
    }
    const uint32_t m = p.meta[n];
    const Nb nb = make_nb(p, x, y, z);
    double gx = 0., gy = 0., gz = 0.;
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const double t = wq(i) * p.phi[at(nb, CX[i], CY[i], CZ[i])];
        addc(gx, CX[i], t); addc(gy, CY[i], t); addc(gz, CZ[i], t);
    }
    gx = 3. * gx; gy = 3. * gy; gz = 3. * gz;
    if (p.nwet > 0 && p.wetting == 2 && ((m >> KIND_SHIFT) & 3u) == 3u) {
        const double nrm = sqrt(gx * gx + gy * gy + gz * gz);
        double ux = 0., uy = 0., uz = 0.;
        if (nrm > 1.0e-8) { ux = -gx / nrm; uy = -gy / nrm; uz = -gz / nrm; }
        const double sx = p.ns[n], sy = p.ns[p.NS + n], sz = p.ns[2 * p.NS + n];
        const double ang = ux * sx + uy * sy + uz * sz;
        const double th = acos(ang);
        double c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
        if (fabs(sin(th)) > 1.0e-9) {
            c1 = p.sinT * cos(th) / sin(th);
            c2 = p.sinT / sin(th);
            c3 = -p.sinT * cos(th) / sin(th);
            c4 = -p.sinT / sin(th);
        }
        const double ax = (p.cosT - c1) * sx + c2 * ux, ay = (p.cosT - c1) * sy + c2 * uy, az = (p.cosT - c1) * sz + c2 * uz;
        const double bx = (p.cosT - c3) * sx + c4 * ux, by = (p.cosT - c3) * sy + c4 * uy, bz = (p.cosT - c3) * sz + c4 * uz;
        const double d1 = sqrt((ax - ux) * (ax - ux) + (ay - uy) * (ay - uy) + (az - uz) * (az - uz));
        const double d2 = sqrt((bx - ux) * (bx - ux) + (by - uy) * (by - uy) + (bz - uz) * (bz - uz));
        if (d1 < d2) { gx = -nrm * ax; gy = -nrm * ay; gz = -nrm * az; }
        else if (d1 > d2) { gx = -nrm * bx; gy = -nrm * by; gz = -nrm * bz; }
    }
    p.G[n] = gx; p.G[p.NS + n] = gy; p.G[2 * p.NS + n] = gz;
    const double gn = sqrt(gx * gx + gy * gy + gz * gz);
    double hx = 0., hy = 0., hz = 0.;
    if (gn > 1.0e-8) { hx = -gx / gn; hy = -gy / gn; hz = -gz / gn; }
    p.nh[n] = hx; p.nh[p.NS + n] = hy; p.nh[2 * p.NS + n] = hz;
    }
}

// rows of the D3Q19 moment basis of d'Humieres et al. 2002:
// rho, e, eps, jx, qx, jy, qy, jz, qz, 3pxx, 3pixx, pww, piww, pxy, pyz, pxz, mx, my, mz  (fold to constants in the unrolled loops)
__device__ __host__ constexpr double mrow(int k, int i)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const double x = CX[i], y = CY[i], z = CZ[i], c2 = x * x + y * y + z * z;
    switch (k) {
    case 0: return 1.;
    case 1: return 19. * c2 - 30.;
    case 2: return (21. * c2 * c2 - 53. * c2 + 24.) / 2.;
    case 3: return x;
    case 4: return (5. * c2 - 9.) * x;
    case 5: return y;
    case 6: return (5. * c2 - 9.) * y;
    case 7: return z;
    case 8: return (5. * c2 - 9.) * z;
    case 9: return 3. * x * x - c2;
    case 10: return (3. * c2 - 5.) * (3. * x * x - c2);
    case 11: return y * y - z * z;
    case 12: return (3. * c2 - 5.) * (y * y - z * z);
    case 13: return x * y;
    case 14: return y * z;
    case 15: return x * z;
    case 16: return (y * y - z * z) * x;
    case 17: return (z * z - x * x) * y;
    default: return (x * x - y * y) * z;
    }
}
__device__ __host__ constexpr double mnorm(int k)
{
    double a = 0.;
    for (int i = 0; i < Q; ++i) a += mrow(k, i) * mrow(k, i);
    return a;
}
template <int K>
__device__ __forceinline__ double moment(const double d[Q])
{
    double acc = 0.;
#pragma unroll
    for (int i = 0; i < Q; ++i)
        if (mrow(K, i) != 0.) acc += mrow(K, i) * d[i];
    return acc;
}
template <int K>
__device__ __forceinline__ void moments_from(const double S[Q], const double d[Q], double m[Q])
{
    constexpr double inv = 1. / mnorm(K);        // (the oracle divides; one rounding apart)
    m[K] = S[K] * moment<K>(d) * inv;
    if constexpr (K + 1 < Q) moments_from<K + 1>(S, d, m);
}
// d <- M^-1 diag(S) M d
__device__ __forceinline__ void mrt_apply(const double S[Q], double d[Q])
{
    double m[Q];
    moments_from<0>(S, d, m);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        double acc = 0.;
#pragma unroll
        for (int k = 0; k < Q; ++k)
            if (mrow(k, i) != 0.) acc += mrow(k, i) * m[k];
        d[i] = acc;
    }
}

// tau(phi), A:1815-1827
__device__ __forceinline__ double tau_of(const CsfDev &p, double Phi, double rR, double rB)
{
    double tau = 1.;
    if (Phi > p.delta) tau = p.tauR;
    else if (Phi < -p.delta) tau = p.tauB;
    else if (fabs(Phi) <= p.delta) {
        if (p.tauType == 1) {
            tau = 0.5 + 1. / ((1. + Phi) / (2. * (p.tauR - 0.5)) + (1. - Phi) / (2. * (p.tauB - 0.5)));
        } else if (p.tauType == 2) {
            const double ratioR = rR / (rR + rB), ratioB = rB / (rR + rB);
            const double miuR = 3. / (p.tauR - 0.5), miuB = 3. / (p.tauB - 0.5);
            const double miu = 1. / (ratioR * miuR + ratioB * miuB);
            tau = 3. * miu + 0.5;
        }
    }
    return tau;
}

// second half of the loop for one cell: curvature and force, collision with the Guo source, recolouring; stores the post-collision
// populations (the next step pulls them).  DIAG: also keep u and K of the step.
#ifndef CSF_MRT_WAVES
#define CSF_MRT_WAVES 2
#endif
template <bool FIRST, bool MRT, bool DIAG>
__global__ __launch_bounds__(256, MRT ? CSF_MRT_WAVES : 3) void csf3d_collide(CsfDev p)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    unsigned kk, kend, kstep;
    work_range(p, L_FULL, kk, kend, kstep);
    for (; kk < kend; kk += kstep) {
    const unsigned blk = p.skip ? list_of(p, L_FULL)[kk] : kk;
    unsigned j, n;
    int x, y, z;
    const bool active = fluid_cell(p, blk, j, n, x, y, z);
    constexpr int deep = 0;                       // (the deep blocks go through csf3d_collide_deep)
    const bool was_deep = p.skip && p.deep_prev[blk] != 0;       // the arrays G, n, F of this block hold zeros
    bool only_red = true, only_blue = true;
    if (active) {
    const uint32_t m = p.meta[n];
    double *fr = p.fout + j, *fb = p.fout + (size_t)Q * p.FS + j;
    double fR[Q], fB[Q], rR, rB;
    cell_state<FIRST, true>(p, x, y, z, fR, fB, rR, rB, deep);
    double t[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) t[i] = fR[i] + fB[i];
    // A:2634-2654: u = (sum e f_tot + F / 2) / rho with the force of the step before
    double mx = 0., my = 0., mz = 0.;
#pragma unroll
    for (int i = 1; i < Q; ++i) { addc(mx, CX[i], t[i]); addc(my, CY[i], t[i]); addc(mz, CZ[i], t[i]); }
    const double rs = rB + rR;
    const double pfx_ = was_deep ? 0. : p.F[n], pfy_ = was_deep ? 0. : p.F[p.NS + n], pfz_ = was_deep ? 0. : p.F[2 * p.NS + n];
    const double vx = (mx + 0.5 * pfx_) / rs, vy = (my + 0.5 * pfy_) / rs, vz = (mz + 0.5 * pfz_) / rs;
    const double phi = (rR - rB) / (rR + rB);
    // A:2499-2551: derivatives of n over the fluid neighbours, K = -(I - n n) : grad n
    double gx = 0., gy = 0., gz = 0., k = 0., fx = 0., fy = 0., fz = 0.;
    if (!deep) {                                 // (deep: G = 0, so n = 0, so K = 0 whatever the neighbours' normals are, so F = 0)
    const Nb nb = make_nb(p, x, y, z);
    gx = p.G[n]; gy = p.G[p.NS + n]; gz = p.G[2 * p.NS + n];
    const double ux = p.nh[n], uy = p.nh[p.NS + n], uz = p.nh[2 * p.NS + n];
    double dxx = 0., dxy = 0., dxz = 0., dyx = 0., dyy = 0., dyz = 0., dzx = 0., dzy = 0., dzz = 0.;     // d<a><b> = d_a n_b
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        if (!((m >> i) & 1u)) continue;
        const unsigned q = at(nb, CX[i], CY[i], CZ[i]);
        const double qx = 3. * wq(i) * p.nh[q], qy = 3. * wq(i) * p.nh[p.NS + q], qz = 3. * wq(i) * p.nh[2 * p.NS + q];
        addc(dxx, CX[i], qx); addc(dxy, CX[i], qy); addc(dxz, CX[i], qz);
        addc(dyx, CY[i], qx); addc(dyy, CY[i], qy); addc(dyz, CY[i], qz);
        addc(dzx, CZ[i], qx); addc(dzy, CZ[i], qy); addc(dzz, CZ[i], qz);
    }
    k = ux * uy * (dyx + dxy) + ux * uz * (dzx + dxz) + uy * uz * (dzy + dyz)
        - (uy * uy + uz * uz) * dxx - (ux * ux + uz * uz) * dyy - (ux * ux + uy * uy) * dzz;
    fx = -0.5 * p.sigma * k * gx; fy = -0.5 * p.sigma * k * gy; fz = -0.5 * p.sigma * k * gz;
    }
    if (!(deep && was_deep)) { p.F[n] = fx; p.F[p.NS + n] = fy; p.F[2 * p.NS + n] = fz; }
    if (DIAG) { p.K[n] = k; p.U[n] = vx; p.U[p.NS + n] = vy; p.U[2 * p.NS + n] = vz; }
    const double tau = tau_of(p, phi, rR, rB);
    if (!MRT) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {                    // A:1804-1848
            const double eT = feq(rR, i, CX[i], CY[i], CZ[i], vx, vy, vz) + feq(rB, i, CX[i], CY[i], CZ[i], vx, vy, vz);
            t[i] = -1. / tau * (t[i] - eT) + t[i];
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) {                    // A:1743-1798
            const double eu = edotv(CX[i], CY[i], CZ[i], vx, vy, vz);
            const double src = wq(i) * ((3. * ((double)CX[i] - vx) + 9. * (double)CX[i] * eu) * fx + (3. * ((double)CY[i] - vy) + 9. * (double)CY[i] * eu) * fy +
                                        (3. * ((double)CZ[i] - vz) + 9. * (double)CZ[i] * eu) * fz) * (1. - 1. / (2. * tau));
            t[i] = t[i] + src;
        }
    } else {
        const double it = 1. / tau;
        double S[Q] = {p.rate[5], p.rate[0], p.rate[1], p.rate[5], p.rate[2], p.rate[5], p.rate[2], p.rate[5], p.rate[2], it, p.rate[3], it, p.rate[3],
                       it, it, it, p.rate[4], p.rate[4], p.rate[4]};
        double d[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) d[i] = t[i] - (feq(rR, i, CX[i], CY[i], CZ[i], vx, vy, vz) + feq(rB, i, CX[i], CY[i], CZ[i], vx, vy, vz));
        mrt_apply(S, d);                                 // A:1938-2017
#pragma unroll
        for (int i = 0; i < Q; ++i) t[i] = -d[i] + t[i];
        const double uf = vx * fx + vy * fy + vz * fz;
#pragma unroll
        for (int i = 0; i < Q; ++i) {                    // A:2027-2113
            const double ef = edotv(CX[i], CY[i], CZ[i], fx, fy, fz), eu = edotv(CX[i], CY[i], CZ[i], vx, vy, vz);
            d[i] = wq(i) * (3. * ef + 9. * eu * ef - 3. * uf);
            S[i] = 1. - 0.5 * S[i];
        }
        mrt_apply(S, d);
#pragma unroll
        for (int i = 0; i < Q; ++i) t[i] = t[i] + d[i];
    }
    // A:1857-1899 calRecoloringProcessM
    const double gn = sqrt(gx * gx + gy * gy + gz * gz), tot = rR + rB;
    // "One colour alone" as a crisp property of a cell (the bulk skip rests on it, as the row flags of rk3dq.h do): a colour whose density
    // is within the rounding of the total (|rho_c| <= 2^-51 rho: absent, or the far end of the other colour's tail, which would otherwise
    // creep outwards one cell per step as ever smaller numbers) is absent -- it hands on exact zeros, the other colour takes f_tot.
    // The oracle keeps the tail; the difference is below 1e-15 of the density fields (tests: 1e-10).
    const double tiny = p.eps * tot;
    only_red = fabs(rB) <= tiny; only_blue = !only_red && fabs(rR) <= tiny;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const double un = i == 0 ? 0. : (i < 7 ? 1. : sqrt(2.));
        double c = 0.;
        if (gn > 1.0e-8 && un > 1.0e-8) c = edotv(CX[i], CY[i], CZ[i], gx, gy, gz) / (un * gn);
        const double a = rR / tot * t[i] + p.beta * rR * rB / tot * wq(i) * c * un, b = rB / tot * t[i] - p.beta * rR * rB / tot * wq(i) * c * un;
        // streaming stores: these lines are not read again before the next step
        __builtin_nontemporal_store(only_red ? t[i] : (only_blue ? 0. : a), fr + (size_t)i * p.FS);
        __builtin_nontemporal_store(only_blue ? t[i] : (only_red ? 0. : b), fb + (size_t)i * p.FS);
    }
    }   // active
    if (p.skip) {                                // what the block hands on, for the next step's deep_colour
        const int allr = __syncthreads_and(only_red), allb = __syncthreads_and(only_blue);
        // (a block that holds ghost cells: what its own cells allow, bit 0 red / bit 1 blue -- both if it has none; the sender's flags for the
        // ghost cells are ANDed in when the face message arrives, csf3d_face_flags)
        if (threadIdx.x == 0) { p.pure[blk] = (p.bcblk[blk] & 2) ? (uint8_t)((allr ? 1 : 0) | (allb ? 2 : 0)) : (uint8_t)(allr ? 1 : (allb ? 2 : 0)); p.deep_prev[blk] = 0; }
    }
    }   // blocks of the list
}

// The deep blocks' collision (deep_colour): one colour alone, no gradient, no force.  What the full path computes then, term by term --
// the absent colour's populations and equilibrium are exact zeros, the Guo source is a sum of products by F = 0, cos(theta_i) = 0,
// rho_c / rho = 1 exactly -- so the present colour's populations are the relaxed f_tot and the other colour's are zeros, bit for bit
// (tests: variant 1 runs every block through csf3d_collide).  19 loads through the table of source cells (no lattice coordinates, no
// meta words: the open planes are never deep), 19 stores; the absent colour is not read, and not written again once its zeros are in
// place (a block deep since the step before: this buffer was written two steps ago, when the block held that colour alone already).
template <bool MRT, bool DIAG>
__global__ __launch_bounds__(256) void csf3d_collide_deep(CsfDev p)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ, OPP[Q] = CSF_OPP;
    unsigned kk, kend, kstep;
    work_range(p, L_DEEP, kk, kend, kstep);
    for (; kk < kend; kk += kstep) {
        const unsigned blk = list_of(p, L_DEEP)[kk];
        const int deep = (int)p.deep_now[blk];
        const bool same_deep = (int)p.deep_prev[blk] == deep, was_deep = p.deep_prev[blk] != 0;
        const unsigned j = blk * 256u + threadIdx.x;
        if (j < p.NF) {
            const size_t colour = deep == 1 ? 0 : (size_t)Q * p.FS;
            const double *f = p.fin + colour;
            double t[Q];
            t[0] = f[j];
#pragma unroll
            for (int i = 1; i < Q; ++i) {
                const unsigned q = p.src[(size_t)(i - 1) * p.FS + j];
                t[i] = f[q != SRC_WALL ? (size_t)i * p.FS + q : (size_t)OPP[i] * p.FS + j];
            }
            const double rc = sum19(t);
            double mx = 0., my = 0., mz = 0.;
#pragma unroll
            for (int i = 1; i < Q; ++i) { addc(mx, CX[i], t[i]); addc(my, CY[i], t[i]); addc(mz, CZ[i], t[i]); }
            const double rR = deep == 1 ? rc : 0., rB = deep == 1 ? 0. : rc;
            const double rs = rB + rR;
            unsigned n = 0;
            if (!was_deep || DIAG) n = p.cells[j];
            const double pfx_ = was_deep ? 0. : p.F[n], pfy_ = was_deep ? 0. : p.F[p.NS + n], pfz_ = was_deep ? 0. : p.F[2 * p.NS + n];
            const double vx = (mx + 0.5 * pfx_) / rs, vy = (my + 0.5 * pfy_) / rs, vz = (mz + 0.5 * pfz_) / rs;
            const double phi = (rR - rB) / (rR + rB);
            if (!was_deep) { p.F[n] = 0.; p.F[p.NS + n] = 0.; p.F[2 * p.NS + n] = 0.; }
            if (DIAG) { p.K[n] = 0.; p.U[n] = vx; p.U[p.NS + n] = vy; p.U[2 * p.NS + n] = vz; }
            const double tau = tau_of(p, phi, rR, rB);
            if (!MRT) {
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const double eT = feq(rc, i, CX[i], CY[i], CZ[i], vx, vy, vz);      // (+ the absent colour's 0 * ... = +0)
                    t[i] = -1. / tau * (t[i] - eT) + t[i];
                }
            } else {
                const double it = 1. / tau;
                const double S[Q] = {p.rate[5], p.rate[0], p.rate[1], p.rate[5], p.rate[2], p.rate[5], p.rate[2], p.rate[5], p.rate[2], it, p.rate[3], it, p.rate[3],
                                     it, it, it, p.rate[4], p.rate[4], p.rate[4]};
                double d[Q];
#pragma unroll
                for (int i = 0; i < Q; ++i) d[i] = t[i] - feq(rc, i, CX[i], CY[i], CZ[i], vx, vy, vz);
                mrt_apply(S, d);
#pragma unroll
                for (int i = 0; i < Q; ++i) t[i] = -d[i] + t[i];
            }
            double *present = p.fout + colour + j, *absent = p.fout + ((size_t)Q * p.FS - colour) + j;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                __builtin_nontemporal_store(t[i], present + (size_t)i * p.FS);
                if (!same_deep) __builtin_nontemporal_store(0., absent + (size_t)i * p.FS);
            }
        }
        __syncthreads();                         // (every wave has read deep_prev)
        if (threadIdx.x == 0) { p.pure[blk] = (uint8_t)deep; p.deep_prev[blk] = (uint8_t)deep; }     // (sums of exact zeros: the absent colour stays absent)
    }
}

// Slabs: with the populations that cross a face travels, for the fluid cells of the sender's two edge planes, what their blocks handed on
// (`pure`: 1 red alone, 2 blue alone, 0 both) -- the receiver's blocks of ghost cells AND it into theirs, so that a block next to a face
// is deep when the cells two planes beyond the face allow it (without it every block within reach of a face would take the full path).
__global__ __launch_bounds__(256) void csf3d_face_flags_pack(const uint8_t *pure, unsigned j0, unsigned count, uint8_t *out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) out[i] = pure[(j0 + i) >> 8];
}
// flags != nullptr: the packed bytes; else the sender's own array (same process), its cells numbered from sj0.  One workgroup per block of
// the receiver that holds some of the `count` cells numbered from j0: one atomic per block.
__global__ __launch_bounds__(256) void csf3d_face_flags(const uint8_t *flags, const uint8_t *src_pure, unsigned sj0, uint8_t *pure, unsigned j0, unsigned count)
{
    const unsigned b = (j0 >> 8) + blockIdx.x, j = b * 256u + threadIdx.x;
    const bool mine = j >= j0 && j < j0 + count;
    unsigned f = 3u;
    if (mine) {
        f = flags ? flags[j - j0] : src_pure[(sj0 + (j - j0)) >> 8];
        if (f > 2u) f = 0u;                      // (the sender's own blocks of ghost cells: a mask, not a colour)
    }
    const int red = __syncthreads_and((f & 1u) != 0u), blue = __syncthreads_and((f & 2u) != 0u);
    if (threadIdx.x == 0) {
        const unsigned m = (red ? 1u : 0u) | (blue ? 2u : 0u), sh = 8u * (b & 3u);   // the byte of block b: AND with m; the other three bytes of the word stay
        atomicAnd(reinterpret_cast<unsigned *>(pure + (b & ~3u)), (m << sh) | ~(0xFFu << sh));
    }
}

// the runs of a face message in one launch (blockIdx.y = the run)
struct RunSet { const double *src[10]; double *dst[10]; unsigned count[10]; };
__global__ __launch_bounds__(256) void csf3d_copy_runs(RunSet r)
{
    const unsigned k = blockIdx.y, n = r.count[k];
    const double *s = r.src[k];
    double *d = r.dst[k];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) d[i] = s[i];
}

// phi / n of the fluid cells of a face's planes (dense arrays [ncomp][stride] indexed by lattice cell) to or from a packed stretch
// [ncomp][count], or straight from one context's arrays into another's (blockIdx.y = the component)
__global__ __launch_bounds__(256) void csf3d_copy_cells(const double *src, const uint32_t *scells, unsigned sj0, size_t sstride,
                                                        double *dst, const uint32_t *dcells, unsigned dj0, size_t dstride, unsigned count)
{
    const unsigned a = blockIdx.y;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const double v = scells ? src[(size_t)a * sstride + scells[sj0 + i]] : src[(size_t)a * count + i];
        if (dcells) dst[(size_t)a * dstride + dcells[dj0 + i]] = v; else dst[(size_t)a * count + i] = v;
    }
}

// set-up: the table of source cells
__global__ __launch_bounds__(256) void csf3d_setup_src(CsfDev p, uint32_t *src)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ, OPP[Q] = CSF_OPP;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= p.NF) return;
    const unsigned n = p.cells[j];
    int x, y, z;
    cell_of(p, n, x, y, z);
    const Nb nb = make_nb(p, x, y, z);
    const uint32_t m = p.meta[n];
#pragma unroll
    for (int i = 1; i < Q; ++i)
        src[(size_t)(i - 1) * p.FS + j] = ((m >> OPP[i]) & 1u) ? p.cidx[at(nb, -CX[i], -CY[i], -CZ[i])] : SRC_WALL;
}

// Per step, before the phase field: four small launches of ceil(nblk / 1024) workgroups.
//   csf3d_tile_count<0>  per tile of 1024 blocks: how many are not purely red / not purely blue
//   csf3d_tile_rank<0>   pfx[0][k] / pfx[1][k] = number of such blocks before k (tile offsets summed on the fly)
//   csf3d_tile_count<1>  deep_now[b] = deep_colour(b); per tile: how many blocks belong to each of the three lists (L_TODO, L_FULL, L_DEEP)
//   csf3d_tile_rank<1>   the lists, each in order and followed by its length
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned *lds)         // sum over the 1024 threads, to every thread
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63u) == 0u) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned t = 0;
    for (int w = 0; w < 16; ++w) t += lds[w];
    __syncthreads();
    return t;
}
__device__ __forceinline__ unsigned block_rank(bool flag, unsigned *lds, unsigned &total)      // number of set flags in lower threads
{
    const unsigned long long b = __ballot(flag);
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 0u) lds[w] = (unsigned)__popcll(b);
    __syncthreads();
    unsigned before = 0; total = 0;
    for (unsigned k = 0; k < 16u; ++k) { if (k < w) before += lds[k]; total += lds[k]; }
    __syncthreads();
    return before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
}
// the three lists' membership of block b (L_TODO, L_FULL, L_DEEP)
__device__ __forceinline__ bool listed(int which, int deep, int prev) { return which == L_TODO ? !(deep != 0 && prev == deep) : (which == L_FULL ? deep == 0 : deep != 0); }
template <int WHAT>
__global__ __launch_bounds__(1024) void csf3d_tile_count(CsfDev p, uint8_t *deep_now, uint32_t *tcnt)
{
    __shared__ unsigned lds[16];
    const unsigned b = blockIdx.x * 1024u + threadIdx.x, ntile = gridDim.x;
    if (WHAT == 0) {
        const unsigned v = b < p.nblk ? p.pure[b] : 1u, w = b < p.nblk ? p.pure[b] : 2u;
        const unsigned a = block_sum(v != 1u, lds), c = block_sum(w != 2u, lds);
        if (threadIdx.x == 0) { tcnt[blockIdx.x] = a; tcnt[ntile + blockIdx.x] = c; }
    } else {
        int d = 0, prev = 0;
        if (b < p.nblk) { d = deep_colour(p, b); deep_now[b] = (uint8_t)d; prev = p.deep_prev[b]; }
        for (int which = 0; which < 3; ++which) {
            const unsigned a = block_sum(b < p.nblk && listed(which, d, prev), lds);
            if (threadIdx.x == 0) tcnt[which * ntile + blockIdx.x] = a;
        }
    }
}
template <int WHAT>
__global__ __launch_bounds__(1024) void csf3d_tile_rank(CsfDev p, const uint8_t *deep_now, const uint32_t *tcnt, uint32_t *out)
{
    __shared__ unsigned lds[16];
    const unsigned b = blockIdx.x * 1024u + threadIdx.x, ntile = gridDim.x;
    for (int arr = 0; arr < (WHAT == 0 ? 2 : 3); ++arr) {
        unsigned mine = 0;
        for (unsigned s = threadIdx.x; s < blockIdx.x; s += 1024u) mine += tcnt[arr * ntile + s];
        const unsigned off = block_sum(mine, lds);
        bool flag;
        if (WHAT == 0) flag = b < p.nblk && p.pure[b] != (arr == 0 ? 1u : 2u);
        else flag = b < p.nblk && listed(arr, deep_now[b], p.deep_prev[b]);
        unsigned total;
        const unsigned r = off + block_rank(flag, lds, total);
        uint32_t *o = out + (size_t)arr * (p.nblk + 1u);
        if (WHAT == 0) {
            if (b < p.nblk) o[b] = r;
        } else {
            if (flag) o[r] = b;
        }
        if (blockIdx.x == ntile - 1u && threadIdx.x == 0) o[p.nblk] = off + total;
    }
}

// set-up of the bulk skip: first / last fluid cell any cell of a block reads its state from (pass 0), then first / last block over the
// blocks of those cells' own ranges (pass 1: covers two cells around); blocks that hold cells of the open planes or their ghosts
__global__ __launch_bounds__(256) void csf3d_setup_ranges(CsfDev p, int pass, uint32_t *lo, uint32_t *hi, const uint32_t *lo0, const uint32_t *hi0, uint8_t *bcblk)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= p.NF) return;
    const unsigned n = p.cells[j], b = j >> 8;
    int x, y, z;
    cell_of(p, n, x, y, z);
    int zs = z;
    const bool ghost = z < p.glo || z >= p.nz - p.ghi;
    bool open = false;
    if (!ghost) {
        const int zg = z + p.zoff;
        int zsg = zg;
        open = zg >= p.nzg - 2 || zg <= 1;
        if (zg == p.nzg - 1) zsg = p.nzg - 2;
        else if (p.conv) { if (zg <= 2) zsg = 3; open = open || zg <= 3; }
        else if (zg == 0) zsg = 1;
        zs = z + (zsg - zg);
    }
    const Nb nb = make_nb(p, x, y, zs);
    const unsigned own = at(nb, 0, 0, 0);
    const uint32_t m = p.meta[own];
    unsigned mn = j, mx = j;
    auto take = [&](unsigned q) {
        if (pass == 0) { mn = min(mn, q); mx = max(mx, q); }
        else { mn = min(mn, lo0[q >> 8]); mx = max(mx, hi0[q >> 8]); }
    };
    take(j);
    take(p.cidx[own]);
    // A slab's ghost cells are read, not computed; what matters is where THEIR phase field comes from (pass 1 of the cells that read it):
    // the cells around them -- all here for the first ghost plane, beyond this lattice for the second one, whose phi (read through the
    // walls of the first) therefore counts as unknown: the whole lattice, never one colour.
    if ((p.glo && z < 1) || (p.ghi && z > p.nz - 2)) { mn = 0u; mx = p.NF - 1u; }
#pragma unroll 1
    for (int i = 1; i < Q; ++i) {
        const unsigned q = at(nb, CX[i], CY[i], CZ[i]);
        if ((m >> i) & 1u) { take(p.cidx[q]); continue; }
        if (ghost) continue;
        // a wall cell next to this one: its phi is the mean over ITS fluid neighbours (up to two cells from here, possibly across the wall)
        const uint32_t ms = p.meta[q];
        if (((ms >> KIND_SHIFT) & 3u) != 2u) continue;
        const unsigned pl = (unsigned)p.nx * (unsigned)p.ny;
        const int qz = (int)(q / pl), qy = (int)((q - (unsigned)qz * pl) / (unsigned)p.nx), qx = (int)(q - (unsigned)qz * pl - (unsigned)qy * (unsigned)p.nx);
        const Nb nq = make_nb(p, qx, qy, qz);
        for (int k = 1; k < Q; ++k)
            if ((ms >> k) & 1u) take(p.cidx[at(nq, CX[k], CY[k], CZ[k])]);
    }
    atomicMin(&lo[b], mn);
    atomicMax(&hi[b], mx);
    if (pass == 0 && open) bcblk[b] = 1;
    if (pass == 1 && ghost) bcblk[b] = 3;        // (a launch of its own after pass 0: every writer stores the same value)
}

// host-layout views of the populations: out_pdf [2][N][19], out_rho [2][N], out_u [3][N] (REC only), out_phi [N] (REC only).
// REC = false: the arrays as a completed step leaves them (streamed, densities re-summed); REC = true: what the next step's first half
// makes of them (boundary planes, velocity with half the force, phase field) -- what the reference records (RKD2Q9.py:1382-1393)
template <bool FIRST, bool REC>
__global__ __launch_bounds__(256) void csf3d_observe(CsfDev p, double *out_pdf, double *out_rho, double *out_u, double *out_phi)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    int x, y, z;
    if (!cell_of(p, n, x, y, z)) return;
    const bool fluid = p.meta[n] & 1u;
    double fR[Q], fB[Q], rR = 0., rB = 0.;
    if (fluid) cell_state<FIRST, REC>(p, x, y, z, fR, fB, rR, rB);
    if (out_pdf) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            out_pdf[(size_t)n * Q + i] = fluid ? fR[i] : 0.;
            out_pdf[((size_t)p.N + n) * Q + i] = fluid ? fB[i] : 0.;
        }
    }
    out_rho[n] = rR; out_rho[(size_t)p.N + n] = rB;
    if (REC) {
        double mx = 0., my = 0., mz = 0.;
        if (fluid) {
#pragma unroll
            for (int i = 1; i < Q; ++i) { const double t = fR[i] + fB[i]; addc(mx, CX[i], t); addc(my, CY[i], t); addc(mz, CZ[i], t); }
            const double rs = rB + rR;
            mx = (mx + 0.5 * p.F[n]) / rs; my = (my + 0.5 * p.F[p.NS + n]) / rs; mz = (mz + 0.5 * p.F[2 * p.NS + n]) / rs;
        }
        out_u[n] = mx; out_u[(size_t)p.N + n] = my; out_u[2 * (size_t)p.N + n] = mz;
        out_phi[n] = fluid ? (rR - rB) / (rR + rB) : 0.;
    }
}

// ------------------------------------------------------------------------------------------------ set-up
__device__ __forceinline__ int wrapn(int v, int n) { v %= n; return v < 0 ? v + n : v; }

// meta words (RKD2Q9.py:657-690, :741-763 on the D3Q19 neighbourhood) and the number of wetting solids
__global__ __launch_bounds__(256) void csf3d_setup_meta(int nx, int ny, int nz, unsigned N, const uint8_t *dom, uint32_t *meta, unsigned *nwet)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    const unsigned pl = (unsigned)nx * (unsigned)ny;
    const int z = (int)(n / pl), y = (int)((n - (unsigned)z * pl) / (unsigned)nx), x = (int)(n - (unsigned)z * pl - (unsigned)y * (unsigned)nx);
    const bool fluid = dom[n] == 1;
    uint32_t m = fluid ? 1u : 0u;
    int other = 0;
    for (int i = 1; i < Q; ++i) {
        const size_t q = ((size_t)wrapn(z + CZ[i], nz) * ny + wrapn(y + CY[i], ny)) * nx + wrapn(x + CX[i], nx);
        const bool f = dom[q] == 1;
        if (f) m |= 1u << i;
        if (f != fluid) ++other;
    }
    const unsigned kind = fluid ? (other ? 3u : 1u) : (other ? 2u : 0u);
    meta[n] = m | (kind << KIND_SHIFT);
    if (kind == 2u) atomicAdd(nwet, 1u);
}
// the wetting solids in lattice order: per workgroup of 256 cells a count, scanned on the host, then every workgroup writes its own stretch
__global__ __launch_bounds__(256) void csf3d_setup_wetcount(unsigned N, const uint32_t *meta, uint32_t *count)
{
    __shared__ unsigned wsum[4];
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    const bool wet = n < N && ((meta[n] >> KIND_SHIFT) & 3u) == 2u;
    const unsigned long long b = __ballot(wet);
    if ((threadIdx.x & 63u) == 0u) wsum[threadIdx.x >> 6] = (unsigned)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void csf3d_setup_wetlist(unsigned N, const uint32_t *meta, const uint32_t *first, uint32_t *wetlist)
{
    __shared__ unsigned wsum[4];
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    const bool wet = n < N && ((meta[n] >> KIND_SHIFT) & 3u) == 2u;
    const unsigned long long b = __ballot(wet);
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 0u) wsum[w] = (unsigned)__popcll(b);
    __syncthreads();
    unsigned base = first[blockIdx.x];
    for (unsigned k = 0; k < w; ++k) base += wsum[k];
    if (wet) wetlist[base + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = n;
}
// the block of a wall cell's first fluid neighbour
__global__ __launch_bounds__(256) void csf3d_setup_wethome(CsfDev p, const uint32_t *wetlist, uint32_t *wethome)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= (unsigned)p.nwet) return;
    const unsigned n = wetlist[k];
    int x, y, z;
    cell_of(p, n, x, y, z);
    const Nb nb = make_nb(p, x, y, z);
    const uint32_t m = p.meta[n];
    unsigned hb = 0;
#pragma unroll 1
    for (int i = Q - 1; i >= 1; --i)
        if ((m >> i) & 1u) hb = p.cidx[at(nb, CX[i], CY[i], CZ[i])] >> 8;
    wethome[k] = hb;
}
__device__ __forceinline__ double e8w(int c2)
{   // the 3-D E8 stencil of Sbragaglia et al. 2007 by |c|^2; its sums along one axis are 4/21, 4/45, 1/60, 2/315, 1/5040 (RKD2Q9.py:811-885)
    switch (c2) {
    case 1: return 4. / 45.;
    case 2: return 1. / 21.;
    case 3: return 2. / 105.;
    case 4: return 5. / 504.;
    case 5: return 1. / 315.;
    case 6: return 1. / 630.;
    case 8: return 1. / 5040.;
    default: return 0.;
    }
}
// RKD2Q9.py:768-892 calVectorNormaltoSolid in three dimensions, at the fluid cells next to solid
__global__ __launch_bounds__(256) void csf3d_setup_normals(int nx, int ny, int nz, unsigned N, size_t NS, const uint8_t *dom, const uint32_t *meta, double *ns)
{
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n >= N) return;
    double ox = 0., oy = 0., oz = 0.;
    if (((meta[n] >> KIND_SHIFT) & 3u) == 3u) {
        const unsigned pl = (unsigned)nx * (unsigned)ny;
        const int z = (int)(n / pl), y = (int)((n - (unsigned)z * pl) / (unsigned)nx), x = (int)(n - (unsigned)z * pl - (unsigned)y * (unsigned)nx);
        double sx = 0., sy = 0., sz = 0.;
        for (int dz = -2; dz <= 2; ++dz)
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const double w = e8w(dx * dx + dy * dy + dz * dz);
                    if (w == 0.) continue;
                    if (dom[((size_t)wrapn(z + dz, nz) * ny + wrapn(y + dy, ny)) * nx + wrapn(x + dx, nx)] == 1) continue;
                    sx += w * dx; sy += w * dy; sz += w * dz;
                }
        const double nrm = sqrt(sx * sx + sy * sy + sz * sz);
        ox = sx / nrm; oy = sy / nrm; oz = sz / nrm;
    }
    ns[n] = ox; ns[NS + n] = oy; ns[2 * NS + n] = oz;
}

// initial populations: f_c,i = rho_c w_i (1 + 3 e.u + 4.5 (e.u)^2 - 1.5 u^2) (RKD2Q9.py:577-601); host arrays dense [N], velocity may be absent
__global__ __launch_bounds__(256) void csf3d_init(CsfDev p, double *f, const double *rho_r, const double *rho_b, const double *vx, const double *vy, const double *vz)
{
    constexpr int CX[Q] = CSF_CX, CY[Q] = CSF_CY, CZ[Q] = CSF_CZ;
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n >= p.N) return;
    if (!(p.meta[n] & 1u)) return;
    const unsigned j = p.cidx[n];
    const double ux = vx ? vx[n] : 0., uy = vy ? vy[n] : 0., uz = vz ? vz[n] : 0.;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        f[(size_t)i * p.FS + j] = feq(rho_r[n], i, CX[i], CY[i], CZ[i], ux, uy, uz);
        f[(size_t)(Q + i) * p.FS + j] = feq(rho_b[n], i, CX[i], CY[i], CZ[i], ux, uy, uz);
    }
}
// populations given in the host layout [2][N][19] -> SoA
__global__ __launch_bounds__(256) void csf3d_import(CsfDev p, double *f, const double *pdf)
{
    const unsigned n = blockIdx.x * 256u + threadIdx.x;
    if (n >= p.N) return;
    if (!(p.meta[n] & 1u)) return;
    const unsigned j = p.cidx[n];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        f[(size_t)i * p.FS + j] = pdf[(size_t)n * Q + i];
        f[(size_t)(Q + i) * p.FS + j] = pdf[((size_t)p.N + n) * Q + i];
    }
}

}  // namespace

struct lbmpm_rk3dcsf {
    lbmpm_rk3dcsf_config cfg;
    int nx = 0, ny = 0, nz = 0;
    size_t N = 0, NS = 0, FS = 0;
    int64_t nfluid = 0, steps = 0, bytes = 0;
    unsigned nwet = 0;
    bool first = true, have_state = false, diag = false, diag_valid = false;
    hipStream_t stream = nullptr, stream2 = nullptr;        // stream2: the deep blocks' collision beside the full path's four launches
    hipEvent_t ev_lists = nullptr, ev_deep = nullptr;
    hipEvent_t ev_stage = nullptr, ev_sent[2] = {nullptr, nullptr};      // slabs: this context's last stage is queued / a face message of its is
    std::vector<uint32_t> pfirst;  // [nz + 1] fluid cells before plane z
    int next_stage = 0;
    int zoff = 0, nzg = 0;         // CsfDev
    uint8_t *dom = nullptr;
    uint32_t *meta = nullptr, *wetlist = nullptr, *wethome = nullptr, *cidx = nullptr, *cells = nullptr, *rng = nullptr, *pfx = nullptr;
    uint8_t *pure = nullptr, *deep_prev = nullptr, *bcblk = nullptr, *deep_now = nullptr;
    uint32_t *work = nullptr, *tcnt = nullptr, *src = nullptr;
    unsigned nblk = 0;
    bool skip = true;
    double *fA = nullptr, *fB = nullptr, *phi = nullptr, *G = nullptr, *nh = nullptr, *F = nullptr, *K = nullptr, *U = nullptr, *ns = nullptr;
    double *obs = nullptr;         // staging of the observe kernel: rho [2][N], u [3][N], phi [N] (the populations [2][N][19] come and go with the call)
    lbmpm::EventPool pool;
    size_t timed_steps = 0;
};

namespace {

template <typename T>
int dev_alloc(lbmpm_rk3dcsf *c, T **ptr, size_t count)
{
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(ptr), count * sizeof(T));
    if (e != hipSuccess) { set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e)); return LBMPM_ERR_NOMEM; }
    c->bytes += (int64_t)(count * sizeof(T));
    return LBMPM_OK;
}

CsfDev make_dev(const lbmpm_rk3dcsf *c)
{
    CsfDev p;
    p.nx = c->nx; p.ny = c->ny; p.nz = c->nz; p.N = (unsigned)c->N; p.NS = c->NS;
    p.glo = c->cfg.ghost_lo; p.ghi = c->cfg.ghost_hi;
    p.zoff = c->zoff; p.nzg = c->nzg;
    p.FS = c->FS; p.meta = c->meta; p.cidx = c->cidx; p.cells = c->cells; p.NF = (unsigned)c->nfluid; p.fin = c->fA; p.fout = c->fB;
    p.phi = c->phi; p.G = c->G; p.nh = c->nh; p.F = c->F; p.K = c->K; p.U = c->U; p.ns = c->ns;
    const double th = c->cfg.contact_angle_deg / 180. * M_PI;
    p.sigma = c->cfg.surface_tension; p.cosT = cos(th); p.sinT = sin(th);
    p.beta = c->cfg.beta; p.delta = c->cfg.delta; p.tauR = c->cfg.tau_r; p.tauB = c->cfg.tau_b;
    p.vzIn = c->cfg.inlet_velocity_z; p.pInB = c->cfg.inlet_rho_b; p.pInR = c->cfg.inlet_rho_r; p.pOut = c->cfg.outlet_rho_total;
    p.tauType = c->cfg.tau_type; p.inletP = c->cfg.inlet_type == LBMPM_INLET_PRESSURE; p.conv = c->cfg.outlet_type == LBMPM_OUTLET_CONVECTIVE;
    p.wetting = c->cfg.wetting_type; p.nwet = (int)c->nwet;
    bool any = false;
    for (int i = 0; i < 6; ++i) any = any || c->cfg.mrt_rates[i] != 0.;
    const double own[6] = {1.19, 1.4, 1.2, 1.4, 1.2, 0.};
    for (int i = 0; i < 6; ++i) p.rate[i] = any ? c->cfg.mrt_rates[i] : own[i];
    p.eps = c->cfg.bulk_epsilon > 0. ? c->cfg.bulk_epsilon : 0x1p-51;
    p.nblk = c->nblk; p.skip = c->skip ? 1 : 0; p.pure = c->pure; p.deep_prev = c->deep_prev; p.bcblk = c->bcblk; p.rng = c->rng; p.pfx = c->pfx; p.deep_now = c->deep_now; p.work = c->work; p.src = c->src;
    return p;
}

unsigned blocks_of(size_t n) { return (unsigned)((n + 255) / 256); }
unsigned blocks8(size_t n) { return (blocks_of(n) + 7u) / 8u * 8u; }      // fluid_cell(): eight XCDs

// stages: bit 0 lists, bulk collision, phase field | bit 1 solid phi, gradient | bit 2 collision of the full path (7: a whole step)
template <bool FIRST>
int launch_step(lbmpm_rk3dcsf *c, const CsfDev &p, hipEvent_t e0, hipEvent_t e1, int stages = 7)
{
    const unsigned g = blocks8((size_t)c->nfluid), gw = g < 4096u ? g : 4096u;        // (both multiples of 8)
    if (c->skip && (stages & 1)) {
        const unsigned nt = (c->nblk + 1023u) / 1024u;
        csf3d_tile_count<0><<<nt, 1024, 0, c->stream>>>(p, c->deep_now, c->tcnt);
        csf3d_tile_rank<0><<<nt, 1024, 0, c->stream>>>(p, c->deep_now, c->tcnt, c->pfx);
        csf3d_tile_count<1><<<nt, 1024, 0, c->stream>>>(p, c->deep_now, c->tcnt);
        csf3d_tile_rank<1><<<nt, 1024, 0, c->stream>>>(p, c->deep_now, c->tcnt, c->work);
    }
    const bool mrt = c->cfg.relaxation == LBMPM_RELAX_MRT;
    const bool deep_launch = c->skip && !FIRST;      // (nothing is deep in the first step after a set_*)
    if (e0) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
    if (deep_launch && (stages & 1)) {
        // the deep blocks' collision reads the last step's populations and its own flags, and writes its own blocks only: it runs on a
        // second stream beside the four launches of the full path (small, latency-bound launches when most of the lattice is bulk)
        LBMPM_HIP_TRY(hipEventRecord(c->ev_lists, c->stream));
        LBMPM_HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_lists, 0));
        if (mrt) { if (c->diag) csf3d_collide_deep<true, true><<<gw, 256, 0, c->stream2>>>(p); else csf3d_collide_deep<true, false><<<gw, 256, 0, c->stream2>>>(p); }
        else { if (c->diag) csf3d_collide_deep<false, true><<<gw, 256, 0, c->stream2>>>(p); else csf3d_collide_deep<false, false><<<gw, 256, 0, c->stream2>>>(p); }
        LBMPM_HIP_TRY(hipEventRecord(c->ev_deep, c->stream2));
    }
    if (stages & 1) csf3d_phase<FIRST><<<gw, 256, 0, c->stream>>>(p);
    if (stages & 2) {
        if (c->nwet) csf3d_solid_phi<<<blocks_of(c->nwet), 256, 0, c->stream>>>(p, c->wetlist, c->wethome);
        csf3d_gradient<<<gw, 256, 0, c->stream>>>(p);
    }
    if (stages & 4) {
        if (mrt) { if (c->diag) csf3d_collide<FIRST, true, true><<<gw, 256, 0, c->stream>>>(p); else csf3d_collide<FIRST, true, false><<<gw, 256, 0, c->stream>>>(p); }
        else { if (c->diag) csf3d_collide<FIRST, false, true><<<gw, 256, 0, c->stream>>>(p); else csf3d_collide<FIRST, false, false><<<gw, 256, 0, c->stream>>>(p); }
        if (deep_launch) LBMPM_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_deep, 0));
    }
    if (e1) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

int run_steps(lbmpm_rk3dcsf *c, int64_t nsteps, bool timed)
{
    if (!c->have_state) { set_error("lbmpm_rk3dcsf_step before set_macro / set_pdf"); return LBMPM_ERR_STATE; }
    if (c->cfg.ghost_lo || c->cfg.ghost_hi) { set_error("lbmpm_rk3dcsf_step: a slab with ghost planes steps by lbmpm_rk3dcsf_stage, the face messages in between"); return LBMPM_ERR_STATE; }
    for (int64_t s = 0; s < nsteps; ++s) {
        const CsfDev p = make_dev(c);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed && c->pool.take(&e0, &e1)) ++c->timed_steps;
        const int rc = c->first ? launch_step<true>(c, p, e0, e1) : launch_step<false>(c, p, e0, e1);
        if (rc != LBMPM_OK) return rc;
        std::swap(c->fA, c->fB);
        c->first = false;
        ++c->steps;
        c->diag_valid = c->diag;
    }
    return LBMPM_OK;
}

int upload(lbmpm_rk3dcsf *c, double *dst, const double *src, size_t count)
{
    LBMPM_HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return LBMPM_OK;
}

}  // namespace

extern "C" void lbmpm_rk3dcsf_destroy(lbmpm_rk3dcsf *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->wethome, c->rng, c->pfx, c->pure, c->deep_prev, c->bcblk, c->deep_now, c->work, c->tcnt, c->src, c->dom, c->meta, c->wetlist, c->cidx, c->cells, c->fA, c->fB, c->phi, c->G, c->nh, c->F, c->K, c->U, c->ns, c->obs};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    c->pool.destroy();
    if (c->ev_lists) (void)hipEventDestroy(c->ev_lists);
    if (c->ev_deep) (void)hipEventDestroy(c->ev_deep);
    for (hipEvent_t e : {c->ev_stage, c->ev_sent[0], c->ev_sent[1]}) if (e) (void)hipEventDestroy(e);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int lbmpm_rk3dcsf_create(const lbmpm_rk3dcsf_config *cfg, const uint8_t *is_domain, lbmpm_rk3dcsf **out)
{
    LBMPM_REQUIRE(cfg && is_domain && out, "lbmpm_rk3dcsf_create: null argument");
    LBMPM_REQUIRE(cfg->nx >= 1 && cfg->ny >= 1 && cfg->nz >= 8, "lbmpm_rk3dcsf_create: lattice %lld x %lld x %lld out of range (nz >= 8: two open planes and their ghosts at each end)",
                  (long long)cfg->nx, (long long)cfg->ny, (long long)cfg->nz);
    LBMPM_REQUIRE((double)cfg->nx * (double)cfg->ny * (double)cfg->nz < 2147483648., "lbmpm_rk3dcsf_create: more than 2^31 cells");
    LBMPM_REQUIRE(cfg->relaxation == LBMPM_RELAX_SRT || cfg->relaxation == LBMPM_RELAX_MRT, "bad relaxation %d", cfg->relaxation);
    LBMPM_REQUIRE(cfg->inlet_type == LBMPM_INLET_VELOCITY || cfg->inlet_type == LBMPM_INLET_PRESSURE, "bad inlet_type %d", cfg->inlet_type);
    LBMPM_REQUIRE(cfg->outlet_type == LBMPM_OUTLET_PRESSURE || cfg->outlet_type == LBMPM_OUTLET_CONVECTIVE, "bad outlet_type %d", cfg->outlet_type);
    LBMPM_REQUIRE(cfg->tau_type == 1 || cfg->tau_type == 2, "TauType must be 1 or 2");
    LBMPM_REQUIRE(cfg->tau_r > 0.5 && cfg->tau_b > 0.5, "TauR, TauB must exceed 0.5");
    if (cfg->wetting_type == 1) {
        set_error("WettingType 1 (Xu et al. 2017, AcceleratedRKGPU2D.py:1639-1679) rotates the normal in the plane: a 2-D rule without a 3-D form; "
                  "use WettingType 2 (Akai et al. 2018)");
        return LBMPM_ERR_UNSUPPORTED;
    }
    LBMPM_REQUIRE(cfg->wetting_type == 2 || cfg->wetting_type == 0, "WettingType must be 2 (or 0: no correction at the walls)");
    LBMPM_REQUIRE(cfg->variant == 0 || cfg->variant == 1, "variant must be 0 or 1");
    LBMPM_REQUIRE(cfg->bulk_epsilon >= 0. && cfg->bulk_epsilon <= 1.0e-3, "bulk_epsilon must lie in [0, 1e-3]");
    const bool slab = cfg->ghost_lo != 0 || cfg->ghost_hi != 0;
    LBMPM_REQUIRE(!slab || (cfg->ghost_lo == 2 && cfg->ghost_hi == 2), "ghost_lo, ghost_hi must be 0, 0 (the undivided lattice) or 2, 2 (a slab)");
    LBMPM_REQUIRE(cfg->nz - cfg->ghost_lo - cfg->ghost_hi >= 4, "lbmpm_rk3dcsf_create: a slab keeps at least 4 planes of its own between its ghost planes");
    const int64_t own_nz = cfg->nz - cfg->ghost_lo - cfg->ghost_hi;
    LBMPM_REQUIRE(!slab || (cfg->slab_z0 >= 0 && cfg->slab_z0 + own_nz <= cfg->global_nz && cfg->global_nz >= 8 && own_nz < cfg->global_nz),
                  "lbmpm_rk3dcsf_create: slab_z0 %lld + %lld planes of its own within global_nz %lld", (long long)cfg->slab_z0, (long long)own_nz, (long long)cfg->global_nz);
    const int64_t zoff = slab ? cfg->slab_z0 - cfg->ghost_lo : 0, nzg = slab ? cfg->global_nz : cfg->nz;
    const size_t pl = (size_t)cfg->nx * cfg->ny, N = pl * (size_t)cfg->nz;
    // the ghost planes copy the plane next to them cell by cell (the reference's kernels take the neighbour's index without looking)
    auto same = [&](int64_t za, int64_t zb) {
        for (size_t k = 0; k < pl; ++k) if ((is_domain[za * pl + k] == 1) != (is_domain[zb * pl + k] == 1)) return false;
        return true;
    };
    // (planes of the undivided lattice; a slab checks the ones it owns: the open planes and the planes they copy lie in one slab, >= 4 planes)
    auto owns = [&](int64_t zg) { return zg - zoff >= cfg->ghost_lo && zg - zoff < cfg->nz - cfg->ghost_hi; };
    if (owns(nzg - 1)) {
        LBMPM_REQUIRE(owns(nzg - 2), "lbmpm_rk3dcsf_create: the inlet plane and its ghost plane belong to one slab");
        LBMPM_REQUIRE(same(nzg - 1 - zoff, nzg - 2 - zoff), "lbmpm_rk3dcsf_create: the ghost plane nz-1 must have the mask of the inlet plane nz-2");
    }
    if (owns(0)) {
        if (cfg->outlet_type == LBMPM_OUTLET_CONVECTIVE) {
            LBMPM_REQUIRE(owns(3), "lbmpm_rk3dcsf_create: the planes 0 .. 3 of the convective outlet belong to one slab");
            LBMPM_REQUIRE(same(0 - zoff, 3 - zoff) && same(1 - zoff, 3 - zoff) && same(2 - zoff, 3 - zoff), "lbmpm_rk3dcsf_create: the convective outlet copies plane 3 onto the planes 2, 1, 0: their masks must coincide");
        } else {
            LBMPM_REQUIRE(owns(1), "lbmpm_rk3dcsf_create: the outlet plane and its ghost plane belong to one slab");
            LBMPM_REQUIRE(same(0 - zoff, 1 - zoff), "lbmpm_rk3dcsf_create: the ghost plane 0 must have the mask of the outlet plane 1");
        }
    }
    LBMPM_HIP_TRY(hipSetDevice(cfg->device));
    lbmpm_rk3dcsf *c = new (std::nothrow) lbmpm_rk3dcsf();
    if (!c) { set_error("out of host memory"); return LBMPM_ERR_NOMEM; }
    c->cfg = *cfg;
    c->nx = (int)cfg->nx; c->ny = (int)cfg->ny; c->nz = (int)cfg->nz;
    c->zoff = (int)zoff; c->nzg = (int)nzg;
    c->N = N; c->NS = (N + 15) / 16 * 16;
    // the populations are kept for fluid cells only, numbered in lattice order (a dense layout streams the solid cells of every
    // 128-byte line that holds a fluid cell: counted 1.5 x the bytes on the bench's porous medium)
    std::vector<uint32_t> hidx, hcells;
    try { hidx.resize(N); hcells.reserve(N); }
    catch (const std::bad_alloc &) { set_error("lbmpm_rk3dcsf_create: out of host memory (8 bytes per lattice cell for the set-up tables)"); delete c; return LBMPM_ERR_NOMEM; }
    try { c->pfirst.assign((size_t)cfg->nz + 1, 0u); }
    catch (const std::bad_alloc &) { set_error("lbmpm_rk3dcsf_create: out of host memory"); delete c; return LBMPM_ERR_NOMEM; }
    for (size_t k = 0; k < N; ++k) {
        if (k % pl == 0) c->pfirst[k / pl] = (uint32_t)hcells.size();
        const bool fl = is_domain[k] == 1;
        hidx[k] = fl ? (uint32_t)hcells.size() : 0xFFFFFFFFu;
        if (fl) hcells.push_back((uint32_t)k);
    }
    c->pfirst[(size_t)cfg->nz] = (uint32_t)hcells.size();
    c->nfluid = (int64_t)hcells.size();
    c->FS = ((size_t)c->nfluid + 15) / 16 * 16;
    // 38 planes a power of two apart would sit in the same HBM channels and cache sets cell by cell: an odd stride
    if ((c->FS * sizeof(double)) % 16384 == 0) c->FS += 1168;
    if (c->nfluid == 0) { set_error("lbmpm_rk3dcsf_create: the domain has no fluid cell (is_domain == 1 marks fluid)"); delete c; return LBMPM_ERR_INVALID; }
    {
        const hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return LBMPM_ERR_HIP; }
        if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_lists, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_deep, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_sent[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_sent[1], hipEventDisableTiming) != hipSuccess) { set_error("hipStreamCreate / hipEventCreate failed"); lbmpm_rk3dcsf_destroy(c); return LBMPM_ERR_HIP; }
    }
    int rc = LBMPM_OK;
#define TRY_RC(e) do { rc = (e); if (rc != LBMPM_OK) { lbmpm_rk3dcsf_destroy(c); return rc; } } while (0)
#define TRY_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) { set_error("%s failed: %s", #e, hipGetErrorString(e_)); lbmpm_rk3dcsf_destroy(c); return LBMPM_ERR_HIP; } } while (0)
    TRY_RC(dev_alloc(c, &c->dom, N));
    TRY_RC(dev_alloc(c, &c->meta, N));
    TRY_RC(dev_alloc(c, &c->cidx, N));
    TRY_RC(dev_alloc(c, &c->cells, (size_t)c->nfluid));
    TRY_RC(dev_alloc(c, &c->fA, 2 * Q * c->FS));
    TRY_RC(dev_alloc(c, &c->fB, 2 * Q * c->FS));
    TRY_RC(dev_alloc(c, &c->phi, c->NS));
    TRY_RC(dev_alloc(c, &c->G, 3 * c->NS));
    TRY_RC(dev_alloc(c, &c->nh, 3 * c->NS));
    TRY_RC(dev_alloc(c, &c->F, 3 * c->NS));
    TRY_RC(dev_alloc(c, &c->ns, 3 * c->NS));
    unsigned *counters = nullptr;
    TRY_HIP(hipMalloc(reinterpret_cast<void **>(&counters), 2 * sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(counters, 0, 2 * sizeof(unsigned), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->dom, is_domain, N, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->cidx, hidx.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->cells, hcells.data(), hcells.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    for (double *q : {c->phi, c->G, c->nh, c->F}) if (e == hipSuccess) e = hipMemsetAsync(q, 0, (q == c->phi ? 1 : 3) * c->NS * sizeof(double), c->stream);
    if (e == hipSuccess) {
        csf3d_setup_meta<<<blocks_of(N), 256, 0, c->stream>>>(c->nx, c->ny, c->nz, (unsigned)N, c->dom, c->meta, counters);
        csf3d_setup_normals<<<blocks_of(N), 256, 0, c->stream>>>(c->nx, c->ny, c->nz, (unsigned)N, c->NS, c->dom, c->meta, c->ns);
        e = hipMemcpyAsync(&c->nwet, counters, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(counters); set_error("set-up failed: %s", hipGetErrorString(e)); lbmpm_rk3dcsf_destroy(c); return LBMPM_ERR_HIP; }
    if (c->nwet) {
        const unsigned nb = blocks_of(N);
        uint32_t *cnt = nullptr;
        std::vector<uint32_t> h(nb);
        rc = dev_alloc(c, &c->wetlist, c->nwet);
        if (rc == LBMPM_OK && hipMalloc(reinterpret_cast<void **>(&cnt), nb * sizeof(uint32_t)) != hipSuccess) { set_error("hipMalloc failed"); rc = LBMPM_ERR_NOMEM; }
        if (rc == LBMPM_OK) {
            csf3d_setup_wetcount<<<nb, 256, 0, c->stream>>>((unsigned)N, c->meta, cnt);
            e = hipMemcpyAsync(h.data(), cnt, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            uint32_t run = 0;
            for (unsigned k = 0; k < nb; ++k) { const uint32_t v = h[k]; h[k] = run; run += v; }
            if (e == hipSuccess && run != c->nwet) { set_error("set-up: %u wetting solids counted, %u listed", c->nwet, run); rc = LBMPM_ERR_STATE; }
            if (e == hipSuccess && rc == LBMPM_OK) e = hipMemcpyAsync(cnt, h.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && rc == LBMPM_OK) {
                csf3d_setup_wetlist<<<nb, 256, 0, c->stream>>>((unsigned)N, c->meta, cnt, c->wetlist);
                e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = hipGetLastError();
            }
            if (e != hipSuccess) { set_error("set-up failed: %s", hipGetErrorString(e)); rc = LBMPM_ERR_HIP; }
        }
        if (cnt) (void)hipFree(cnt);
        if (rc != LBMPM_OK) { (void)hipFree(counters); lbmpm_rk3dcsf_destroy(c); return rc; }
    }
    {   // the bulk skip's tables
        c->skip = cfg->variant == 0;
        c->nblk = blocks_of((size_t)c->nfluid);
        const unsigned nb = c->nblk;
        uint32_t *lo0 = nullptr, *hi0 = nullptr, *lo1 = nullptr, *hi1 = nullptr;
        rc = dev_alloc(c, &c->pure, (size_t)nb + 4);      // (csf3d_face_flags works on whole words)
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->deep_prev, nb);
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->bcblk, nb);
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->rng, 2 * (size_t)nb);
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->pfx, 2 * ((size_t)nb + 1));
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->deep_now, nb);
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->work, 3 * ((size_t)nb + 1));
        if (rc == LBMPM_OK) rc = dev_alloc(c, &c->tcnt, 3 * ((size_t)nb / 1024 + 1));
        if (rc == LBMPM_OK && c->skip) rc = dev_alloc(c, &c->src, 18 * c->FS);
        if (rc == LBMPM_OK && c->nwet) rc = dev_alloc(c, &c->wethome, c->nwet);
        if (rc == LBMPM_OK && hipMalloc(reinterpret_cast<void **>(&lo0), 4 * (size_t)nb * sizeof(uint32_t)) != hipSuccess) { set_error("hipMalloc failed"); rc = LBMPM_ERR_NOMEM; }
        if (rc == LBMPM_OK) {
            hi0 = lo0 + nb; lo1 = hi0 + nb; hi1 = lo1 + nb;
            const CsfDev p = make_dev(c);
            e = hipMemsetAsync(lo0, 0xFF, nb * sizeof(uint32_t), c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(hi0, 0, nb * sizeof(uint32_t), c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(lo1, 0xFF, nb * sizeof(uint32_t), c->stream);
            if (e == hipSuccess) e = hipMemsetAsync(hi1, 0, nb * sizeof(uint32_t), c->stream);
            for (uint8_t *q : {c->pure, c->deep_prev, c->bcblk}) if (e == hipSuccess) e = hipMemsetAsync(q, 0, nb, c->stream);
            if (e == hipSuccess) {
                csf3d_setup_ranges<<<nb, 256, 0, c->stream>>>(p, 0, lo0, hi0, nullptr, nullptr, c->bcblk);
                csf3d_setup_ranges<<<nb, 256, 0, c->stream>>>(p, 1, lo1, hi1, lo0, hi0, c->bcblk);
                if (c->src) csf3d_setup_src<<<nb, 256, 0, c->stream>>>(p, c->src);
                if (c->nwet) csf3d_setup_wethome<<<blocks_of(c->nwet), 256, 0, c->stream>>>(p, c->wetlist, c->wethome);
                std::vector<uint32_t> h(2 * (size_t)nb);
                e = hipMemcpyAsync(h.data(), lo1, 2 * (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                for (uint32_t &v : h) v >>= 8;          // cells -> blocks
                if (e == hipSuccess) e = hipMemcpyAsync(c->rng, h.data(), 2 * (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = hipGetLastError();
            }
            if (e != hipSuccess) { set_error("set-up failed: %s", hipGetErrorString(e)); rc = LBMPM_ERR_HIP; }
        }
        if (lo0) (void)hipFree(lo0);
        if (rc != LBMPM_OK) { (void)hipFree(counters); lbmpm_rk3dcsf_destroy(c); return rc; }
    }
    (void)hipFree(counters);
#undef TRY_RC
#undef TRY_HIP
    *out = c;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_enable_diagnostics(lbmpm_rk3dcsf *c, int on)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (on && !c->K) {
        int rc = dev_alloc(c, &c->K, c->NS); if (rc) return rc;
        rc = dev_alloc(c, &c->U, 3 * c->NS); if (rc) return rc;
        LBMPM_HIP_TRY(hipMemsetAsync(c->K, 0, c->NS * sizeof(double), c->stream));
        LBMPM_HIP_TRY(hipMemsetAsync(c->U, 0, 3 * c->NS * sizeof(double), c->stream));
    }
    c->diag = on != 0;
    if (!on) c->diag_valid = false;
    return LBMPM_OK;
}

static int reset_state(lbmpm_rk3dcsf *c, const double *fx, const double *fy, const double *fz)
{
    std::vector<double> tmp;
    const double *src[3] = {fx, fy, fz};
    for (int a = 0; a < 3; ++a) {
        if (src[a]) { const int rc = upload(c, c->F + a * c->NS, src[a], c->N); if (rc) return rc; }
        else LBMPM_HIP_TRY(hipMemsetAsync(c->F + a * c->NS, 0, c->NS * sizeof(double), c->stream));
    }
    LBMPM_HIP_TRY(hipMemsetAsync(c->G, 0, 3 * c->NS * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->nh, 0, 3 * c->NS * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->phi, 0, c->NS * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->pure, 0, c->nblk, c->stream));          // nothing is known about the new state: the first step takes the full path
    LBMPM_HIP_TRY(hipMemsetAsync(c->deep_prev, 0, c->nblk, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    c->first = true; c->have_state = true; c->steps = 0; c->diag_valid = false; c->next_stage = 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_set_macro(lbmpm_rk3dcsf *c, const double *rho_r, const double *rho_b, const double *vx, const double *vy, const double *vz)
{
    LBMPM_REQUIRE(c && rho_r && rho_b, "lbmpm_rk3dcsf_set_macro: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    // staged through the gradient and normal arrays (3 NS doubles each; reset_state clears them)
    double *st[5] = {c->G, c->G + c->NS, c->G + 2 * c->NS, c->nh, c->nh + c->NS};
    const double *src[5] = {rho_r, rho_b, vx, vy, vz};
    for (int a = 0; a < 5; ++a) if (src[a]) { const int rc = upload(c, st[a], src[a], c->N); if (rc) return rc; }
    const CsfDev p = make_dev(c);
    csf3d_init<<<blocks_of(c->N), 256, 0, c->stream>>>(p, c->fA, st[0], st[1], vx ? st[2] : nullptr, vy ? st[3] : nullptr, vz ? st[4] : nullptr);
    LBMPM_HIP_TRY(hipGetLastError());
    return reset_state(c, nullptr, nullptr, nullptr);
}

extern "C" int lbmpm_rk3dcsf_set_pdf(lbmpm_rk3dcsf *c, const double *pdf_r, const double *pdf_b, const double *fx, const double *fy, const double *fz)
{
    LBMPM_REQUIRE(c && pdf_r && pdf_b, "lbmpm_rk3dcsf_set_pdf: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    double *st = nullptr;                        // [2][N][19] in the host's layout, for the length of the call
    LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st), 2 * c->N * Q * sizeof(double)));
    int rc = upload(c, st, pdf_r, c->N * Q);
    if (rc == LBMPM_OK) rc = upload(c, st + c->N * Q, pdf_b, c->N * Q);
    if (rc == LBMPM_OK) {
        const CsfDev p = make_dev(c);
        csf3d_import<<<blocks_of(c->N), 256, 0, c->stream>>>(p, c->fA, st);
        if (hipGetLastError() != hipSuccess) { set_error("csf3d_import did not launch"); rc = LBMPM_ERR_HIP; }
    }
    if (rc == LBMPM_OK) rc = reset_state(c, fx, fy, fz);        // (synchronises the stream)
    else (void)hipStreamSynchronize(c->stream);
    (void)hipFree(st);
    return rc;
}

extern "C" int lbmpm_rk3dcsf_step(lbmpm_rk3dcsf *c, int64_t nsteps)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk3dcsf_step: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    return run_steps(c, nsteps, false);
}

extern "C" int lbmpm_rk3dcsf_stage(lbmpm_rk3dcsf *c, int stage)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_REQUIRE(stage >= 0 && stage <= 2, "lbmpm_rk3dcsf_stage: stage %d", stage);
    if (!c->have_state) { set_error("lbmpm_rk3dcsf_stage before set_macro / set_pdf"); return LBMPM_ERR_STATE; }
    if (stage != c->next_stage) { set_error("lbmpm_rk3dcsf_stage: stage %d is next, not %d", c->next_stage, stage); return LBMPM_ERR_STATE; }
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const CsfDev p = make_dev(c);
    const int rc = c->first ? launch_step<true>(c, p, nullptr, nullptr, 1 << stage) : launch_step<false>(c, p, nullptr, nullptr, 1 << stage);
    if (rc != LBMPM_OK) return rc;
    if (stage == 2) {
        std::swap(c->fA, c->fB);
        c->first = false;
        ++c->steps;
        c->diag_valid = c->diag;
    }
    c->next_stage = (stage + 1) % 3;
    LBMPM_HIP_TRY(hipEventRecord(c->ev_stage, c->stream));
    return LBMPM_OK;
}

namespace {
// A face message as runs of doubles inside the context's arrays: what this context SENDS through `face` (recv = false: its edge planes)
// or where what comes through `face` lands (recv = true: its ghost planes).  The two sides list their runs in the same order.
struct Run { double *ptr; size_t count; };
int face_runs(lbmpm_rk3dcsf *c, int msg, int face, bool recv, Run runs[10])
{
    const int g = face == 0 ? c->cfg.ghost_lo : c->cfg.ghost_hi;
    if (g == 0) return 0;
    const int lo = c->cfg.ghost_lo, top = c->nz - c->cfg.ghost_hi;       // own planes: lo .. top - 1
    if (msg == LBMPM_CSF_MSG_PDF) {
        // the populations that cross the face, of the plane next to it: moving down (c_z < 0) through the low face, up through the high one;
        // the receiver pulls them out of its first ghost plane.  Fluid cells of a plane are one stretch of every population plane.
        static const int DN[5] = {6, 12, 13, 16, 17}, UP[5] = {5, 11, 14, 15, 18};
        const int z = face == 0 ? (recv ? lo - 1 : lo) : (recv ? top : top - 1);
        const int *dir = (face == 0) != recv ? DN : UP;      // sent through the low face: down; received through the low face: up
        const size_t first = c->pfirst[(size_t)z], count = c->pfirst[(size_t)z + 1] - first;
        for (int col = 0; col < 2; ++col)
            for (int a = 0; a < 5; ++a) runs[col * 5 + a] = Run{c->fA + ((size_t)col * Q + (size_t)dir[a]) * c->FS + first, count};
        return 10;
    }
    return 0;                                    // (phi and n: cell_msg)
}
bool msg_ok(int msg, int face) { return msg >= 0 && msg <= 2 && (face == 0 || face == 1); }
// n runs from `from` to `to` (either side a packed buffer when its runs are null: consecutive stretches of `buf`), one launch on `stream`
int copy_runs(const Run *from, const Run *to, const double *buf_in, double *buf_out, const Run *shape, int n, hipStream_t stream)
{
    RunSet r;
    unsigned most = 0;
    size_t off = 0;
    for (int k = 0; k < 10; ++k) {
        r.count[k] = k < n ? (unsigned)shape[k].count : 0u;
        r.src[k] = k < n ? (from ? from[k].ptr : buf_in + off) : nullptr;
        r.dst[k] = k < n ? (to ? to[k].ptr : buf_out + off) : nullptr;
        if (k < n) { off += shape[k].count; most = most > r.count[k] ? most : r.count[k]; }
    }
    if (n == 0 || most == 0) return LBMPM_OK;
    unsigned gx = (most + 2047u) / 2048u;            // eight doubles per thread
    if (gx > 1024u) gx = 1024u;
    csf3d_copy_runs<<<dim3(gx, (unsigned)n), 256, 0, stream>>>(r);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}
// the fluid cells of the two planes next to a face (sent) / beyond it (received), as a stretch of the cells' numbers: the flags of csf3d_face_flags
void flag_cells(const lbmpm_rk3dcsf *c, int face, bool recv, unsigned &j0, unsigned &count)
{
    const int lo = c->cfg.ghost_lo, top = c->nz - c->cfg.ghost_hi;
    const int z = face == 0 ? (recv ? 0 : lo) : (recv ? top : top - 2);
    j0 = c->pfirst[(size_t)z]; count = c->pfirst[(size_t)z + 2] - j0;
}
// phi (two planes) and n (the plane next to the face / the first ghost plane) travel for fluid cells only: the receiver recomputes phi on
// the walls of its first ghost plane itself (same inputs, same order), nobody reads the second one's walls or a wall's n
struct CellMsg { double *base; size_t stride; unsigned ncomp, j0, count; };
CellMsg cell_msg(const lbmpm_rk3dcsf *c, int msg, int face, bool recv)
{
    const int lo = c->cfg.ghost_lo, top = c->nz - c->cfg.ghost_hi;
    CellMsg m;
    if (msg == LBMPM_CSF_MSG_PHI) {
        const int z = face == 0 ? (recv ? 0 : lo) : (recv ? top : top - 2);
        m.base = c->phi; m.stride = 0; m.ncomp = 1;
        m.j0 = c->pfirst[(size_t)z]; m.count = c->pfirst[(size_t)z + 2] - m.j0;
    } else {
        const int z = face == 0 ? (recv ? lo - 1 : lo) : (recv ? top : top - 1);
        m.base = c->nh; m.stride = c->NS; m.ncomp = 3;
        m.j0 = c->pfirst[(size_t)z]; m.count = c->pfirst[(size_t)z + 1] - m.j0;
    }
    return m;
}
int copy_cells(const CellMsg *from, const lbmpm_rk3dcsf *fc, const CellMsg *to, const lbmpm_rk3dcsf *tc, const double *buf_in, double *buf_out, hipStream_t stream)
{
    const CellMsg &shape = from ? *from : *to;
    if (shape.count == 0) return LBMPM_OK;
    unsigned gx = (shape.count + 1023u) / 1024u;
    if (gx > 1024u) gx = 1024u;
    csf3d_copy_cells<<<dim3(gx, shape.ncomp), 256, 0, stream>>>(from ? from->base : buf_in, from ? fc->cells : nullptr, from ? from->j0 : 0u, from ? from->stride : 0,
                                                                 to ? to->base : buf_out, to ? tc->cells : nullptr, to ? to->j0 : 0u, to ? to->stride : 0, shape.count);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}
size_t flag_doubles(const lbmpm_rk3dcsf *c, int face, bool recv)
{
    unsigned j0, count;
    flag_cells(c, face, recv, j0, count);
    return ((size_t)count + 7) / 8;
}
}  // namespace

static int64_t face_total(const lbmpm_rk3dcsf *c, int msg, int face, bool recv)
{
    if (!c || !msg_ok(msg, face)) return 0;
    if (msg != LBMPM_CSF_MSG_PDF) {
        if ((face == 0 ? c->cfg.ghost_lo : c->cfg.ghost_hi) == 0) return 0;
        const CellMsg m = cell_msg(c, msg, face, recv);
        return (int64_t)m.ncomp * m.count;
    }
    Run runs[10];
    const int n = face_runs(const_cast<lbmpm_rk3dcsf *>(c), msg, face, recv, runs);
    int64_t t = 0;
    for (int k = 0; k < n; ++k) t += (int64_t)runs[k].count;
    if (n && msg == LBMPM_CSF_MSG_PDF) t += (int64_t)flag_doubles(c, face, recv);      // one byte per fluid cell of two planes behind the populations
    return t;
}
extern "C" int64_t lbmpm_rk3dcsf_face_doubles(const lbmpm_rk3dcsf *c, int msg, int face) { return face_total(c, msg, face, false); }
extern "C" int64_t lbmpm_rk3dcsf_face_doubles_in(const lbmpm_rk3dcsf *c, int msg, int face) { return face_total(c, msg, face, true); }

extern "C" int lbmpm_rk3dcsf_face_pack(lbmpm_rk3dcsf *c, int msg, int face, double *buf)
{
    LBMPM_REQUIRE(c && buf && msg_ok(msg, face), "lbmpm_rk3dcsf_face_pack: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if ((face == 0 ? c->cfg.ghost_lo : c->cfg.ghost_hi) == 0) return LBMPM_OK;
    if (msg != LBMPM_CSF_MSG_PDF) { const CellMsg m = cell_msg(c, msg, face, false); return copy_cells(&m, c, nullptr, nullptr, nullptr, buf, c->stream); }
    Run runs[10];
    const int n = face_runs(c, msg, face, false, runs);
    { const int rc = copy_runs(runs, nullptr, nullptr, buf, runs, n, c->stream); if (rc) return rc; }
    for (int k = 0; k < n; ++k) buf += runs[k].count;
    if (n && msg == LBMPM_CSF_MSG_PDF) {
        unsigned j0, count;
        flag_cells(c, face, false, j0, count);
        if (count) csf3d_face_flags_pack<<<blocks_of(count), 256, 0, c->stream>>>(c->pure, j0, count, reinterpret_cast<uint8_t *>(buf));
        LBMPM_HIP_TRY(hipGetLastError());
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_face_unpack(lbmpm_rk3dcsf *c, int msg, int face, const double *buf)
{
    LBMPM_REQUIRE(c && buf && msg_ok(msg, face), "lbmpm_rk3dcsf_face_unpack: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if ((face == 0 ? c->cfg.ghost_lo : c->cfg.ghost_hi) == 0) return LBMPM_OK;
    if (msg != LBMPM_CSF_MSG_PDF) { const CellMsg m = cell_msg(c, msg, face, true); return copy_cells(nullptr, nullptr, &m, c, buf, nullptr, c->stream); }
    Run runs[10];
    const int n = face_runs(c, msg, face, true, runs);
    { const int rc = copy_runs(nullptr, runs, buf, nullptr, runs, n, c->stream); if (rc) return rc; }
    for (int k = 0; k < n; ++k) buf += runs[k].count;
    if (n && msg == LBMPM_CSF_MSG_PDF) {
        unsigned j0, count;
        flag_cells(c, face, true, j0, count);
        if (count) csf3d_face_flags<<<((j0 + count - 1u) >> 8) - (j0 >> 8) + 1u, 256, 0, c->stream>>>(reinterpret_cast<const uint8_t *>(buf), nullptr, 0u, c->pure, j0, count);
        LBMPM_HIP_TRY(hipGetLastError());
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_face_copy(lbmpm_rk3dcsf *src, int src_face, lbmpm_rk3dcsf *dst, int msg)
{
    LBMPM_REQUIRE(src && dst && src != dst && msg_ok(msg, src_face), "lbmpm_rk3dcsf_face_copy: bad argument");
    Run out[10], in[10];
    int n = 0;
    CellMsg cm_out{}, cm_in{};
    if ((src_face == 0 ? src->cfg.ghost_lo : src->cfg.ghost_hi) == 0 || (src_face == 0 ? dst->cfg.ghost_hi : dst->cfg.ghost_lo) == 0) {
        set_error("lbmpm_rk3dcsf_face_copy: the two contexts do not share that face"); return LBMPM_ERR_INVALID;
    }
    if (msg == LBMPM_CSF_MSG_PDF) {
        n = face_runs(src, msg, src_face, false, out);
        const int m = face_runs(dst, msg, 1 - src_face, true, in);
        if (n == 0 || n != m) { set_error("lbmpm_rk3dcsf_face_copy: the two contexts do not share that face"); return LBMPM_ERR_INVALID; }
        for (int k = 0; k < n; ++k)
            if (out[k].count != in[k].count) { set_error("lbmpm_rk3dcsf_face_copy: the masks of the two sides of the face differ (%zu / %zu cells)", out[k].count, in[k].count); return LBMPM_ERR_INVALID; }
    } else {
        cm_out = cell_msg(src, msg, src_face, false); cm_in = cell_msg(dst, msg, 1 - src_face, true);
        if (cm_out.count != cm_in.count) { set_error("lbmpm_rk3dcsf_face_copy: the masks of the two sides of the face differ (%u / %u cells)", cm_out.count, cm_in.count); return LBMPM_ERR_INVALID; }
    }
    // on the sender's stream, once the receiver's last stage (which may read the planes written here) has run; the receiver's stream then waits
    LBMPM_HIP_TRY(hipSetDevice(src->cfg.device));
    if (src->cfg.device != dst->cfg.device) {    // two GPUs of one process: the sender's kernels write the receiver's memory
        const hipError_t e = hipDeviceEnablePeerAccess(dst->cfg.device, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        else if (e != hipSuccess) { set_error("lbmpm_rk3dcsf_face_copy: device %d cannot reach device %d (%s); use face_pack / face_unpack", src->cfg.device, dst->cfg.device, hipGetErrorString(e)); return LBMPM_ERR_UNSUPPORTED; }
    }
    LBMPM_HIP_TRY(hipStreamWaitEvent(src->stream, dst->ev_stage, 0));
    if (msg == LBMPM_CSF_MSG_PDF) { const int rc = copy_runs(out, in, nullptr, nullptr, out, n, src->stream); if (rc) return rc; }
    else { const int rc = copy_cells(&cm_out, src, &cm_in, dst, nullptr, nullptr, src->stream); if (rc) return rc; }
    if (msg == LBMPM_CSF_MSG_PDF) {
        unsigned sj0, scount, dj0, dcount;
        flag_cells(src, src_face, false, sj0, scount);
        flag_cells(dst, 1 - src_face, true, dj0, dcount);
        if (scount != dcount) { set_error("lbmpm_rk3dcsf_face_copy: the masks of the two sides of the face differ"); return LBMPM_ERR_INVALID; }
        if (scount) csf3d_face_flags<<<((dj0 + scount - 1u) >> 8) - (dj0 >> 8) + 1u, 256, 0, src->stream>>>(nullptr, src->pure, sj0, dst->pure, dj0, scount);
        LBMPM_HIP_TRY(hipGetLastError());
    }
    LBMPM_HIP_TRY(hipEventRecord(src->ev_sent[src_face], src->stream));
    LBMPM_HIP_TRY(hipSetDevice(dst->cfg.device));
    LBMPM_HIP_TRY(hipStreamWaitEvent(dst->stream, src->ev_sent[src_face], 0));
    // (the sender's next stage must not overwrite its planes before the copy has run: it is queued behind it on the same stream)
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_step_timed(lbmpm_rk3dcsf *c, int64_t nsteps, double *ms_total, double *ms_dominant)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk3dcsf_step_timed: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t pairs = (size_t)(nsteps < 4096 ? nsteps : 4096);
    if (c->pool.reserve(pairs + 1) != LBMPM_OK) { set_error("hipEventCreate failed"); return LBMPM_ERR_HIP; }
    c->pool.reset();
    c->timed_steps = 0;
    hipEvent_t t0, t1;
    c->pool.take(&t0, &t1);
    LBMPM_HIP_TRY(hipEventRecord(t0, c->stream));
    const int rc = run_steps(c, nsteps, true);
    if (rc != LBMPM_OK) return rc;
    LBMPM_HIP_TRY(hipEventRecord(t1, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    LBMPM_HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    if (ms_total) *ms_total = ms;
    if (ms_dominant) {
        double s = 0.0;
        for (size_t k = 2; k + 1 < c->pool.used; k += 2) {
            float m = 0.f;
            LBMPM_HIP_TRY(hipEventElapsedTime(&m, c->pool.ev[k], c->pool.ev[k + 1]));
            s += m;
        }
        *ms_dominant = c->timed_steps ? s * (double)nsteps / (double)c->timed_steps : 0.0;
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_sync(lbmpm_rk3dcsf *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

static int observe(lbmpm_rk3dcsf *c, bool rec, double *pdf)
{
    if (!c->obs) { const int rc = dev_alloc(c, &c->obs, 6 * c->N); if (rc) return rc; }
    const CsfDev p = make_dev(c);
    double *rho = c->obs, *u = rho + 2 * c->N, *phi = u + 3 * c->N;
    const unsigned g = blocks_of(c->N);
    if (c->first) { if (rec) csf3d_observe<true, true><<<g, 256, 0, c->stream>>>(p, pdf, rho, u, phi); else csf3d_observe<true, false><<<g, 256, 0, c->stream>>>(p, pdf, rho, u, phi); }
    else { if (rec) csf3d_observe<false, true><<<g, 256, 0, c->stream>>>(p, pdf, rho, u, phi); else csf3d_observe<false, false><<<g, 256, 0, c->stream>>>(p, pdf, rho, u, phi); }
    LBMPM_HIP_TRY(hipGetLastError());
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3dcsf_get_field(lbmpm_rk3dcsf *c, int field, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk3dcsf_get_field: null argument");
    LBMPM_REQUIRE(c->have_state, "lbmpm_rk3dcsf_get_field before set_macro / set_pdf");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t N = c->N, NS = c->NS;
    auto down = [&](const double *src, size_t count) -> int {
        LBMPM_HIP_TRY(hipMemcpy(out, src, count * sizeof(double), hipMemcpyDeviceToHost));
        return LBMPM_OK;
    };
    const bool rec = field >= LBMPM_RK3DCSF_REC_PDF_R;
    switch (field) {
    case LBMPM_RK3DCSF_PDF_R: case LBMPM_RK3DCSF_PDF_B: case LBMPM_RK3DCSF_RHO_R: case LBMPM_RK3DCSF_RHO_B:
    case LBMPM_RK3DCSF_REC_PDF_R: case LBMPM_RK3DCSF_REC_PDF_B: case LBMPM_RK3DCSF_REC_RHO_R: case LBMPM_RK3DCSF_REC_RHO_B:
    case LBMPM_RK3DCSF_REC_VX: case LBMPM_RK3DCSF_REC_VY: case LBMPM_RK3DCSF_REC_VZ: case LBMPM_RK3DCSF_REC_PHI: {
        const int base = rec ? field - LBMPM_RK3DCSF_REC_PDF_R : field;
        double *pdf = nullptr;                   // [2][N][19] in the host's layout, for the length of the call
        if (base <= 1) LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&pdf), 2 * N * Q * sizeof(double)));
        int rc = observe(c, rec, pdf);
        if (rc == LBMPM_OK && base <= 1) rc = down(pdf + (size_t)base * N * Q, N * Q);
        if (pdf) (void)hipFree(pdf);
        if (rc != LBMPM_OK || base <= 1) return rc;
        const double *rho = c->obs, *u = rho + 2 * N, *phi = u + 3 * N;
        if (base <= 3) return down(rho + (size_t)(base - 2) * N, N);
        if (field == LBMPM_RK3DCSF_REC_PHI) return down(phi, N);
        return down(u + (size_t)(field - LBMPM_RK3DCSF_REC_VX) * N, N);
    }
    case LBMPM_RK3DCSF_PHI: return down(c->phi, N);
    case LBMPM_RK3DCSF_GX: case LBMPM_RK3DCSF_GY: case LBMPM_RK3DCSF_GZ: return down(c->G + (size_t)(field - LBMPM_RK3DCSF_GX) * NS, N);
    case LBMPM_RK3DCSF_FX: case LBMPM_RK3DCSF_FY: case LBMPM_RK3DCSF_FZ: return down(c->F + (size_t)(field - LBMPM_RK3DCSF_FX) * NS, N);
    case LBMPM_RK3DCSF_NSX: case LBMPM_RK3DCSF_NSY: case LBMPM_RK3DCSF_NSZ: return down(c->ns + (size_t)(field - LBMPM_RK3DCSF_NSX) * NS, N);
    case LBMPM_RK3DCSF_VX: case LBMPM_RK3DCSF_VY: case LBMPM_RK3DCSF_VZ: case LBMPM_RK3DCSF_K:
        if (!c->diag_valid) { set_error("u and K of the last step are kept with lbmpm_rk3dcsf_enable_diagnostics(ctx, 1) before stepping"); return LBMPM_ERR_STATE; }
        return field == LBMPM_RK3DCSF_K ? down(c->K, N) : down(c->U + (size_t)(field - LBMPM_RK3DCSF_VX) * NS, N);
    case LBMPM_RK3DCSF_KIND: {
        std::vector<uint32_t> m(N);
        LBMPM_HIP_TRY(hipMemcpy(m.data(), c->meta, N * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < N; ++k) out[k] = (double)((m[k] >> KIND_SHIFT) & 3u);
        return LBMPM_OK;
    }
    default:
        set_error("unknown field id %d", field);
        return LBMPM_ERR_INVALID;
    }
}

extern "C" int64_t lbmpm_rk3dcsf_num_fluid_nodes(const lbmpm_rk3dcsf *c) { return c ? c->nfluid : 0; }
extern "C" int64_t lbmpm_rk3dcsf_num_wetting_solids(const lbmpm_rk3dcsf *c) { return c ? (int64_t)c->nwet : 0; }
extern "C" int64_t lbmpm_rk3dcsf_bulk_cells(lbmpm_rk3dcsf *c)
{
    if (!c || !c->skip || !c->nblk) return 0;
    if (hipSetDevice(c->cfg.device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    std::vector<uint8_t> h(c->nblk);
    if (hipMemcpy(h.data(), c->deep_prev, c->nblk, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int64_t n = 0;
    for (unsigned k = 0; k < c->nblk; ++k) if (h[k]) n += k + 1 < c->nblk ? 256 : c->nfluid - 256 * (int64_t)k;
    return n;
}
extern "C" int64_t lbmpm_rk3dcsf_steps_done(const lbmpm_rk3dcsf *c) { return c ? c->steps : 0; }
extern "C" int64_t lbmpm_rk3dcsf_device_bytes(const lbmpm_rk3dcsf *c) { return c ? c->bytes : 0; }
extern "C" const char *lbmpm_rk3dcsf_dominant_kernel(const lbmpm_rk3dcsf *c) { (void)c; return "csf3d_collide"; }
