// rk3dq.h -- "q23" storage of the D3Q19 colour-gradient solver; included by rk3d.hip inside its
// anonymous namespace (it uses RK3Dev, RowTab, row_cell, TileRows, March, collide_store, ... of that file).
//
// The recolouring (AcceleratedRKGPU2D.py:1241-1267; collide_store in rk3d.hip) writes
//     f_R,i = k_R g_i + c_i (e_i . A),     f_B,i = (1 - k_R) g_i - c_i (e_i . A),
//     k_R = rho_R / rho,   A = rho_R rho_B / rho^2 * G / |G|,   c_i = beta w_i / |e_i|   (two constants: rcA, rcD)
// with g_i the post-collision colour-blind population: the 38 values a cell hands to the next step are an
// affine image of 23 -- the 19 g_i, k_R and the vector A.  This storage keeps those 23 doubles per fluid cell
// (184 B instead of 304 B; one step moves 368 B per cell instead of 608 B) and rebuilds the red population of
// every pulled direction from the *upstream* cell's k_R and A, which the marching kernel holds in an LDS tile:
//     pulled  g_i(x)   = g_i(x - e_i)                      or  g_i'(x)  (i' opposite; x - e_i not fluid: bounce-back)
//     pulled  f_R,i(x) = k_R(x - e_i) g_i(x - e_i) + c_i e_i.A(x - e_i)     or  k_R(x) g_i'(x) - c_i e_i.A(x)
// The collision needs only the 19 pulled g_i (= f_R,i + f_B,i exactly) and rho_R = sum_i f_R,i (rho_B = rho - rho_R);
// the per-colour Zou-He planes need the red sum by the class of c_z (zouhe_inlet / zouhe_outlet of rk3d.hip are
// linear in the populations apart from rho_c = (S0 + 2 S+) / (1 + u_c)), so every cell accumulates three class sums.
// Mathematically the same step as the 38-value kernels; results differ at round-off (tests: 1e-12 against
// rk3dc_fused, 1e-10 against the oracle).
//
// Layout: compact storage as rk3dc_* (fluid cells only, tile-major numbering, row-segment records), runs padded to
// 16 cells (= one 128-byte line of 8-byte values); per plane zl with cnt cells
//     g[zl][q][j]  q = 0..18, 8 bytes            s[zl][j] = {k_R, A_x, A_y, A_z}, 32 bytes        (184 * cnt bytes per plane)
// Ghost planes of the global lattice (z = 0, nz-1) are never collided or stored: the planes next to them overwrite
// everything they pull from there by the Zou-He closures, and the ghost's phase field equals its neighbour's.

constexpr int QS = 23;                         // doubles per stored cell
constexpr unsigned CELLB = QS * 8u;
constexpr int QLINE = 16;                      // cells per 128-byte line of 8-byte values: padding unit of the tile runs
constexpr int SR = 12, SC = 68;                // scalar tile: 64 x 8 tile + 2 cells of halo (the rim cells pull too)
constexpr int SCOMP = SR * SC;                 // doubles between the four components of the LDS scalar tile
constexpr int SSLOT = 4 * SCOMP;               // doubles between the ring slots (one per plane)

// The plane table through the constant address space: it does not change while a kernel runs, and only so may the compiler use
// scalar loads for it (s_load, lgkmcnt).  As ordinary global loads they count in vmcnt with the pulls and the stores, and every
// s_waitcnt in front of their use drains the pulls in flight.
typedef const unsigned long long __attribute__((address_space(4))) *const_u64_ptr;
__device__ __forceinline__ unsigned long long pstart_of(const RK3Dev &p, int zl) { return ((const_u64_ptr)p.pstart)[zl]; }

struct PlaneAddrQ {
    const char *base;            // g_0 of the plane below the pulled one
    unsigned off[3], cnt[3];     // byte offset of the three planes' blocks from base, stored cells per plane
};

__device__ __forceinline__ PlaneAddrQ plane_addr_q(const RK3Dev &p, const double *f, int zl)
{
    PlaneAddrQ a;
    const unsigned long long p1 = pstart_of(p, zl), p2 = pstart_of(p, zl + 1);
    const unsigned long long p0 = zl > 0 ? pstart_of(p, zl - 1) : p1, p3 = zl <= p.nzl ? pstart_of(p, zl + 2) : p2;   // no plane beyond the halo planes
    a.base = reinterpret_cast<const char *>(f) + (size_t)p0 * CELLB;
    a.cnt[0] = (unsigned)(p1 - p0); a.cnt[1] = (unsigned)(p2 - p1); a.cnt[2] = (unsigned)(p3 - p2);
    a.off[0] = 0u; a.off[1] = a.cnt[0] * CELLB; a.off[2] = a.off[1] + a.cnt[1] * CELLB;
    return a;
}

// pull of the 19 colour-blind populations of the node at bit b of the row rows(zl, 0) (fluid there); 8-byte loads,
// bounce-back folded into the address as in pull3c
// ASM: the load is an asm statement, i.e. absent from the compiler's s_waitcnt bookkeeping -- the marching kernel waits for
// its own-cell pulls itself (see there); everywhere else the compiler does.
template <bool ASM>
__device__ __forceinline__ double ldq(const char *uniform_base, unsigned off)
{
    if (!ASM) return ldg(uniform_base, off);
    double v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(uniform_base));
    return v;
}
// s_waitcnt simm16 of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]; the other two counters left alone
#define LBMPM_WAIT_VMCNT0 0x0F70

template <bool FIRST, bool UNI, typename Rows, bool ASM = false>
__device__ __forceinline__ void pull_q(const RK3Dev &p, const Rows &rows, int zl, unsigned b, double g[Q], unsigned &own_j)
{
    constexpr int OPP[Q] = LBMPM_D3Q19_OPP;
    const PlaneAddrQ a = plane_addr_q(p, p.fin, zl);
    {
        const RowTab t = rows(zl, 0);
        own_j = t.first + bits_below<UNI>(t.m, b);
    }
    const unsigned own8 = own_j * 8u;
    if (FIRST) {
#pragma unroll
        for (int i = 0; i < Q; ++i) g[i] = ldq<ASM>(a.base, a.off[1] + (unsigned)i * a.cnt[1] * 8u + own8);
        return;
    }
#pragma unroll
    for (int rz = -1; rz <= 1; ++rz)
#pragma unroll
        for (int ry = -1; ry <= 1; ++ry) {
            const RowTab t = rows(zl + rz, ry);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int i = DIRT[(1 - rz) * 9 + (1 - ry) * 3 + (1 - dx)];
                if (i < 0) continue;
                unsigned off;
                if (i == 0) off = a.off[1] + own8;
                else {
                    unsigned j;
                    bool fl;
                    row_cell<UNI>(t, dx, b, j, fl);
                    const unsigned up = a.off[1 + rz] + (unsigned)i * a.cnt[1 + rz] * 8u + j * 8u;
                    const unsigned back = a.off[1] + (unsigned)OPP[i] * a.cnt[1] * 8u + own8;
                    off = fl ? up : back;
                }
                g[i] = ldq<ASM>(a.base, off);
            }
        }
}

// Row flags.  One word per row segment (64 cells) and plane, written with the populations (ping-pong like them):
// bit 0 / bit 1 = every fluid cell of the segment is pure red (k_R = 1, A = 0) / pure blue (k_R = 0, A = 0); 3 = no fluid cell.
// A flagged segment's records are neither written nor read -- whoever needs them takes the constant; away from the
// interface a step then moves 19 + 19 doubles per cell.
// (32-bit: (nzl + 2) * ny * nseg row segments of a slab -- 2.1 M at 512^3, checked against 2^31 at set-up)
__device__ __forceinline__ unsigned row_index(const RK3Dev &p, int zl, int y, int sg) { return ((unsigned)zl * (unsigned)p.ny + (unsigned)y) * (unsigned)p.nseg + (unsigned)sg; }

// where the scalar records {k_R, A} of the cells around a node come from
struct LdsScal {                  // the marching kernel's LDS tile (flagged rows are filled with the constant there)
    const double *tile;           // ssc
    int own;                      // index of the node inside a component array of the tile
    int slot[3];                  // index offset of the ring slots that hold the planes zp-1, zp, zp+1
    typedef int H;
    __device__ __forceinline__ H own_cell() const { return slot[1] + own; }
    __device__ __forceinline__ H cell(int rz, int ry, int dx, bool fl, unsigned, unsigned, unsigned) const { return fl ? slot[1 + rz] + own + ry * SC + dx : slot[1] + own; }
    __device__ __forceinline__ double comp(H h, int k) const { return tile[h + k * SCOMP]; }
};
struct GlbScal {                  // straight from global memory (set-up, diagnostics and face kernels)
    const uint32_t *pur_in;       // (field copies, not a reference to the kernel argument: see GlobalRows)
    int ny, nseg;
    const char *base;             // plane_addr_q(...).base
    unsigned soff[3];             // byte offsets of the s arrays of the planes zp-1, zp, zp+1
    unsigned own_j;
    int zp, sg, y;                // plane, row segment and row of the node
    struct H { unsigned off, cst; };      // cst: 0 read the record, else the row flag (bit 0 red)
    __device__ __forceinline__ H at(int rz, int ry, int sgs, unsigned j) const
    {
        int s = sg + sgs;
        s = s < 0 ? s + nseg : (s >= nseg ? s - nseg : s);
        H h;
        h.cst = pur_in[((size_t)(zp + rz) * ny + wrapi(y + ry, ny)) * nseg + s];
        h.off = soff[1 + rz] + j * 32u;
        return h;
    }
    __device__ __forceinline__ H own_cell() const { return at(0, 0, 0, own_j); }
    __device__ __forceinline__ H cell(int rz, int ry, int dx, bool fl, unsigned j, unsigned b, unsigned last) const
    {
        if (!fl) return own_cell();
        return at(rz, ry, (dx < 0 && b == 0u) ? -1 : ((dx > 0 && b == last) ? 1 : 0), j);
    }
    __device__ __forceinline__ double comp(H h, int k) const
    {
        if (h.cst) return k == 0 && (h.cst & 1u) ? 1. : 0.;
        return *reinterpret_cast<const double *>(base + h.off + 8u * (unsigned)k);
    }
};
__device__ __forceinline__ GlbScal glb_scal(const RK3Dev &p, const PlaneAddrQ &a, unsigned own_j, int zp, int sg, int y)
{
    GlbScal s{p.pur_in, p.ny, p.nseg, a.base, {0u, 0u, 0u}, own_j, zp, sg, y};
#pragma unroll
    for (int k = 0; k < 3; ++k) s.soff[k] = a.off[k] + (unsigned)Q * a.cnt[k] * 8u;
    return s;
}

// sums over the pulled populations of a node by the class of c_z (0, +1, -1): k = sum k_R g, a = sum c_i e_i.A (signed for
// bounce-back), t = sum g.  ONE statement of this arithmetic (explicit fma, no contraction) serves the marching kernel, the
// diagnostics and the face kernels of the slab exchange: the slab-decomposed run equals the single-domain run bit for bit.
// With k_R = 1 and A = 0 around a node (pure red) k + a equals t bit for bit, i.e. rho_B = 0 exactly.
struct Sums { double k0, kp, km, a0, ap, am, t0, tp, tm; };

// CLS = c_z of the class: 0, +1 (upstream cells in the plane below), -1 (plane above)
template <int CLS, bool FIRST, bool UNI, typename Rows, typename Scal>
__device__ __forceinline__ void class_sum_one(const RK3Dev &p, const Rows &rows, const Scal &sc, int zp, unsigned b, const double g[Q],
                                              double &ks, double &as, double &ts)
{
#pragma clang fp contract(off)
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    constexpr int rz = -CLS;
    ks = 0.; as = 0.; ts = 0.;
    if (CLS == 0) { ks = sc.comp(sc.own_cell(), 0) * g[0]; ts = g[0]; }
#pragma unroll
    for (int ry = -1; ry <= 1; ++ry) {
        RowTab t{};
        if (!FIRST) t = rows(zp + rz, ry);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int i = DIRT[(1 - rz) * 9 + (1 - ry) * 3 + (1 - dx)];
            if (i <= 0) continue;
            unsigned j = 0;
            bool fl = true;
            typename Scal::H h;
            if (FIRST) h = sc.own_cell();
            else {
                row_cell<UNI>(t, dx, b, j, fl);
                h = sc.cell(rz, ry, dx, fl, j, b, t.last);
            }
            const double k = sc.comp(h, 0);
            double ea = 0.;
            bool any = false;
            if (CX[i] != 0) { const double v = sc.comp(h, 1); ea = CX[i] > 0 ? v : -v; any = true; }
            if (CY[i] != 0) { const double v = sc.comp(h, 2); ea = any ? (CY[i] > 0 ? ea + v : ea - v) : (CY[i] > 0 ? v : -v); any = true; }
            if (CZ[i] != 0) { const double v = sc.comp(h, 3); ea = any ? (CZ[i] > 0 ? ea + v : ea - v) : (CZ[i] > 0 ? v : -v); }
            const double c = i < 7 ? p.rcA : p.rcD;
            const double cs = fl ? c : -c;       // bounce-back: the opposite population of the node itself
            ks = __builtin_fma(k, g[i], ks); as = __builtin_fma(cs, ea, as); ts += g[i];
        }
    }
}

template <bool FIRST, bool UNI, typename Rows, typename Scal>
__device__ __forceinline__ void class_sums(const RK3Dev &p, const Rows &rows, const Scal &sc, int zp, unsigned b, const double g[Q], Sums &S)
{
    class_sum_one<1, FIRST, UNI>(p, rows, sc, zp, b, g, S.kp, S.ap, S.tp);
    class_sum_one<0, FIRST, UNI>(p, rows, sc, zp, b, g, S.k0, S.a0, S.t0);
    class_sum_one<-1, FIRST, UNI>(p, rows, sc, zp, b, g, S.km, S.am, S.tm);
}

// the pulls of one class only (face kernels of the slab exchange: the other classes belong to the neighbour rank)
template <int CLS, bool FIRST, typename Rows>
__device__ __forceinline__ void pull_class(const RK3Dev &p, const Rows &rows, int zl, unsigned b, unsigned own_j, double g[Q])
{
    constexpr int OPP[Q] = LBMPM_D3Q19_OPP;
    constexpr int rz = -CLS;
    const PlaneAddrQ a = plane_addr_q(p, p.fin, zl);
    const unsigned own8 = own_j * 8u;
    if (CLS == 0) g[0] = ldg(a.base, a.off[1] + own8);
#pragma unroll
    for (int ry = -1; ry <= 1; ++ry) {
        RowTab t{};
        if (!FIRST) t = rows(zl + rz, ry);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int i = DIRT[(1 - rz) * 9 + (1 - ry) * 3 + (1 - dx)];
            if (i <= 0) continue;
            unsigned j = 0;
            bool fl = true;
            if (!FIRST) row_cell<false>(t, dx, b, j, fl);
            const unsigned up = a.off[1 + rz] + (unsigned)i * a.cnt[1 + rz] * 8u + j * 8u;
            const unsigned back = a.off[1] + (unsigned)OPP[i] * a.cnt[1] * 8u + own8;
            g[i] = ldg(a.base, FIRST ? a.off[1] + (unsigned)i * a.cnt[1] * 8u + own8 : (fl ? up : back));
        }
    }
}

// class sums of a node whose pulled directions all come from cells of ONE colour (k_R = 1 or 0 and A = 0 at every upstream
// cell): k = t or 0 and a = 0, bit for bit what class_sums returns there (fma(1, g, s) = s + g; c * 0 adds +0).  The marching
// kernel keeps one flag per tile row and plane; away from the interface -- most of a two-phase lattice -- the 49 LDS reads
// and ~200 instructions of class_sums, and the pulls of the rim cells altogether, are skipped.
__device__ __forceinline__ void class_sums_pure(const double g[Q], bool red, Sums &S)
{
#pragma clang fp contract(off)
    constexpr int CZ[Q] = LBMPM_D3Q19_CZ;
    S.t0 = g[0]; S.tp = 0.; S.tm = 0.;
#pragma unroll
    for (int rz = -1; rz <= 1; ++rz)
#pragma unroll
        for (int ry = -1; ry <= 1; ++ry)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int i = DIRT[(1 - rz) * 9 + (1 - ry) * 3 + (1 - dx)];
                if (i <= 0) continue;
                if (CZ[i] == 0) S.t0 += g[i];
                else if (CZ[i] > 0) S.tp += g[i];
                else S.tm += g[i];
            }
    S.k0 = red ? S.t0 : 0.; S.kp = red ? S.tp : 0.; S.km = red ? S.tm : 0.;
    S.a0 = S.ap = S.am = 0.;
}

// densities of the streamed, boundary-corrected node on plane zl from its class sums; on the inlet / outlet plane pairs
// the per-colour Zou-He closures (zouhe_inlet / zouhe_outlet above, summed over the colours: they are linear in the
// populations once rho_c is known) rewrite the unknown colour-blind populations in g
// PIN: 0 velocity inlet, 1 pressure inlet, -1 decided at run time (p.inletP).  The marching kernel knows it at compile time: the
// run-time test cost rk3dq_fused two registers and 1.2 % (A/B, round 6)
template <bool WITH_G, int PIN = -1>
__device__ __forceinline__ void bc_q(const RK3Dev &p, int zl, const Sums &S, double *g, double &rR, double &rho)
{
#pragma clang fp contract(off)
    const bool pin = PIN < 0 ? p.inletP != 0 : PIN == 1;
    const int zsg = p.z0 + source_plane(p, zl) - 1;
    const double r0 = S.k0 + S.a0, rp = S.kp + S.ap, rm = S.km + S.am;
    rR = (r0 + rp) + rm;
    rho = (S.t0 + S.tp) + S.tm;
    if (zsg == p.nzg - 2) {                    // velocity inlet: unknown e_z = -1
        const double wR = r0 + 2. * rp, wT = S.t0 + 2. * S.tp;
        double J;
        if (pin) {                             // pressure inlet per colour: rho_c u_c = (S0 + 2 S+)_c - rho_c, summed over the colours
            rR = p.vzR;                        // (densityRH, densityBH: RK3Dev::inletP)
            rho = p.vzR + p.vzB;
            J = wT - rho;
        } else {
            const double dR = wR / (1. + p.vzR), dB = (wT - wR) / (1. + p.vzB);
            rR = dR;
            rho = dR + dB;
            J = dR * p.vzR + dB * p.vzB;
        }
        if (WITH_G) {
            const double Nx = 0.5 * ((g[1] + g[7] + g[9]) - (g[2] + g[8] + g[10]));
            const double Ny = 0.5 * ((g[3] + g[7] + g[10]) - (g[4] + g[8] + g[9]));
            g[6] = g[5] - 1. / 3. * J;
            g[12] = g[11] - 1. / 6. * J + Nx;
            g[13] = g[14] - 1. / 6. * J - Nx;
            g[16] = g[15] - 1. / 6. * J + Ny;
            g[17] = g[18] - 1. / 6. * J - Ny;
        }
    }
    if (zsg == 1) {                            // pressure outlet: unknown e_z = +1; rho_c u_c = rho_c - (S0 + 2 S-)_c
        rR = p.rhoOutR;
        rho = p.rhoOutR + p.rhoOutB;
        if (WITH_G) {
            const double J = rho - (S.t0 + 2. * S.tm);
            const double Nx = 0.5 * ((g[1] + g[7] + g[9]) - (g[2] + g[8] + g[10]));
            const double Ny = 0.5 * ((g[3] + g[7] + g[10]) - (g[4] + g[8] + g[9]));
            g[5] = g[6] + 1. / 3. * J;
            g[11] = g[12] + 1. / 6. * J - Nx;
            g[14] = g[13] + 1. / 6. * J + Nx;
            g[15] = g[16] + 1. / 6. * J - Ny;
            g[18] = g[17] + 1. / 6. * J + Ny;
        }
    }
}
__device__ __forceinline__ double phi_q(double rR, double rho) { return (rR - (rho - rR)) / rho; }

// ---------------------------------------------------------------------------------------------- the marching kernel
// Structure of rk3dc_fused (64 x 8 tile, one block per CU marching along z, pulls two planes ahead, phase field of tile +
// 1-cell rim in a four-plane LDS ring, one barrier per plane), with
//   * the scalar records of tile + 2 cells of halo in a four-plane LDS ring (104 KB): every pulled direction of the tile's
//     and of the rim's cells finds its upstream k_R and A there,
//   * no LDS park: the 19 pulled values of the plane that waits for its neighbours' phase field stay in registers
//     (38 VGPRs where the 38-value kernel carried 76 in flight).
// RAGGED: nx is not a multiple of 64 -- a tile's rows hold w = 32 .. 64 (or all nx < 64) lattice cells from x0 = tx nx / tilesX on
// (seg_x0); the lanes behind them are idle (or write line padding), the tile's right rim column and record halo sit at tile columns
// w and w + 1, and the segment to the left ends at its bit wl - 1.  All of that lives in the per-thread geometry words; the march
// step differs by where it takes a rim column's bit from.  RAGGED = false is the kernel as it was (64-cell segments, constants).
template <bool FIRST, bool MRT, bool RAGGED, bool PIN = false>       // PIN: [BoundaryCondition] BoundaryTypeInlet = 'Dirichlet' (bc_q)
__global__ __launch_bounds__(512, 1) void rk3dq_fused(RK3Dev p, int tilesX, int tilesY, int rows_per_xcd, int chunk_len, int z_first, int z_last,
                                                       int nchunks1, int z_first2, int z_last2,      // a second range of planes in the same launch
                                                       unsigned *slotq)                              // eight zeroed counters of this launch
{
    constexpr int TX = 64, TY = 8;
    using M = March<TX, TY>;
    using TR = TileRows<TY>;
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    __shared__ double sphi[M::RING][M::FY][M::FX];
    __shared__ double ssc[4 * SSLOT];              // [ring slot][k_R, A_x, A_y, A_z][SR][SC]
    __shared__ u32x4 srow[TR::SLOTS][TR::ROWS][6];
    // Which tile?  Every XCD owns a band of tile rows (its L2 then serves the rim rows and row records its workgroups share).  "Workgroup b
    // runs on XCD b % 8" is only an observation, and it does not hold for a launch that follows another queue's launch (the slab step:
    // per-workgroup time stamps showed the eight "bands" running at 440 .. 560 us instead of 405 each, 7.6 instead of 6.2 us per march
    // step).  So a workgroup asks the hardware where it is and takes the next tile of THAT XCD's band from a counter; a band that is
    // exhausted (the dispatcher gave this XCD more workgroups than its share) sends it to the next band with tiles left.
    __shared__ int sslot[2];
    const int tid = threadIdx.x;
    const int band_slots = (int)(gridDim.x >> 3);
    if (tid == 0 && !slotq) { sslot[0] = (int)(blockIdx.x & 7u); sslot[1] = (int)(blockIdx.x >> 3); }      // LBMPM_RK3D_XCC=0: by block index
    if (tid == 0 && slotq) {
        const int here = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);          // HW_REG_XCC_ID, bits 3:0
        int band = -1, slot = 0;
        for (int k = 0; k < 8 && band < 0; ++k) {
            const int b = (here + k) & 7;
            const int q = (int)atomicAdd(&slotq[b], 1u);
            if (q < band_slots) { band = b; slot = q; }
        }
        sslot[0] = band; sslot[1] = slot;
    }
    __syncthreads();
    const int xcd = sslot[0], slot = sslot[1];
    if (xcd < 0) return;
#ifdef LBMPM_DEV
    const int bid = xcd + 8 * slot;                     // (the time-stamp trace is indexed by it)
#endif
    const int tx = slot % tilesX, r = slot / tilesX, ty = xcd * rows_per_xcd + r % rows_per_xcd, chunk = r / rows_per_xcd;
    if (ty >= tilesY) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: branches on it are branches, not exec masks
#ifdef LBMPM_DEV
    // LBMPM_RK3D_TRACE: per-workgroup time stamps (dev tool); launches of fewer than 8 planes (a slab's boundary ranges) leave no stamps
    unsigned long long *trace = p.trace && z_last - z_first >= 8 ? p.trace + (size_t)bid * 4 : nullptr;
    if (trace && tid == 0) trace[0] = wall_clock64();
#endif
    // Everything a thread derives from its id -- tile coordinates, its rim cell (bottom row, top row: one wave each; then the two
    // columns, corners included), its second entry of the scalar-tile fill (waves 3..6 take the tile rows -2, -1, 8, 9, wave 7 the
    // 4 x 12 cells left and right) -- is RE-DERIVED at the top of every march step from an id the compiler cannot see through:
    // hoisted out of the loop these ~30 values stay in registers for good, the kernel spills, and every spill reload is followed by
    // s_waitcnt vmcnt(0).  ~60 integer instructions per step buy the registers back.
    int lx, ly, x, y, yo, hlx, hly, hx, hy, xrow, xc, xcol;
    bool own, has_own, has_rim, has_x, row_ok;
    TileRowsU<TY, true, RAGGED> rows_own{{srow, 0, 1}}, rows_rimrow{{srow, 0, 1}}, rows_xrow{{srow, 0, 1}};
    TileRowsU<TY, false, RAGGED> rows_rimcol{{srow, 0, 0}}, rows_xcol{{srow, 0, 0}};
    // first lattice column of the tile, its width, the width of the tile to its left (periodic in x): wave-uniform, fixed for the kernel
    const int x0 = RAGGED ? seg_x0(tx, p.nx, tilesX) : tx * TX;
    const int tw = RAGGED ? seg_x0(tx + 1, p.nx, tilesX) - x0 : TX;
    const int twl = RAGGED ? (tx > 0 ? x0 - seg_x0(tx - 1, p.nx, tilesX) : p.nx - seg_x0(tilesX - 1, p.nx, tilesX)) : TX;
    auto set_geometry = [&](int t) {
        lx = t % TX; ly = t / TX;
        x = x0 + lx; y = ty * TY + ly;
        own = y < p.ny;                                  // (lanes behind a ragged row's last cell stay: they write line padding)
        yo = ring_coord(y, p.ny);
        row_ok = yo >= 0;
        has_own = row_ok && lx < tw;
        hlx = 0; hly = 0; hx = 0; hy = 0;
        has_rim = false;
        if (t < M::NH) {
            bool in_row = true;
            if (t < 2 * TX) { hlx = 1 + t % TX; hly = t < TX ? 0 : M::FY - 1; in_row = t % TX < tw; }
            else { const int k = t - 2 * TX; hlx = k < M::FY ? 0 : tw + 1; hly = k < M::FY ? k : k - M::FY; }
            hx = ring_coord(x0 + hlx - 1, p.nx);
            hy = ring_coord(ty * TY + hly - 1, p.ny);
            has_rim = hx >= 0 && hy >= 0 && in_row;
        }
        xrow = ly >= 3 && ly <= 6 ? (ly < 5 ? ly - 5 : ly + 3) : (lx >> 2) - 2;            // tile row of the second entry
        xc = lx & 3; xcol = ly == 7 ? (xc < 2 ? xc - 2 : tw + xc - 2) : lx;               // its tile column
        has_x = (ly >= 3 && ly <= 6) || (ly == 7 && lx < 4 * SR);
        rows_own.t.lrow = ly + 2;
        rows_rimrow.t.lrow = hly + 1;
        rows_rimcol.t.lrow = hly + 1; rows_rimcol.t.k = hlx == 0 ? 0 : 2;
        rows_xrow.t.lrow = xrow + 2;
        rows_xcol.t.lrow = xrow + 2; rows_xcol.t.k = xc < 2 ? 0 : 2;
    };
    set_geometry(tid);
    // ... re-derived from three packed words per thread in LDS since round 5: set_geometry costs ~ 80 vector instructions per march step, the
    // unpacking ~ 30 (the words are read through an address the compiler cannot see through, so nothing is hoisted here either)
    __shared__ unsigned sgeo[3][TX * TY];
    sgeo[0][tid] = (unsigned)lx | ((unsigned)ly << 6) | ((unsigned)hlx << 9) | ((unsigned)hly << 16) | ((unsigned)(xrow + 2) << 20) | ((unsigned)xc << 24) |
                   ((unsigned)own << 26) | ((unsigned)has_own << 27) | ((unsigned)has_rim << 28) | ((unsigned)has_x << 29) | ((unsigned)row_ok << 30);
    sgeo[1][tid] = ((unsigned)hx & 0xffffu) | ((unsigned)hy << 16);
    sgeo[2][tid] = ((unsigned)yo & 0xffffu) | ((unsigned)(xcol + 2) << 16);
    auto load_geometry = [&](int t) {
        const unsigned w0 = sgeo[0][t], w1 = sgeo[1][t], w2 = sgeo[2][t];
        lx = (int)(w0 & 63u); ly = (int)((w0 >> 6) & 7u);
        x = x0 + lx; y = ty * TY + ly;
        hlx = (int)((w0 >> 9) & 127u); hly = (int)((w0 >> 16) & 15u);
        xrow = (int)((w0 >> 20) & 15u) - 2; xc = (int)((w0 >> 24) & 3u);
        own = (w0 >> 26) & 1u; has_own = (w0 >> 27) & 1u; has_rim = (w0 >> 28) & 1u; has_x = (w0 >> 29) & 1u;
        row_ok = RAGGED ? (w0 >> 30) & 1u : has_own;
        hx = (int)(short)(w1 & 0xffffu); hy = (int)w1 >> 16;
        yo = (int)(short)(w2 & 0xffffu); xcol = (int)(w2 >> 16) - 2;
        rows_own.t.lrow = ly + 2;
        rows_rimrow.t.lrow = hly + 1;
        rows_rimcol.t.lrow = hly + 1; rows_rimcol.t.k = hlx == 0 ? 0 : 2;
        rows_xrow.t.lrow = xrow + 2;
        rows_xcol.t.lrow = xrow + 2; rows_xcol.t.k = xc < 2 ? 0 : 2;
    };
    // chunks 0 .. nchunks1-1 march the planes z_first .. z_last, the chunks behind them z_first2 .. z_last2 (the two boundary ranges of a slab
    // in one launch)
    const int za = chunk < nchunks1 ? z_first + chunk * chunk_len : z_first2 + (chunk - nchunks1) * chunk_len;
    const int zb = min(za + chunk_len - 1, chunk < nchunks1 ? z_last : z_last2);
    const int zl_gb = 1 - p.z0, zl_gt = p.nzg - p.z0;          // local index of the ghost planes z = 0 and z = nz-1 (when owned)
    auto is_ghost = [&](int zl) { return zl == zl_gb || zl == zl_gt; };
    // one record per lane (72 lanes) + the row flags that ride in its spare words: .z this segment's, .w (centre record) the three
    // segments' combined.  fetch_rows only LOADS (its results are consumed a march step later, no wait in between); put_rows combines
    struct RowRec { u32x4 v; unsigned f0, f1, f2; };
    auto fetch_rows = [&](int zl) -> RowRec {
        RowRec q;
        q.v = u32x4{0u, 0u, 0u, 0u}; q.f0 = 0u; q.f1 = ~0u; q.f2 = ~0u;
        if (tid < TR::ROWS * 6 && zl >= 0 && zl <= p.nzl + 1) {
            const int row = tid / 6, k = tid % 6;
            int yy = (ty * TY - 2 + row) % p.ny;
            if (yy < 0) yy += p.ny;
            int sg = tx - 1 + (k >> 1);
            sg = sg < 0 ? sg + p.nseg : (sg >= p.nseg ? sg - p.nseg : sg);
            const unsigned rr = ((unsigned)zl * (unsigned)p.ny + (unsigned)yy) * (unsigned)p.nseg + (unsigned)sg;
            q.v = (k & 1) ? p.seg2[rr] : p.seg[rr];
            if (k & 1) {
                q.f0 = p.pur_in[rr];
                if (k == 3) {
                    const int sl = tx > 0 ? tx - 1 : p.nseg - 1, sr = tx + 1 < p.nseg ? tx + 1 : 0;
                    q.f1 = p.pur_in[rr - sg + sl]; q.f2 = p.pur_in[rr - sg + sr];
                }
            }
        }
        return q;
    };
    auto put_rows = [&](int zl, const RowRec &q) {
        if (tid < TR::ROWS * 6) {
            u32x4 v = q.v;
            if (tid & 1) { v.z = q.f0; v.w = q.f0 & q.f1 & q.f2; }       // k = tid % 6 is odd exactly when tid is
            const_cast<u32x4 &>(srow[zl & (TR::SLOTS - 1)][tid / 6][tid % 6]) = v;
        }
    };
    auto row_flag = [&](int zl, int lrow, int k) -> unsigned { return srow[zl & (TR::SLOTS - 1)][lrow][2 * k + 1].z; };   // segment k = 0, 1, 2
    // scalar records of plane zl -> registers (two entries per thread at most) -> LDS tile
    struct SRec { double2 a, b; bool ok; };
    auto load_s = [&](int zl, unsigned j) -> SRec {
        const unsigned long long p0 = pstart_of(p, zl);
        const unsigned cnt = (unsigned)(pstart_of(p, zl + 1) - p0);
        const char *s = reinterpret_cast<const char *>(p.fin) + (size_t)p0 * CELLB + (size_t)cnt * (Q * 8u) + (size_t)j * 32u;
        SRec v;
        v.a = *reinterpret_cast<const double2 *>(s); v.b = *reinterpret_cast<const double2 *>(s + 16);
        v.ok = true;
        return v;
    };
    auto const_s = [&](unsigned flag) -> SRec {     // the record of every cell of a flagged segment
        SRec v;
        v.a.x = (flag & 1u) ? 1. : 0.; v.a.y = 0.; v.b.x = 0.; v.b.y = 0.; v.ok = true;
        return v;
    };
    auto fetch_s = [&](int zl, SRec &e0, SRec &e1) {
        e0.ok = false; e1.ok = false;
        if (zl < 0 || zl > p.nzl + 1) return;
        {
            const RowTab t = rows_own(zl, 0);
            const unsigned fg = (unsigned)__builtin_amdgcn_readfirstlane((int)row_flag(zl, ly + 2, 1));
            if (bit_of<true>(t.m, (unsigned)lx)) e0 = fg ? const_s(fg) : load_s(zl, t.first + bits_below<true>(t.m, (unsigned)lx));
        }
        if (wave >= 3 && wave <= 6) {
            const RowTab t = rows_xrow(zl, 0);
            const unsigned fg = (unsigned)__builtin_amdgcn_readfirstlane((int)row_flag(zl, xrow + 2, 1));
            if (bit_of<true>(t.m, (unsigned)lx)) e1 = fg ? const_s(fg) : load_s(zl, t.first + bits_below<true>(t.m, (unsigned)lx));
        } else if (wave == 7) {
            if (has_x) {
                const RowTab t = rows_xcol(zl, 0);
                const unsigned fg = row_flag(zl, xrow + 2, xc < 2 ? 0 : 2);
                const unsigned bb = xc < 2 ? (unsigned)(twl - 2) + (unsigned)xc : (unsigned)xc - 2u;      // the left segment's last two cells / the right one's first two
                if (bit_of<false>(t.m, bb)) e1 = fg ? const_s(fg) : load_s(zl, t.first + bits_below<false>(t.m, bb));
            }
        }
    };
    auto put_s = [&](int zl, const SRec &e0, const SRec &e1) {
        double *sl = ssc + (zl & 3) * SSLOT;
        if (e0.ok) {
            double *d = sl + (ly + 2) * SC + lx + 2;
            d[0] = e0.a.x; d[SCOMP] = e0.a.y; d[2 * SCOMP] = e0.b.x; d[3 * SCOMP] = e0.b.y;
        }
        if (wave >= 3 && e1.ok) {
            double *d = sl + (xrow + 2) * SC + xcol + 2;
            d[0] = e1.a.x; d[SCOMP] = e1.a.y; d[2 * SCOMP] = e1.b.x; d[3 * SCOMP] = e1.b.y;
        }
    };
    // colour of everything the cells of the tile rows row_lo .. row_hi of plane zp pull from: 1 red, 2 blue, 0 mixed
    auto purity = [&](int zp, int row_lo, int row_hi) -> int {
        unsigned c = 3u;
        for (int dz = -1; dz <= 1; ++dz)
            for (int rr = row_lo - 1; rr <= row_hi + 1; ++rr) c &= srow[(zp + dz) & (TR::SLOTS - 1)][rr + 2][3].w;
        return __builtin_amdgcn_readfirstlane((int)c);
    };
    auto lds_scal = [&](int zp, int row, int col) {       // accessor for the node at tile coordinates (col, row) pulled around plane zp
        LdsScal s;
        s.tile = ssc; s.own = (row + 2) * SC + col + 2;
        s.slot[0] = ((zp - 1) & 3) * SSLOT; s.slot[1] = (zp & 3) * SSLOT; s.slot[2] = ((zp + 1) & 3) * SSLOT;
        return s;
    };
    // prologue: the six planes of row records, then the three planes of scalar records, each as ONE batch of loads (fetched and put plane
    // by plane they were nine dependent memory round trips per workgroup -- a third of a slab's boundary launch, whose workgroups march
    // two planes each)
    {
        RowRec rr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) rr[k] = fetch_rows(za - 3 + k);
#pragma unroll
        for (int k = 0; k < 6; ++k) put_rows(za - 3 + k, rr[k]);
    }
    RowRec staged = fetch_rows(za + 3);
    __syncthreads();
    {
        SRec e0[3], e1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) fetch_s(za - 2 + k, e0[k], e1[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) put_s(za - 2 + k, e0[k], e1[k]);
    }
    bool fluid = false, fl_raw = false;         // node of the plane that waits / of the plane in flight is fluid
    bool padz = false, pad_raw = false;         // idle lane that writes line padding for that plane
    unsigned jz = 0, j_raw = 0;                 // their j
    double rR = 1., rho = 2.;                   // densities of the waiting plane
    bool mixed_w = false, mixed_n = false;      // the waiting / landed plane's row took the general class sums (wave-uniform)
    double raw[Q], cur[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) { raw[i] = 0.; cur[i] = 0.; }
    auto issue = [&](int zl) {                  // pulls of the own cell of plane zl
        fl_raw = false; pad_raw = false;
        if (!row_ok || zl < 1 || zl > p.nzl || is_ghost(zl)) return;
        const RowTab t = rows_own(zl, 0);
        fl_raw = bit_of<true>(t.m, (unsigned)lx);
        if (!fl_raw) {
            const unsigned rank = bits_below<true>(~t.m, (unsigned)lx);
            pad_raw = own && rank < t.pad;
            j_raw = t.first + (unsigned)__popcll(t.m) + rank;
            return;
        }
        pull_q<FIRST, true, decltype(rows_own), true>(p, rows_own, zl, (unsigned)lx, raw, j_raw);
    };
    // The own-cell pulls are waited for BY HAND.  vmcnt counts loads and stores alike and retires them in order; hipcc places its
    // waits per register with the count of the path that issued the fewest younger operations, so with the pulls in its books it
    // drains them right after the barrier (a load "pending" on some skipped branch) and drains the collision's stores at the loop
    // latch (the path without a collision has no stores).  Measured: pulls-only + stores-only = the fused time, nothing overlapped.
    // Schedule of a march step:   <everything hipcc knows about is complete: s_waitcnt vmcnt(0), stated with the builtin so that
    // its books are empty too>  ->  pulls of plane z + 2 (asm)  ->  row records of z + 6 (hipcc's; consumed after the next wait)
    // ->  barrier  ->  collision arithmetic of plane z  ->  s_waitcnt vmcnt(0) (builtin: the pulls have had the whole arithmetic
    // to arrive, no store is outstanding)  ->  the collision's stores, which nobody waits for until the next step's first line.
    auto pulls_landed = [&]() {
        __builtin_amdgcn_s_waitcnt(LBMPM_WAIT_VMCNT0);
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("" : "+v"(raw[i]));       // nothing that reads raw moves above the wait
    };
    __syncthreads();
    issue(za - 1);
    pulls_landed();
#ifdef LBMPM_DEV
    if (trace && tid == 0) { trace[1] = wall_clock64(); trace[3] = ((unsigned long long)(unsigned)za << 32) | (unsigned)zb; }
#endif

    // -DLBMPM_DEV -DLBMPM_PHASES (tools/dev/phases.py): cycle counter at six marks of the march step, summed per wave
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES)
    unsigned long long ph_acc[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0;
#define PH_MARK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); if (z >= za) ph_acc[k] += now_ - ph_t; ph_t = now_; }
#else
#define PH_MARK(k)
#endif
    for (int z = za - 2; z <= zb; ++z) {
        const int zn = z + 1;
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES)
        ph_t = __builtin_readcyclecounter();
#endif
        {
            int t = tid;
            asm volatile("" : "+v"(t));
#ifdef LBMPM_RK3D_GEO_COMPUTE
            set_geometry(t);
#else
            load_geometry(t);
#endif
        }
        // fetched here, in uniform control flow, as scalar loads: inside the collision's branch they become a vector load, and the
        // s_waitcnt vmcnt(0) in front of its use drains the pulls in flight (the collision then overlaps nothing)
        const unsigned long long pz0 = pstart_of(p, z > 0 ? z : 0), pz1 = pstart_of(p, z > 0 ? z + 1 : 1);
        put_rows(z + 5, staged);
        SRec e0, e1;
        fetch_s(z + 3, e0, e1);                 // in LDS before this step's barrier, read from the next step on
        const bool halo_n = zn <= p.halo_lo || zn >= p.nzl + 1, ghost_n = !halo_n && is_ghost(zn);
        const bool fill_gb = !halo_n && !ghost_n && zn - 1 == zl_gb && zn - 1 >= 1;     // the bottom ghost's phase field = this plane's
        // ---- plane z + 1, rim cells: phase field only
        if (wave < 3) {
            if (has_rim) {
                double ph = p.solidPhi;
                if (halo_n) ph = (p.phi + (size_t)zn * p.plane2)[(size_t)hy * p.pitch + hx];
                else if (ghost_n) { if (zn == zl_gt) ph = sphi[z & (M::RING - 1)][hly][hlx]; }
                else {
                    double g[Q], a, c;
                    unsigned j;
                    Sums S;
                    const int pc = wave < 2 ? purity(zn, hly - 1, hly - 1) : purity(zn, -1, TY);
                    const unsigned hb = RAGGED ? (hlx == 0 ? (unsigned)(twl - 1) : 0u) : (unsigned)(hx & 63);      // a rim column's bit in its segment
                    const bool fl = wave < 2 ? bit_of<true>(rows_rimrow(zn, 0).m, (unsigned)lx) : bit_of<false>(rows_rimcol(zn, 0).m, hb);
                    if (pc != 0) {          // single colour around: phi = +-1 (or the planes' boundary values) without a pull
                        if (fl) {
                            S.t0 = 1.; S.tp = S.tm = 0.; S.k0 = (pc & 1) ? 1. : 0.; S.kp = S.km = S.a0 = S.ap = S.am = 0.;
                            bc_q<false, PIN ? 1 : 0>(p, zn, S, nullptr, a, c);
                            ph = phi_q(a, c);
                        }
                    } else if (wave < 2) {
                        if (fl) {
                            pull_q<FIRST, true>(p, rows_rimrow, zn, (unsigned)lx, g, j);
                            class_sums<FIRST, true>(p, rows_rimrow, lds_scal(zn, hly - 1, hlx - 1), zn, (unsigned)lx, g, S);
                            bc_q<false, PIN ? 1 : 0>(p, zn, S, nullptr, a, c);
                            ph = phi_q(a, c);
                        }
                    } else if (fl) {
                        pull_q<FIRST, false>(p, rows_rimcol, zn, hb, g, j);
                        class_sums<FIRST, false>(p, rows_rimcol, lds_scal(zn, hly - 1, hlx - 1), zn, hb, g, S);
                        bc_q<false, PIN ? 1 : 0>(p, zn, S, nullptr, a, c);
                        ph = phi_q(a, c);
                    }
                }
                sphi[zn & (M::RING - 1)][hly][hlx] = ph;
                if (fill_gb) sphi[(zn - 1) & (M::RING - 1)][hly][hlx] = ph;
            }
        }
        PH_MARK(0)
        // ---- plane z + 1, own cell (pulled during the previous march step): class sums from the LDS records, boundary
        //      rules, phase field into the ring; the plane that waited (z) moves on to its collision
        double ft[Q];
        const double rRz = rR, rhoz = rho;
        const unsigned jzz = jz;
        const bool mixed_z = mixed_w;
        mixed_n = false;
        const bool fluidn = fl_raw && !halo_n && !ghost_n;
        {
            double rRn = 1., rhon = 2., ph = p.solidPhi;
            if (halo_n) { if (has_own) ph = (p.phi + (size_t)zn * p.plane2)[(size_t)yo * p.pitch + x]; }
            else if (ghost_n) { if (has_own && zn == zl_gt) ph = sphi[z & (M::RING - 1)][ly + 1][lx + 1]; }
            else if (fl_raw) {
                Sums S;
                const int pc = purity(zn, ly, ly);
                mixed_n = pc == 0;
                if (pc != 0) class_sums_pure(raw, (pc & 1) != 0, S);
                else class_sums<FIRST, true>(p, rows_own, lds_scal(zn, ly, lx), zn, (unsigned)lx, raw, S);
                bc_q<true, PIN ? 1 : 0>(p, zn, S, raw, rRn, rhon);
                ph = phi_q(rRn, rhon);
            }
            if (has_own) {
                sphi[zn & (M::RING - 1)][ly + 1][lx + 1] = ph;
                if (fill_gb) sphi[(zn - 1) & (M::RING - 1)][ly + 1][lx + 1] = ph;
            }
#pragma unroll
            for (int i = 0; i < Q; ++i) { ft[i] = cur[i]; cur[i] = raw[i]; }
            rR = rRn; rho = rhon; jz = j_raw; mixed_w = mixed_n;
        }
        const bool padzz = padz;
        padz = pad_raw;
        PH_MARK(1)
        // ---- pulls of plane z + 2 into flight
        put_s(z + 3, e0, e1);
        __builtin_amdgcn_s_waitcnt(LBMPM_WAIT_VMCNT0);
        PH_MARK(2)
        if (z + 2 <= zb + 1) issue(z + 2);
        else { fl_raw = false; pad_raw = false; }
        staged = fetch_rows(z + 6);
        PH_MARK(3)
        __syncthreads();
        PH_MARK(4)
        // ---- plane z: collide.  Fluid cells only go through the collision (no zero selects in it); the idle lanes that complete the
        //      last line of the tile's run (padzz) store zeros afterwards
        const bool live = z >= za && !is_ghost(z);
        const bool active = live && fluid && own;
        if (__ballot(active) == 0ull) {
            pulls_landed();       // a wave without a collision: its pulls still have to land before raw moves on
            // a row segment without a fluid cell: flag 3 ("nothing here"), so that its neighbours' rows can still be taken as single-colour
            if (live && own && lx == 0) p.pur_out[row_index(p, z, y, tx)] = 3u;
        } else if (active) {
            LBMPM_TAKEN;          // (the ballot above: a lane of this wave is here -- every path of the step passes a wait for the pulls)
            // colour gradient 3 sum_i w_i e_i phi(x + e_i): the axis and the diagonal neighbours summed separately (signed), one
            // product per weight class and component
            double ax = 0., ay = 0., az = 0., dx = 0., dy = 0., dz = 0.;
#pragma unroll
            for (int i = 1; i < Q; ++i) {
                const double ph = sphi[(z + CZ[i]) & (M::RING - 1)][ly + 1 + CY[i]][lx + 1 + CX[i]];
                if (i < 7) {
                    if (CX[i] != 0) ax = CX[i] > 0 ? ax + ph : ax - ph;
                    if (CY[i] != 0) ay = CY[i] > 0 ? ay + ph : ay - ph;
                    if (CZ[i] != 0) az = CZ[i] > 0 ? az + ph : az - ph;
                } else {
                    if (CX[i] != 0) dx = CX[i] > 0 ? dx + ph : dx - ph;
                    if (CY[i] != 0) dy = CY[i] > 0 ? dy + ph : dy - ph;
                    if (CZ[i] != 0) dz = CZ[i] > 0 ? dz + ph : dz - ph;
                }
            }
            const double gx = 3. * wq(1) * ax + 3. * wq(7) * dx, gy = 3. * wq(1) * ay + 3. * wq(7) * dy, gz = 3. * wq(1) * az + 3. * wq(7) * dz;
            if (__builtin_amdgcn_readfirstlane((int)mixed_z))
                collide_store<2, MRT, true>(p, reinterpret_cast<char *>(p.fout) + (size_t)pz0 * CELLB, (unsigned)(pz1 - pz0) * 8u, jzz * 8u, true, ft, rRz,
                                            rhoz - rRz, gx, gy, gz, p.pur_out + row_index(p, z, y, tx));
            else
                collide_store<2, MRT>(p, reinterpret_cast<char *>(p.fout) + (size_t)pz0 * CELLB, (unsigned)(pz1 - pz0) * 8u, jzz * 8u, true, ft, rRz, rhoz - rRz,
                                      gx, gy, gz, p.pur_out + row_index(p, z, y, tx));
        }
        if (live && padzz) {
            char *pl = reinterpret_cast<char *>(p.fout) + (size_t)pz0 * CELLB;
            const unsigned stride = (unsigned)(pz1 - pz0) * 8u;
#pragma unroll
            for (int i = 0; i < Q; ++i) stg(pl + (size_t)i * stride, jzz * 8u, 0.);
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) asm volatile("" : "+v"(raw[i]));
        fluid = fluidn;
        PH_MARK(5)
    }
#undef PH_MARK
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES)
    if (p.trace && (tid & 63) == 0 && bid < 4096) {          // second half of the 4 MB trace area: [workgroup][wave][8]
        unsigned long long *o = p.trace + (1u << 18) + ((size_t)bid * 8 + (size_t)wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = ph_acc[k];
        o[6] = (unsigned long long)(zb - za + 1);
    }
#endif
#ifdef LBMPM_DEV
    if (trace && tid == 0) trace[2] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------- set-up and diagnostics
// f = w rho at rest (3-D analogue of RKD2Q9.py:577-601): g_i = w_i (rho_R + rho_B), k_R = rho_R / rho, A = 0
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_init_rest(RK3Dev p, const double *rho_r, const double *rho_b, double *f, uint32_t *pur)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;    // a wave = a row segment
    if (y >= p.ny) return;
    const bool fluid = x >= 0 && (p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1);
    double a = 0., b = 0.;
    if (fluid) {
        const GlobalRows rows{p, sg, y};
        const RowTab t = rows(zl, 0);
        const unsigned j = t.first + bits_below<false>(t.m, threadIdx.x);
        const size_t sd = ((size_t)(zl - 1) * p.ny + y) * p.nx + x;
        a = rho_r[sd]; b = rho_b[sd];
        const unsigned long long p0 = pstart_of(p, zl);
        const size_t cnt = (size_t)(pstart_of(p, zl + 1) - p0);
        double *pl = f + (size_t)p0 * QS;
        for (int i = 0; i < Q; ++i) pl[(size_t)i * cnt + j] = wq(i) * (a + b);
        double *s = pl + (size_t)Q * cnt + (size_t)j * 4;
        s[0] = a / (a + b); s[1] = 0.; s[2] = 0.; s[3] = 0.;
    }
    const unsigned code = (__ballot(fluid && b != 0.) == 0ull ? 1u : 0u) | (__ballot(fluid && a != 0.) == 0ull ? 2u : 0u);
    if (threadIdx.x == 0) pur[row_index(p, zl, y, sg)] = code;
}

// phase field (and rho_R, rho_B, u with diag) of the streamed, boundary-corrected lattice on the planes zl0.. (diagnostics
// and the planes a neighbour rank needs); ghost planes are pulled around their source plane like rk3dc_phase_field
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_phase_field(RK3Dev p, int zl0)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + zl0;
    if (y >= p.ny || x < 0) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const int zs = source_plane_x(p, zl);          // (convective outlet: the planes z <= 2 are plane 3's streamed populations)
    const unsigned b = threadIdx.x;
    double g[Q], rR, rho;
    unsigned j;
    Sums S;
    const GlobalRows rows{p, sg, y};
    if (p.first) {
        pull_q<true, false>(p, rows, zs, b, g, j);
        class_sums<true, false>(p, rows, glb_scal(p, plane_addr_q(p, p.fin, zs), j, zs, sg, y), zs, b, g, S);
    } else {
        pull_q<false, false>(p, rows, zs, b, g, j);
        class_sums<false, false>(p, rows, glb_scal(p, plane_addr_q(p, p.fin, zs), j, zs, sg, y), zs, b, g, S);
    }
    bc_q<true>(p, p.conv && p.z0 + zl - 1 <= 2 ? zs : zl, S, g, rR, rho);
    p.phi[idx] = phi_q(rR, rho);
    if (p.diag) {
        constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
        double mx = 0., my = 0., mz = 0.;
#pragma unroll
        for (int i = 0; i < Q; ++i) { mx += (double)CX[i] * g[i]; my += (double)CY[i] * g[i]; mz += (double)CZ[i] * g[i]; }
        p.diag[idx] = rR; p.diag[p.vol + idx] = rho - rR;
        p.diag[2 * p.vol + idx] = mx / rho; p.diag[3 * p.vol + idx] = my / rho; p.diag[4 * p.vol + idx] = mz / rho;
    }
}

// ---------------------------------------------------------------------------------------------- convective outlet
// [BoundaryCondition] BoundaryTypeOutlet = 'Convective' (AcceleratedRKGPU2D.py:700-784: convectiveOutletGPU, ...Ghost2GPU, ...Ghost3GPU copy
// the streamed populations of row 3 onto the rows 2, 1, 0 and re-sum the densities) as z planes.  The marching kernel does not compute
// those planes at all: on the slab that holds the lattice's bottom it starts at plane 3 (local zs) and takes the planes below as it takes a
// neighbour rank's halo plane (RK3Dev::halo_lo) -- phase field from `phi`, stored state as it stands.  Around it, per step:
//   before:  rk3dq_phase_field on plane zs (the streamed lattice's phase field there), rk3dq_conv_phi copies it onto the planes below;
//   after:   rk3dq_conv_collide collides the planes zs-1, zs-2 (z = 2, 1; the ghost plane z = 0 is neither collided nor stored, as ever)
//            from the populations pulled around plane zs, with the colour gradient of the copied phase field.
// The masks of the planes z = 0 .. 3 must coincide (lbmpm_rk3d_create checks): cell numbers and row records are then the same.
__global__ __launch_bounds__(256) void rk3dq_conv_phi(RK3Dev p, int zs)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.plane2 || !(p.flags[(size_t)zs * p.plane2 + k] & 1)) return;
    const double v = p.phi[(size_t)zs * p.plane2 + k];
    for (int zl = zs - 3; zl < zs; ++zl) p.phi[(size_t)zl * p.plane2 + k] = v;
}

template <bool FIRST, bool MRT>
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_conv_collide(RK3Dev p, int zs)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, zl = zs - 1 - (int)blockIdx.z;
    if (y >= p.ny) return;
    const bool fluid = x >= 0 && (p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1);
    if (__ballot(fluid) == 0ull) {        // (a wave = a row segment) nothing here: flag 3, as the marching kernel writes it
        if (threadIdx.x == 0) p.pur_out[row_index(p, zl, y, sg)] = 3u;
        return;
    }
    if (!fluid) return;
    const GlobalRows rows{p, sg, y};
    const unsigned b = threadIdx.x;
    double g[Q], rR, rho;
    unsigned js;
    Sums S;
    pull_q<FIRST, false>(p, rows, zs, b, g, js);
    class_sums<FIRST, false>(p, rows, glb_scal(p, plane_addr_q(p, p.fin, zs), js, zs, sg, y), zs, b, g, S);
    bc_q<true>(p, zs, S, g, rR, rho);             // (plane 3 carries no boundary rule: densities re-summed, A:718-719)
    // colour gradient of the copied phase field (the planes zl-1, zl, zl+1 all hold plane zs's; non-fluid cells solidPhi)
    double gx = 0., gy = 0., gz = 0.;
    const int xs[3] = {wrapi(x - 1, p.nx), x, wrapi(x + 1, p.nx)}, ys[3] = {wrapi(y - 1, p.ny), y, wrapi(y + 1, p.ny)};
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const double ph = p.phi[(size_t)(zl + CZ[i]) * p.plane2 + (size_t)ys[1 + CY[i]] * p.pitch + xs[1 + CX[i]]];
        gx += 3. * wq(i) * (double)CX[i] * ph;
        gy += 3. * wq(i) * (double)CY[i] * ph;
        gz += 3. * wq(i) * (double)CZ[i] * ph;
    }
    const RowTab t = rows(zl, 0);
    const unsigned j = t.first + bits_below<false>(t.m, b);
    const unsigned long long p0 = pstart_of(p, zl);
    const unsigned cnt = (unsigned)(pstart_of(p, zl + 1) - p0);
    collide_store<2, MRT>(p, reinterpret_cast<char *>(p.fout) + (size_t)p0 * CELLB, cnt * 8u, j * 8u, true, g, rR, rho - rR, gx, gy, gz,
                          p.pur_out + row_index(p, zl, y, sg));
}

// ---------------------------------------------------------------------------------------------- slab exchange (one per step)
// What a rank ships across a cut, per fluid cell of its face plane (plane nzl upwards, plane 1 downwards), cnt cells:
//   [0, 5 cnt)      the five populations that cross the cut
//   [5 cnt, 9 cnt)  the cell's record {k_R, A} (the constant for a flagged row: the receiver stores it like any other)
//   [9 cnt, 13 cnt) {r0, rX, t0, tX}: the class sums (red, total) of the cell's NEXT pull that the sender owns -- c_z = 0 and the
//                   class whose upstream cells lie in the sender's own second plane (X = +1 upwards, -1 downwards).  The receiver adds
//                   the class that crosses the cut, from its own face plane, and has the phase field of the neighbour's face plane
//                   (the halo plane its colour gradient reads) without a second exchange: rho_R = (r0 + r+) + r-, bit for bit what the
//                   owner computes in rk3dq_fused.  Before the first step nothing streams: {r0 + r+, r-, t0 + t+, t-} of the cell itself.
//   [13 cnt, ...)   the row flags of the face plane, one 32-bit word per row segment
constexpr int FACE_DOUBLES = 13;
__device__ constexpr int FACE_UP[5] = {5, 11, 14, 15, 18}, FACE_DN[5] = {6, 12, 13, 16, 17};

__global__ __launch_bounds__(BX3 *BY3) void rk3dq_face_pack(RK3Dev p, double *send_up, double *send_dn, int has_below, int has_above)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, face = blockIdx.z;    // face 0: top plane, upwards
    if ((face == 0 && !has_above) || (face == 1 && !has_below) || y >= p.ny) return;
    const int zf = face == 0 ? p.nzl : 1;
    double *msg = face == 0 ? send_up : send_dn;
    const unsigned long long p0 = pstart_of(p, zf);
    const size_t cnt = (size_t)(pstart_of(p, zf + 1) - p0);
    const unsigned flag = p.pur_in[row_index(p, zf, y, sg)];
    if (threadIdx.x == 0) reinterpret_cast<uint32_t *>(msg + FACE_DOUBLES * cnt)[y * p.nseg + sg] = flag;
    if (x < 0 || !(p.flags[(size_t)zf * p.plane2 + (size_t)y * p.pitch + x] & 1)) return;
    const GlobalRows rows{p, sg, y};
    const unsigned b = threadIdx.x;
    const RowTab t = rows(zf, 0);
    const unsigned j = t.first + bits_below<false>(t.m, b);
    const double *pl = p.fin + (size_t)p0 * QS;
    for (int k = 0; k < 5; ++k) msg[(size_t)k * cnt + j] = pl[(size_t)(face == 0 ? FACE_UP[k] : FACE_DN[k]) * cnt + j];
    const double *s = pl + (size_t)Q * cnt + (size_t)j * 4;
    double *ms = msg + 5 * cnt + (size_t)j * 4;
    if (flag) { ms[0] = (flag & 1u) ? 1. : 0.; ms[1] = 0.; ms[2] = 0.; ms[3] = 0.; }
    else { ms[0] = s[0]; ms[1] = s[1]; ms[2] = s[2]; ms[3] = s[3]; }
    double g[Q], k0, a0, t0, kx, ax, tx;
    const GlbScal sc = glb_scal(p, plane_addr_q(p, p.fin, zf), j, zf, sg, y);
    double *mp = msg + 9 * cnt + (size_t)j * 4;
    if (p.first) {
        double km, am, tm;
        pull_class<0, true>(p, rows, zf, b, j, g); pull_class<1, true>(p, rows, zf, b, j, g); pull_class<-1, true>(p, rows, zf, b, j, g);
        class_sum_one<0, true, false>(p, rows, sc, zf, b, g, k0, a0, t0);
        class_sum_one<1, true, false>(p, rows, sc, zf, b, g, kx, ax, tx);
        class_sum_one<-1, true, false>(p, rows, sc, zf, b, g, km, am, tm);
        mp[0] = (k0 + a0) + (kx + ax); mp[1] = km + am; mp[2] = t0 + tx; mp[3] = tm;
        return;
    }
    pull_class<0, false>(p, rows, zf, b, j, g);
    class_sum_one<0, false, false>(p, rows, sc, zf, b, g, k0, a0, t0);
    if (face == 0) { pull_class<1, false>(p, rows, zf, b, j, g); class_sum_one<1, false, false>(p, rows, sc, zf, b, g, kx, ax, tx); }
    else { pull_class<-1, false>(p, rows, zf, b, j, g); class_sum_one<-1, false, false>(p, rows, sc, zf, b, g, kx, ax, tx); }
    mp[0] = k0 + a0; mp[1] = kx + ax; mp[2] = t0; mp[3] = tx;
}

// received face -> halo plane (0 from below, nzl + 1 from above) of the state f / pur
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_face_unpack(RK3Dev p, double *f, uint32_t *pur, const double *recv_below, const double *recv_above,
                                                              int has_below, int has_above)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, face = blockIdx.z;    // face 0: halo plane 0
    if ((face == 0 && !has_below) || (face == 1 && !has_above) || y >= p.ny) return;
    const int zh = face == 0 ? 0 : p.nzl + 1;
    const double *msg = face == 0 ? recv_below : recv_above;
    const unsigned long long p0 = pstart_of(p, zh);
    const size_t cnt = (size_t)(pstart_of(p, zh + 1) - p0);
    if (threadIdx.x == 0) pur[row_index(p, zh, y, sg)] = reinterpret_cast<const uint32_t *>(msg + FACE_DOUBLES * cnt)[y * p.nseg + sg];
    if (x < 0 || !(p.flags[(size_t)zh * p.plane2 + (size_t)y * p.pitch + x] & 1)) return;
    const GlobalRows rows{p, sg, y};
    const RowTab t = rows(zh, 0);
    const unsigned j = t.first + bits_below<false>(t.m, threadIdx.x);
    double *pl = f + (size_t)p0 * QS;
    for (int k = 0; k < 5; ++k) pl[(size_t)(face == 0 ? FACE_UP[k] : FACE_DN[k]) * cnt + j] = msg[(size_t)k * cnt + j];
    double *s = pl + (size_t)Q * cnt + (size_t)j * 4;
    const double *ms = msg + 5 * cnt + (size_t)j * 4;
    s[0] = ms[0]; s[1] = ms[1]; s[2] = ms[2]; s[3] = ms[3];
}

// phase field of the halo planes from the neighbour's class sums + the class that crosses the cut (p.fin / p.pur_in = the state
// whose halo planes rk3dq_face_unpack has just filled)
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_halo_phi(RK3Dev p, const double *recv_below, const double *recv_above, int has_below, int has_above)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, face = blockIdx.z;
    if ((face == 0 && !has_below) || (face == 1 && !has_above) || y >= p.ny || x < 0) return;
    const int zh = face == 0 ? 0 : p.nzl + 1;
    const size_t idx = (size_t)zh * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const double *msg = face == 0 ? recv_below : recv_above;
    const size_t cnt = (size_t)(pstart_of(p, zh + 1) - pstart_of(p, zh));
    const GlobalRows rows{p, sg, y};
    const unsigned b = threadIdx.x;
    const RowTab t = rows(zh, 0);
    const unsigned j = t.first + bits_below<false>(t.m, b);
    const double *mp = msg + 9 * cnt + (size_t)j * 4;
    Sums S;
    S.a0 = S.ap = S.am = 0.;
    if (p.first) {          // nothing has streamed: the neighbour sent the cell's own sums, {r0 + r+, r-, t0 + t+, t-}
        S.k0 = mp[0]; S.kp = 0.; S.km = mp[1]; S.t0 = mp[2]; S.tp = 0.; S.tm = mp[3];
    } else {
        double g[Q];
        const GlbScal sc = glb_scal(p, plane_addr_q(p, p.fin, zh), j, zh, sg, y);
        S.k0 = mp[0]; S.t0 = mp[2];
        if (face == 0) {    // the neighbour below owns c_z = 0 and +1 of its top plane; c_z = -1 comes from this rank's plane 1
            S.kp = mp[1]; S.tp = mp[3];
            pull_class<-1, false>(p, rows, zh, b, j, g);
            class_sum_one<-1, false, false>(p, rows, sc, zh, b, g, S.km, S.am, S.tm);
        } else {
            S.km = mp[1]; S.tm = mp[3];
            pull_class<1, false>(p, rows, zh, b, j, g);
            class_sum_one<1, false, false>(p, rows, sc, zh, b, g, S.kp, S.ap, S.tp);
        }
    }
    double rR, rho;
    bc_q<false>(p, zh, S, nullptr, rR, rho);
    p.phi[idx] = phi_q(rR, rho);
}

// development aid (lbmpm_rk3d_debug_plane): component comp of the stored state f at plane zl (halo planes included) as a dense
// nx x ny plane -- 0..18 the populations g_i, 19..22 the record as stored (not the constant of a flagged row); zeros off the fluid
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_debug_plane(RK3Dev p, const double *f, int zl, int comp, double *out)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y;
    if (y >= p.ny || x < 0) return;
    double v = 0.;
    if (p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1) {
        const GlobalRows rows{p, sg, y};
        const RowTab t = rows(zl, 0);
        const unsigned j = t.first + bits_below<false>(t.m, threadIdx.x);
        const unsigned long long p0 = pstart_of(p, zl);
        const size_t cnt = (size_t)(pstart_of(p, zl + 1) - p0);
        const double *pl = f + (size_t)p0 * QS;
        v = comp < Q ? pl[(size_t)comp * cnt + j] : pl[(size_t)Q * cnt + (size_t)j * 4 + (comp - Q)];
    }
    out[(size_t)y * p.nx + x] = v;
}
