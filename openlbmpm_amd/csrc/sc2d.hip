// sc2d.hip -- two-component Shan-Chen D2Q9 time stepper for gfx950: original Shan-Chen
// (velocity-shift forcing) and the explicit forcing scheme (EFS), SRT / MRT.
//
// Replaces the per-kernel loops of ShanChenD2Q9.runOptimizedLBM / runOptimizedEFLBM
// (ShanChen2D/ShanChenD2Q9.py:1433-1629, :1631-2087) -- see include/lbmpm.h for the kernel
// list.  "O:" = ShanChen2D/OptimizedD2Q9GPU.py, "E:" = ShanChen2D/ExplicitD2Q9GPU.py,
// "D:" = ShanChen2D/ShanChenD2Q9.py.
//
// Layout: dense f[q][y*pitch+x] = {f_0, f_1} (both components side by side), ping-pong; solidnbr / flags bytes as
// in rk2d.hip.  State between steps = post-collision populations at their own node.
// One fused kernel per time step: a 64x8 tile + 1-node halo recomputes rho_k (= psi_k) of the
// streamed, boundary-corrected lattice into LDS, then every node evaluates the
// pseudopotential force from the LDS tile, collides and stores.
#include "lbmpm_common.h"
#include "d2q9_device.h"

#include <cmath>
#include <cstdlib>

namespace {

using lbmpm::set_error;
using namespace lbmpm_dev;

struct SCDev {
    int nx, ny, pitch;
    size_t plane;
    const uint8_t *flags;
    const uint8_t *solidnbr;
    const double *fin;
    double *fout;
    double *F;               // [4][plane] Fx0, Fx1, Fy0, Fy1 of the last iteration
    const double *fold_in;   // [18][3][pitch] pre-collision f-bar of rows 0..2 (EFS convective outlet), then [2][pitch]: F_y of both components
    double *fold_out;        // on row 3 as the last step left it -- ping-pong, so that the rows 0..2 (halo columns of x-adjacent tiles, the
                             // wrap-around halo of the last tile row) never read what another workgroup of the same launch writes
    double *diag;            // [28][plane] or nullptr
    double *psi;             // [2][plane]   (initialisation only)
    double *scrA, *scrB;     // [18][plane]  (initialisation only): f_eq, F_i
    double tau[2], G, Gs[2], vyIn[2];
    int model, mrt, outlet, first, keep_force;
    int scheme;              // [ForceScheme] ExplicitScheme: 4, 8, 10
    int sh, nobc;            // scheme 8: boundary rows one row further inside (two ghost rows); scheme 10: no boundary kernels
    int chang;               // BoundaryMethod 'Chang': velocity inlet from this step's and the last step's populations
    const double *chg_in;    // [6][pitch]: populations 4, 7, 8 of both components on the inlet row as the last step left them
    double *chg_out;
};

constexpr int D_RHO = 18, D_VX = 20, D_VY = 21, D_FX = 22, D_FY = 24, D_UEQ = 26, D_PLANES = 28;

// O:839-863 constantVelocityZouHeBoundaryHigher (one component)
__device__ __forceinline__ void bc_inlet(double v, double g[9])
{
    const double rho = (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / (1. + v);
    g[4] = g[2] - 2. / 3. * rho * v;
    g[7] = g[5] + (g[1] - g[3]) / 2. - 1. / 6. * rho * v;
    g[8] = g[6] - (g[1] - g[3]) / 2. - 1. / 6. * rho * v;
}

// O:1127-1161 calVelocityBoundaryHigherChangGPU (one component): o4, o7, o8 = populations 4, 7, 8 of this node when the
// previous loop pass began (savePDFLastStep, S:1853: the state its inlet kernels left)
__device__ __forceinline__ void bc_inlet_chang(double v, double g[9], double o4, double o7, double o8)
{
    const double rho = (g[0] + g[1] + g[3] + 2. * (g[2] + g[5] + g[6])) / (1. + v);
    g[4] = o4 - 2. / 3. * (rho * v + o4 + o7 + o8) + 2. / 3. * (g[2] + g[5] + g[6]);
    g[7] = o7 + 1. / 2. * (g[1] - g[3]) + 1. / 6. * (g[2] - o4) + 2. / 3. * (g[5] - o7) - 1. / 3. * (g[6] - o8) - 1. / 6. * rho * v;
    g[8] = o8 - 1. / 6. * rho * v - 1. / 2. * (g[1] - g[3]) + 1. / 6. * (g[2] - o4) - 1. / 3. * (g[5] - o7) + 2. / 3. * (g[6] - o8);
}
__device__ __forceinline__ void inlet_row(const SCDev &p, int x, double f0[9], double f1[9])
{
    if (p.chang) {
        const double *o = p.chg_in + x;
        bc_inlet_chang(p.vyIn[0], f0, o[0], o[p.pitch], o[2 * (size_t)p.pitch]);
        bc_inlet_chang(p.vyIn[1], f1, o[3 * (size_t)p.pitch], o[4 * (size_t)p.pitch], o[5 * (size_t)p.pitch]);
    } else { bc_inlet(p.vyIn[0], f0); bc_inlet(p.vyIn[1], f1); }
}
// the owner of an inlet-row node keeps what the next step's Chang inlet needs
__device__ __forceinline__ void keep_inlet_row(const SCDev &p, int x, const double f0[9], const double f1[9])
{
    double *o = p.chg_out + x;
    o[0] = f0[4]; o[p.pitch] = f0[7]; o[2 * (size_t)p.pitch] = f0[8];
    o[3 * (size_t)p.pitch] = f1[4]; o[4 * (size_t)p.pitch] = f1[7]; o[5 * (size_t)p.pitch] = f1[8];
}

// O:555-585 constantPressureZouHeBoundaryLower: density hard-coded per component
__device__ __forceinline__ void bc_outlet(double d, double g[9])
{
    const double v = 1. - (g[0] + g[1] + g[3] + 2. * (g[4] + g[7] + g[8])) / d;
    g[2] = g[4] + 2. / 3. * v * d;
    g[5] = g[7] + 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
    g[6] = g[8] - 1. / 2. * (g[3] - g[1]) + 1. / 6. * d * v;
}

__device__ __forceinline__ double mom_x(const double g[9]) { return g[1] - g[3] + g[5] - g[6] - g[7] + g[8]; }
__device__ __forceinline__ double mom_y(const double g[9]) { return g[2] - g[4] + g[5] + g[6] - g[7] - g[8]; }

// Lattice state of node (x,y) as the reference holds it when the force chain starts:
//   EFS (D:1897-2024): streamed f-bar, outlet BC, inlet BC + ghost rows, rho = sum f.
//   SC  (D:1592-1622, :1523-1539, :1579): streamed f, outlet copy rows, [INLET: inlet BC +
//        ghost row], rho = sum f.
// INLET=false gives the SC end-of-iteration view (what calPhysicalVelocity :1626 sees).
template <bool INLET>
__device__ __forceinline__ void node_state(const SCDev &p, int x, int y, double f0[9], double f1[9],
                                           double &r0, double &r1);

// node_state in two halves, so that a thread can have the pulls of two nodes in flight at once:
//   node_source_row: the row the node's populations are pulled at (ghost rows: their source row; convective outlet rows: row 3),
//   -- pull_node(p, x, ys, f0, f1) --
//   node_finish:     outlet / inlet rules on the pulled populations, densities.
template <bool INLET>
__device__ __forceinline__ int node_source_row(const SCDev &p, int y)
{
    const bool bc = !p.nobc;
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2 && !p.first) return 3;
    int ys = y;
    if (INLET && bc && y >= p.ny - 1 - p.sh) ys = p.ny - 2 - p.sh;       // ghost row(s) <- inlet row
    if (p.model == LBMPM_SC_MODEL_EFS && p.outlet == LBMPM_OUTLET_PRESSURE && bc && y <= p.sh) ys = 1 + p.sh;   // ghost row(s) <- outlet row
    return ys;
}
template <bool INLET>
__device__ __forceinline__ void node_finish(const SCDev &p, int x, int y, int ys, double f0[9], double f1[9], double &r0, double &r1)
{
    const bool bc = !p.nobc;
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2 && !p.first) {
        ys = y;                                  // (pulled at row 3; no inlet rule down here)
        if (p.model == LBMPM_SC_MODEL_EFS) {
            // O:1044-1120 convectiveOutletEach{,2,3}GPU: rows 2,1,0 in sequence,
            // f = (f_old + |vy(row 3)| f(row above)) / (1 + |vy(row 3)|)
            const double q0 = sum9(f0), q1 = sum9(f1);
            double ty = 0., tr = 0.;
            ty += (mom_y(f0) + 1. / 2. * p.fold_in[(size_t)54 * p.pitch + x]); tr += q0;
            ty += (mom_y(f1) + 1. / 2. * p.fold_in[(size_t)55 * p.pitch + x]); tr += q1;
            const double v = fabs(ty / tr);
            for (int r = 2; r >= y; --r) {
                const size_t o = (size_t)r * p.pitch + x;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    f0[j] = (p.fold_in[(size_t)j * 3 * p.pitch + o] + v * f0[j]) / (1. + v);
                    f1[j] = (p.fold_in[(size_t)(9 + j) * 3 * p.pitch + o] + v * f1[j]) / (1. + v);
                }
            }
        }
        // SC: O:960-1038 plain copies of row 3 into rows 2,1,0
    } else if (p.model == LBMPM_SC_MODEL_EFS && p.outlet == LBMPM_OUTLET_PRESSURE && bc && ys == 1 + p.sh) {
        bc_outlet(1.0, f0);
        bc_outlet(0.02, f1);
    }
    if (INLET && bc && ys == p.ny - 2 - p.sh) inlet_row(p, x, f0, f1);
    r0 = sum9(f0);
    r1 = sum9(f1);
}
template <bool INLET>
__device__ __forceinline__ void node_state(const SCDev &p, int x, int y, double f0[9], double f1[9],
                                           double &r0, double &r1)
{
    const int ys = node_source_row<INLET>(p, y);
    pull_node(p, x, ys, f0, f1);
    node_finish<INLET>(p, x, y, ys, f0, f1, r0, r1);
}

// Pseudopotential force on both components from psi of the 8 neighbours (nb0/nb1 = psi_0/psi_1
// at x + e_i, i = 1..8; sn = solid-neighbour bits).
//   EFS: E:51-216, weights 1/3, 1/12 (D:1675):  F_k = -6 psi_k sum_j G_kj sum_i w_i (psi_j(x+e_i) - psi_j(x)) e_i
//                                                     + sum_{solid i} -w_i Gs_k psi_k e_i
//   SC : O:1295-1392, weights 1/9, 1/36:         F_k = sum_i -w_i G_kj psi_k psi_j(x+e_i) e_i  (+ same solid term)
__device__ __forceinline__ void sc_force(const SCDev &p, unsigned sn, const double psi[2], const double nb0[9],
                                         const double nb1[9], double Fx[2], double Fy[2])
{
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const bool efs = p.model == LBMPM_SC_MODEL_EFS;
    const double wa = efs ? 1. / 3. : 1. / 9., wd = efs ? 1. / 12. : 1. / 36.;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double gx = 0., gy = 0., sx = 0., sy = 0.;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const double w = (i < 5) ? wa : wd;
            const double ex = EX[i], ey = EY[i];
            if (!((sn >> (i - 1)) & 1u)) {
                const double other = (k == 0) ? nb1[i] : nb0[i];      // G_kk = 0
                if (efs) {
                    const double d = other - psi[1 - k];
                    if (EX[i] != 0) gx += w * d * ex * p.G;
                    if (EY[i] != 0) gy += w * d * ey * p.G;
                } else {
                    if (EX[i] != 0) gx += -w * p.G * psi[k] * other * ex;
                    if (EY[i] != 0) gy += -w * p.G * psi[k] * other * ey;
                }
            } else {
                if (EX[i] != 0) sx += -w * p.Gs[k] * psi[k] * ex;
                if (EY[i] != 0) sy += -w * p.Gs[k] * psi[k] * ey;
            }
        }
        if (efs) {
            Fx[k] = -6.0 * psi[k] * gx + sx;
            Fy[k] = -6.0 * psi[k] * gy + sy;
        } else {
            Fx[k] = gx + sx;
            Fy[k] = gy + sy;
        }
    }
}

// ---- higher-isotropy force stencils, [ForceScheme] ExplicitScheme = 8 | 10 (E:627-955, :957-1377).
// Neighbour order of fillNeighboringNodesISO8 / ISO10 (E:392, :488): the 8 D2Q9 neighbours, distance-2
// axis, (2,2) diagonals, the 8 knight moves, then (scheme 10) distance-3 axis and the 8 (3,1) moves.
// A far neighbour counts only if the cells on the way to it are fluid (the `if` in front of every
// block of the reference kernels); a missing nearest neighbour adds the solid term -1/9 | -1/36.
// Scheme 8 differences psi_j(n) - psi_j(x), scheme 10 uses psi_j(n) alone.  psi comes from the
// global psi planes (two-kernel schedule, see sc2d_iso_psi / sc2d_iso_collide).
__device__ __forceinline__ void sc_force_iso(const SCDev &p, int x, int y, const double psi[2], double Fx[2], double Fy[2])
{
    constexpr int DX[36] = {1, 0, -1, 0, 1, -1, -1, 1, 2, 0, -2, 0, 2, -2, -2, 2, 2, 1, -1, -2, -2, -1, 1, 2,
                            3, 0, -3, 0, 3, 1, -1, -3, -3, -1, 1, 3};
    constexpr int DY[36] = {0, 1, 0, -1, 1, 1, -1, -1, 0, 2, 0, -2, 2, 2, -2, -2, 1, 2, 2, 1, -1, -2, -2, -1,
                            0, 3, 0, -3, 1, 3, 3, 1, -1, -3, -3, -1};
    constexpr int K2[8][2] = {{0, 4}, {1, 4}, {1, 5}, {2, 5}, {2, 6}, {3, 6}, {3, 7}, {0, 7}};
    constexpr int K3[8][4] = {{4, 16, 0, 8}, {1, 9, 4, 17}, {1, 9, 5, 18}, {2, 10, 5, 19}, {2, 10, 6, 20}, {3, 11, 6, 21}, {3, 11, 7, 22}, {0, 8, 7, 23}};
    const bool s10 = p.scheme == 10;
    const int nn = s10 ? 36 : 24;
    size_t nidx[36];
    unsigned long long fl = 0;
    for (int m = 0; m < nn; ++m) {
        int yy = y + DY[m], xx = x + DX[m];
        yy = yy < 0 ? yy + p.ny : (yy >= p.ny ? yy - p.ny : yy);
        xx = xx < 0 ? xx + p.nx : (xx >= p.nx ? xx - p.nx : xx);
        nidx[m] = (size_t)yy * p.pitch + xx;
        if (p.flags[nidx[m]] & 1) fl |= 1ull << m;
    }
    auto F = [&](int k) { return (fl >> k) & 1ull; };
    double fx[2] = {0., 0.}, fy[2] = {0., 0.};
    for (int m = 0; m < nn; ++m) {
        bool on = F(m);
        if (on && m >= 8) {
            if (m < 16) on = F(m - 8);
            else if (m < 24) on = F(K2[m - 16][0]) || F(K2[m - 16][1]);
            else if (m < 28) on = F(m - 24) && F(m - 16);
            else on = (F(K3[m - 28][0]) && F(K3[m - 28][1])) || (F(K3[m - 28][2]) && F(K3[m - 28][3]));
        }
        const int dx = DX[m], dy = DY[m];
        const double sx = dx > 0 ? 1. : -1., sy = dy > 0 ? 1. : -1.;
        if (on) {
            const double w = s10 ? (m < 4 ? 262. / 1785. : m < 8 ? 93. / 1190. : m < 12 ? 7. / 340. : m < 16 ? 9. / 9520. : m < 24 ? 6. / 595. : m < 28 ? 2. / 5355. : 1. / 7140.)
                                 : (m < 4 ? 4. / 21. : m < 8 ? 4. / 45. : m < 12 ? 1. / 60. : m < 16 ? 1. / 5040. : 2. / 315.);
            const double q[2] = {p.psi[nidx[m]], p.psi[p.plane + nidx[m]]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 1 - i;                                   // G_ii = 0
                const double d = s10 ? q[j] : q[j] - psi[j];
                if (dx == 1 || dx == -1) fx[i] += -6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dx == 2 || dx == -2) fx[i] += -2. * 6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dx == 3 || dx == -3) fx[i] += -3. * 6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dy == 1 || dy == -1) fy[i] += -6.0 * w * p.G * psi[i] * (d) * (sy);
                if (dy == 2 || dy == -2) fy[i] += -2. * 6.0 * w * p.G * psi[i] * (d) * (sy);
                if (dy == 3 || dy == -3) fy[i] += -3. * 6.0 * w * p.G * psi[i] * (d) * (sy);
            }
        } else if (m < 8 && !F(m)) {
            const double c = m < 4 ? -1. / 9. : -1. / 36.;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (dx != 0) fx[i] += c * p.Gs[i] * psi[i] * (sx);
                if (dy != 0) fy[i] += c * p.Gs[i] * psi[i] * (sy);
            }
        }
    }
    Fx[0] = fx[0]; Fx[1] = fx[1]; Fy[0] = fy[0]; Fy[1] = fy[1];
}

// EFS MRT: out = M^-1 S M d with S = diag(1,.6,1.5,1,1.2,1,1.2,1/tau,1/tau) (D:99-106, :484-496);
// M from SimpleD2Q9.py:107-124 (rows mutually orthogonal => M^-1 = M^T diag(1/|row|^2)).
__device__ __forceinline__ void mrt_relax(const double d[9], double itau, double out[9])
{
    // The two products by M and M^T with the sums every row shares formed once (the matrix entries are 0, +-1, +-2, +-4: 118
    // multiply-adds as written out row by row, 60 additions and 12 scalings this way).  Same algebra, another order of the additions.
    const double a = (d[1] + d[2]) + (d[3] + d[4]), b = (d[5] + d[6]) + (d[7] + d[8]);
    const double px = d[1] - d[3], py = d[2] - d[4];
    const double qx = (d[5] - d[6]) - (d[7] - d[8]), qy = (d[5] + d[6]) - (d[7] + d[8]);
    const double m0 = ((d[0] + a) + b) * (1. / 9.);
    const double m1 = ((2. * b - a) - 4. * d[0]) * (0.6 / 36.);
    const double m2 = ((4. * d[0] - 2. * a) + b) * (1.5 / 36.);
    const double m3 = (px + qx) * (1. / 6.);
    const double m4 = (qx - 2. * px) * (1.2 / 12.);
    const double m5 = (py + qy) * (1. / 6.);
    const double m6 = (qy - 2. * py) * (1.2 / 12.);
    const double m7 = ((d[1] - d[2]) + (d[3] - d[4])) * (itau * (1. / 4.));
    const double m8 = ((d[5] - d[6]) + (d[7] - d[8])) * (itau * (1. / 4.));
    const double A = (m0 - m1) - 2. * m2, B = (m0 + 2. * m1) + m2;
    const double X = m3 - 2. * m4, Y = m5 - 2. * m6, U = m3 + m4, V = m5 + m6;
    out[0] = (m0 - 4. * m1) + 4. * m2;
    out[1] = (A + X) + m7;
    out[2] = (A + Y) - m7;
    out[3] = (A - X) + m7;
    out[4] = (A - Y) - m7;
    out[5] = (B + (U + V)) + m8;
    out[6] = (B - (U - V)) - m8;
    out[7] = (B - (U + V)) + m8;
    out[8] = (B + (U - V)) - m8;
}

// Force chain + collision of one node, both components; f0/f1 updated in place.
//   EFS: u_eq E:340-363 (SRT) / E:1426-1449 (MRT), f_eq E:227-247, F_i E:255-271,
//        collision E:294-304 (SRT) / E:1379-1469 (MRT)
//   SC : fused force + velocity shift + BGK, O:1274-1449
template <bool MRT>
__device__ __forceinline__ void chain_collide(const SCDev &p, double f0[9], double f1[9], const double rho[2],
                                              const double Fx[2], const double Fy[2], double &ueqx, double &ueqy)
{
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    double *f[2] = {f0, f1};
    if (p.model == LBMPM_SC_MODEL_EFS) {
        double mx = 0., my = 0., rt = 0.;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double ex = mom_x(f[k]) + 1. / 2. * Fx[k], ey = mom_y(f[k]) + 1. / 2. * Fy[k];
            if (MRT) { mx += ex * 1.; my += ey * 1.; rt += rho[k] * 1.; }
            else { mx += ex / p.tau[k]; my += ey / p.tau[k]; rt = rt + rho[k] / p.tau[k]; }
        }
        const double ux = mx / rt, uy = my / rt;
        ueqx = ux; ueqy = uy;
        const double usq = ux * ux + uy * uy;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double d[9], ff[9];
            const double ics = 1. / (1. / 3. * rho[k]);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double eu = (double)EX[j] * ux + (double)EY[j] * uy;
                const double feq = W[j] * rho[k] * (1. + 3. * eu + 9. / 2. * (eu * eu) - 3. / 2. * usq);
                ff[j] = ((Fx[k] * ((double)EX[j] - ux)) + (Fy[k] * ((double)EY[j] - uy))) * feq * ics;
                d[j] = feq - f[k][j] - 1. / 2. * ff[j];
            }
            if (MRT) {
                double rl[9];
                mrt_relax(d, 1. / p.tau[k], rl);
#pragma unroll
                for (int j = 0; j < 9; ++j) f[k][j] = f[k][j] + rl[j] + 1. * ff[j];
            } else {
                const double om = 1. / p.tau[k];
#pragma unroll
                for (int j = 0; j < 9; ++j) f[k][j] = f[k][j] + om * d[j] + 1. * ff[j];
            }
        }
    } else {
        double vxt = 0., vyt = 0., rt = 0.;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            vxt += mom_x(f[k]) / p.tau[k];
            vyt += mom_y(f[k]) / p.tau[k];
            rt += rho[k] / p.tau[k];
        }
        const double pvx = vxt / rt, pvy = vyt / rt;
        ueqx = pvx; ueqy = pvy;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double ux = pvx + p.tau[k] * Fx[k] / rho[k], uy = pvy + p.tau[k] * Fy[k] / rho[k];
            const double usq = ux * ux + uy * uy;
            const double keep = 1. - 1. / p.tau[k], rw = rho[k] / p.tau[k];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double eu = (double)EX[j] * ux + (double)EY[j] * uy;
                f[k][j] = keep * f[k][j] + W[j] * rw * (1. + 3. * eu + 4.5 * (eu * eu) - 1.5 * usq);
            }
        }
    }
}

// ---------------------------------------------------------------- fused step kernel
constexpr int TW = 64, TH = 4, HALO = 1, RW = TW + 2 * HALO, RH = TH + 2 * HALO, THREADS = TW * TH;

template <bool MRT, bool STREAM>      // STREAM: non-temporal population stores (d2q9_device.h::store_pairs)
__global__ __launch_bounds__(THREADS, 4) void sc2d_fused(SCDev p, int tiles_x)
{
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    __shared__ double s_psi0[RH * RW];
    __shared__ double s_psi1[RH * RW];
    const int t = xcd_tile(blockIdx.x, gridDim.x, tiles_x);
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    const int tid = threadIdx.x, lx = tid % TW, ly = tid / TW;

    // phase A: the pull of the thread's own node and of "its" halo node (psi only) are issued back to back: one memory
    // round trip before the barrier
    const int x = tx0 + lx, y = ty0 + ly;
    const bool inside = (x < p.nx) && (y < p.ny);
    const int xw = inside ? x : wrapm_fast(x, p.nx), yw = inside ? y : wrapm_fast(y, p.ny);
    const size_t idx = (size_t)yw * p.pitch + xw;
    const bool fluid = p.flags[idx] & 1;
    const bool act = inside && fluid;
    const int ri = (HALO + ly) * RW + HALO + lx;
    double f0[9], f1[9], rho[2] = {0., 0.}, Fpx[2] = {0., 0.}, Fpy[2] = {0., 0.};
    unsigned sn = 0;
    constexpr int NHALO = 2 * RW + 2 * TH;
    static_assert(NHALO <= THREADS, "one halo node per thread");
    int hri = 0, hx = 0, hy = 0, hys = 0;
    bool hdo = false;
    double g0[9], g1[9];
    if (tid < NHALO) {
        int rx, ry;
        if (tid < RW) { ry = 0; rx = tid; }
        else if (tid < 2 * RW) { ry = RH - 1; rx = tid - RW; }
        else { const int m = tid - 2 * RW; ry = 1 + m / 2; rx = (m & 1) ? RW - 1 : 0; }
        hri = ry * RW + rx;
        hx = wrapm_fast(tx0 - HALO + rx, p.nx); hy = wrapm_fast(ty0 - HALO + ry, p.ny);
        hdo = p.flags[(size_t)hy * p.pitch + hx] & 1;
    }
    int ys = 0;
    if (fluid) { ys = node_source_row<true>(p, yw); pull_node(p, xw, ys, f0, f1); }
    if (hdo) { hys = node_source_row<true>(p, hy); pull_node(p, hx, hys, g0, g1); }
    if (fluid) {
        sn = p.solidnbr[idx];
        if (p.keep_force) {
            Fpx[0] = p.F[idx]; Fpx[1] = p.F[p.plane + idx];
            Fpy[0] = p.F[2 * p.plane + idx]; Fpy[1] = p.F[3 * p.plane + idx];
        }
        node_finish<true>(p, xw, yw, ys, f0, f1, rho[0], rho[1]);
        if (p.chang && inside && y == p.ny - 2) keep_inlet_row(p, x, f0, f1);
        s_psi0[ri] = rho[0];            // O:99-106 calFluidPotentialGPUEql: psi = rho
        s_psi1[ri] = rho[1];
    }
    if (hdo) {
        double a, b;
        node_finish<true>(p, hx, hy, hys, g0, g1, a, b);
        s_psi0[hri] = a;
        s_psi1[hri] = b;
    }
    __syncthreads();
    const bool line = lbmpm_dev::line_has_active<8>(act, tid & 63) && inside;     // populations: 16 bytes per lane
    if (!line) return;
    if (!act) {
#pragma unroll
        for (int j = 0; j < 9; ++j) { f0[j] = 0.; f1[j] = 0.; }
    }
    else {
    // phase D: force chain, collision, store
    double nb0[9], nb1[9];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int rn = ri + EY[i] * RW + EX[i];
        nb0[i] = s_psi0[rn];
        nb1[i] = s_psi1[rn];
    }
    double Fx[2], Fy[2], ueqx, ueqy;
    sc_force(p, sn, rho, nb0, nb1, Fx, Fy);
    if (p.diag) {     // end-of-iteration view of the EFS loop (D:2022-2087)
        double tx = 0., ty = 0., tr = 0.;
        tx += (mom_x(f0) + 1. / 2. * Fpx[0]); ty += (mom_y(f0) + 1. / 2. * Fpy[0]); tr += rho[0];
        tx += (mom_x(f1) + 1. / 2. * Fpx[1]); ty += (mom_y(f1) + 1. / 2. * Fpy[1]); tr += rho[1];
#pragma unroll
        for (int j = 0; j < 9; ++j) { p.diag[j * p.plane + idx] = f0[j]; p.diag[(9 + j) * p.plane + idx] = f1[j]; }
        p.diag[D_RHO * p.plane + idx] = rho[0]; p.diag[(D_RHO + 1) * p.plane + idx] = rho[1];
        p.diag[D_VX * p.plane + idx] = tx / tr; p.diag[D_VY * p.plane + idx] = ty / tr;
        p.diag[D_FX * p.plane + idx] = Fx[0]; p.diag[(D_FX + 1) * p.plane + idx] = Fx[1];
        p.diag[D_FY * p.plane + idx] = Fy[0]; p.diag[(D_FY + 1) * p.plane + idx] = Fy[1];
    }
    if (p.keep_force || (p.model == LBMPM_SC_MODEL_EFS && p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3)) {
        p.F[idx] = Fx[0]; p.F[p.plane + idx] = Fx[1];
        p.F[2 * p.plane + idx] = Fy[0]; p.F[3 * p.plane + idx] = Fy[1];
    }
    if (p.model == LBMPM_SC_MODEL_EFS && p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3) {
        p.fold_out[(size_t)54 * p.pitch + x] = Fy[0]; p.fold_out[(size_t)55 * p.pitch + x] = Fy[1];
    }
    if (p.model == LBMPM_SC_MODEL_EFS && p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2) {
        const size_t o = (size_t)y * p.pitch + x;     // savePDFLastStep O:70-80, rows 0..2 only
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            p.fold_out[(size_t)j * 3 * p.pitch + o] = f0[j];
            p.fold_out[(size_t)(9 + j) * 3 * p.pitch + o] = f1[j];
        }
    }
    chain_collide<MRT>(p, f0, f1, rho, Fx, Fy, ueqx, ueqy);
    if (p.diag) { p.diag[D_UEQ * p.plane + idx] = ueqx; p.diag[(D_UEQ + 1) * p.plane + idx] = ueqy; }
    }
    // non-fluid lanes of a line that holds fluid write zeros into their dead slots (full-line stores)
    // (the node index is rebuilt from an opaque copy of the thread id: kept across the collision it was the one value sc2d_fused<MRT>
    // spilled under its 128-register cap; `line` implies the node lies inside the lattice)
    unsigned tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
    store_pairs<STREAM>(p.fout, p.plane, (size_t)(ty0 + (int)(tq / TW)) * p.pitch + (tx0 + (int)(tq % TW)), f0, f1);
}

// ---------------------------------------------------------------- schemes 8 / 10: two sweeps per step
// The force stencil reaches 2 (3) cells: psi = rho of the streamed, boundary-corrected lattice goes
// through global memory (sweep 1), sweep 2 pulls again, evaluates the force, collides and stores.
__global__ __launch_bounds__(256) void sc2d_iso_psi(SCDev p)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double f0[9], f1[9], r0, r1;
    node_state<true>(p, x, y, f0, f1, r0, r1);
    p.psi[idx] = r0; p.psi[p.plane + idx] = r1;
}

template <bool MRT>
__global__ __launch_bounds__(256) void sc2d_iso_collide(SCDev p)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double f0[9], f1[9], rho[2];
    node_state<true>(p, x, y, f0, f1, rho[0], rho[1]);
    double Fpx[2] = {0., 0.}, Fpy[2] = {0., 0.};
    if (p.keep_force) { Fpx[0] = p.F[idx]; Fpx[1] = p.F[p.plane + idx]; Fpy[0] = p.F[2 * p.plane + idx]; Fpy[1] = p.F[3 * p.plane + idx]; }
    double Fx[2], Fy[2], ueqx, ueqy;
    sc_force_iso(p, x, y, rho, Fx, Fy);
    if (p.diag) {
        double tx = 0., ty = 0., tr = 0.;
        tx += (mom_x(f0) + 1. / 2. * Fpx[0]); ty += (mom_y(f0) + 1. / 2. * Fpy[0]); tr += rho[0];
        tx += (mom_x(f1) + 1. / 2. * Fpx[1]); ty += (mom_y(f1) + 1. / 2. * Fpy[1]); tr += rho[1];
#pragma unroll
        for (int j = 0; j < 9; ++j) { p.diag[j * p.plane + idx] = f0[j]; p.diag[(9 + j) * p.plane + idx] = f1[j]; }
        p.diag[D_RHO * p.plane + idx] = rho[0]; p.diag[(D_RHO + 1) * p.plane + idx] = rho[1];
        p.diag[D_VX * p.plane + idx] = tx / tr; p.diag[D_VY * p.plane + idx] = ty / tr;
        p.diag[D_FX * p.plane + idx] = Fx[0]; p.diag[(D_FX + 1) * p.plane + idx] = Fx[1];
        p.diag[D_FY * p.plane + idx] = Fy[0]; p.diag[(D_FY + 1) * p.plane + idx] = Fy[1];
    }
    if (p.keep_force || (p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3)) {
        p.F[idx] = Fx[0]; p.F[p.plane + idx] = Fx[1];
        p.F[2 * p.plane + idx] = Fy[0]; p.F[3 * p.plane + idx] = Fy[1];
    }
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3) { p.fold_out[(size_t)54 * p.pitch + x] = Fy[0]; p.fold_out[(size_t)55 * p.pitch + x] = Fy[1]; }
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2) {
        const size_t o = (size_t)y * p.pitch + x;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            p.fold_out[(size_t)j * 3 * p.pitch + o] = f0[j];
            p.fold_out[(size_t)(9 + j) * 3 * p.pitch + o] = f1[j];
        }
    }
    chain_collide<MRT>(p, f0, f1, rho, Fx, Fy, ueqx, ueqy);
    if (p.diag) { p.diag[D_UEQ * p.plane + idx] = ueqx; p.diag[(D_UEQ + 1) * p.plane + idx] = ueqy; }
    store_pairs<false>(p.fout, p.plane, idx, f0, f1);
}

// ---------------------------------------------------------------- schemes 8 / 10: one sweep per step
// The same step as sc2d_iso_psi + sc2d_iso_collide in ONE launch: a workgroup owns a 64 x 8 tile and recomputes psi = rho of the
// streamed, boundary-corrected lattice on tile + 2 (scheme 8) or tile + 3 (scheme 10) into LDS, every halo node by one thread beside
// its own pull; the 24 / 36-point force stencil with its line-of-sight rules then reads psi and the fluid mask from LDS (constant
// offsets: no index arrays, no scratch).  The arithmetic is sc_force_iso's, statement for statement: bit-equal to the two sweeps.
template <bool S10, typename Psi0, typename Psi1, typename Fl>
__device__ __forceinline__ void sc_force_iso_tile(const SCDev &p, const Psi0 &q0, const Psi1 &q1, const Fl &F, const double psi[2], double Fx[2], double Fy[2])
{
    constexpr int DX[36] = {1, 0, -1, 0, 1, -1, -1, 1, 2, 0, -2, 0, 2, -2, -2, 2, 2, 1, -1, -2, -2, -1, 1, 2,
                            3, 0, -3, 0, 3, 1, -1, -3, -3, -1, 1, 3};
    constexpr int DY[36] = {0, 1, 0, -1, 1, 1, -1, -1, 0, 2, 0, -2, 2, 2, -2, -2, 1, 2, 2, 1, -1, -2, -2, -1,
                            0, 3, 0, -3, 1, 3, 3, 1, -1, -3, -3, -1};
    constexpr int K2[8][2] = {{0, 4}, {1, 4}, {1, 5}, {2, 5}, {2, 6}, {3, 6}, {3, 7}, {0, 7}};
    constexpr int K3[8][4] = {{4, 16, 0, 8}, {1, 9, 4, 17}, {1, 9, 5, 18}, {2, 10, 5, 19}, {2, 10, 6, 20}, {3, 11, 6, 21}, {3, 11, 7, 22}, {0, 8, 7, 23}};
    constexpr int nn = S10 ? 36 : 24;
    unsigned long long fl = 0;
#pragma unroll
    for (int m = 0; m < nn; ++m) if (F(DX[m], DY[m])) fl |= 1ull << m;
    auto On = [&](int k) { return (fl >> k) & 1ull; };
    double fx[2] = {0., 0.}, fy[2] = {0., 0.};
#pragma unroll
    for (int m = 0; m < nn; ++m) {
        bool on = On(m);
        if (on && m >= 8) {
            if (m < 16) on = On(m - 8);
            else if (m < 24) on = On(K2[m - 16][0]) || On(K2[m - 16][1]);
            else if (m < 28) on = On(m - 24) && On(m - 16);
            else on = (On(K3[m - 28][0]) && On(K3[m - 28][1])) || (On(K3[m - 28][2]) && On(K3[m - 28][3]));
        }
        const int dx = DX[m], dy = DY[m];
        const double sx = dx > 0 ? 1. : -1., sy = dy > 0 ? 1. : -1.;
        if (on) {
            const double w = S10 ? (m < 4 ? 262. / 1785. : m < 8 ? 93. / 1190. : m < 12 ? 7. / 340. : m < 16 ? 9. / 9520. : m < 24 ? 6. / 595. : m < 28 ? 2. / 5355. : 1. / 7140.)
                                 : (m < 4 ? 4. / 21. : m < 8 ? 4. / 45. : m < 12 ? 1. / 60. : m < 16 ? 1. / 5040. : 2. / 315.);
            const double q[2] = {q0(dx, dy), q1(dx, dy)};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 1 - i;                                   // G_ii = 0
                const double d = S10 ? q[j] : q[j] - psi[j];
                if (dx == 1 || dx == -1) fx[i] += -6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dx == 2 || dx == -2) fx[i] += -2. * 6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dx == 3 || dx == -3) fx[i] += -3. * 6.0 * w * p.G * psi[i] * (d) * (sx);
                if (dy == 1 || dy == -1) fy[i] += -6.0 * w * p.G * psi[i] * (d) * (sy);
                if (dy == 2 || dy == -2) fy[i] += -2. * 6.0 * w * p.G * psi[i] * (d) * (sy);
                if (dy == 3 || dy == -3) fy[i] += -3. * 6.0 * w * p.G * psi[i] * (d) * (sy);
            }
        } else if (m < 8 && !On(m)) {
            const double c = m < 4 ? -1. / 9. : -1. / 36.;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (dx != 0) fx[i] += c * p.Gs[i] * psi[i] * (sx);
                if (dy != 0) fy[i] += c * p.Gs[i] * psi[i] * (sy);
            }
        }
    }
    Fx[0] = fx[0]; Fx[1] = fx[1]; Fy[0] = fy[0]; Fy[1] = fy[1];
}

template <bool MRT, bool S10>
__global__ __launch_bounds__(512, 4) void sc2d_iso_fused(SCDev p, int tiles_x)
{
    constexpr int IW = 64, IH = 8, H = S10 ? 3 : 2, QW = IW + 2 * H, QH = IH + 2 * H, NT = IW * IH;
    __shared__ double s_psi0[QH * QW];
    __shared__ double s_psi1[QH * QW];
    __shared__ uint8_t s_fl[QH * QW];
    const int t = xcd_tile(blockIdx.x, gridDim.x, tiles_x);
    const int tx0 = (t % tiles_x) * IW, ty0 = (t / tiles_x) * IH;
    const int tid = threadIdx.x, lx = tid % IW, ly = tid / IW;
    const int x = tx0 + lx, y = ty0 + ly;
    const bool inside = (x < p.nx) && (y < p.ny);
    const int xw = inside ? x : wrapm_fast(x, p.nx), yw = inside ? y : wrapm_fast(y, p.ny);
    const size_t idx = (size_t)yw * p.pitch + xw;
    const bool fluid = p.flags[idx] & 1;
    const bool act = inside && fluid;
    const int ri = (H + ly) * QW + H + lx;
    double f0[9], f1[9], rho[2] = {0., 0.};
    constexpr int NHALO = 2 * H * QW + 2 * H * IH;
    static_assert(NHALO <= NT, "one halo node per thread");
    int hri = -1;
    bool hfl = false;
    double ha = 0., hb = 0.;
    if (fluid) node_state<true>(p, xw, yw, f0, f1, rho[0], rho[1]);
    if (tid < NHALO) {
        int rx, ry, m = tid;
        if (m < H * QW) { ry = m / QW; rx = m % QW; }
        else if ((m -= H * QW) < H * QW) { ry = QH - H + m / QW; rx = m % QW; }
        else { m -= H * QW; ry = H + m / (2 * H); const int c = m % (2 * H); rx = c < H ? c : IW + c; }
        hri = ry * QW + rx;
        const int hx = wrapm_fast(tx0 - H + rx, p.nx), hy = wrapm_fast(ty0 - H + ry, p.ny);
        hfl = p.flags[(size_t)hy * p.pitch + hx] & 1;
        if (hfl) {
            double g0[9], g1[9];
            node_state<true>(p, hx, hy, g0, g1, ha, hb);
        }
    }
    s_fl[ri] = fluid;
    if (fluid) { s_psi0[ri] = rho[0]; s_psi1[ri] = rho[1]; }
    if (hri >= 0) { s_fl[hri] = hfl; if (hfl) { s_psi0[hri] = ha; s_psi1[hri] = hb; } }
    __syncthreads();
    const bool line = lbmpm_dev::line_has_active<8>(act, tid & 63) && inside;
    if (!line) return;
    if (!act) {
#pragma unroll
        for (int j = 0; j < 9; ++j) { f0[j] = 0.; f1[j] = 0.; }
    } else {
        double Fpx[2] = {0., 0.}, Fpy[2] = {0., 0.};
        if (p.keep_force) { Fpx[0] = p.F[idx]; Fpx[1] = p.F[p.plane + idx]; Fpy[0] = p.F[2 * p.plane + idx]; Fpy[1] = p.F[3 * p.plane + idx]; }
        double Fx[2], Fy[2], ueqx, ueqy;
        auto q0 = [&](int dx, int dy) { return s_psi0[ri + dy * QW + dx]; };
        auto q1 = [&](int dx, int dy) { return s_psi1[ri + dy * QW + dx]; };
        auto fl = [&](int dx, int dy) { return s_fl[ri + dy * QW + dx] != 0; };
        sc_force_iso_tile<S10>(p, q0, q1, fl, rho, Fx, Fy);
        if (p.diag) {
            double tx = 0., ty = 0., tr = 0.;
            tx += (mom_x(f0) + 1. / 2. * Fpx[0]); ty += (mom_y(f0) + 1. / 2. * Fpy[0]); tr += rho[0];
            tx += (mom_x(f1) + 1. / 2. * Fpx[1]); ty += (mom_y(f1) + 1. / 2. * Fpy[1]); tr += rho[1];
#pragma unroll
            for (int j = 0; j < 9; ++j) { p.diag[j * p.plane + idx] = f0[j]; p.diag[(9 + j) * p.plane + idx] = f1[j]; }
            p.diag[D_RHO * p.plane + idx] = rho[0]; p.diag[(D_RHO + 1) * p.plane + idx] = rho[1];
            p.diag[D_VX * p.plane + idx] = tx / tr; p.diag[D_VY * p.plane + idx] = ty / tr;
            p.diag[D_FX * p.plane + idx] = Fx[0]; p.diag[(D_FX + 1) * p.plane + idx] = Fx[1];
            p.diag[D_FY * p.plane + idx] = Fy[0]; p.diag[(D_FY + 1) * p.plane + idx] = Fy[1];
        }
        if (p.keep_force || (p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3)) {
            p.F[idx] = Fx[0]; p.F[p.plane + idx] = Fx[1];
            p.F[2 * p.plane + idx] = Fy[0]; p.F[3 * p.plane + idx] = Fy[1];
        }
        if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3) { p.fold_out[(size_t)54 * p.pitch + x] = Fy[0]; p.fold_out[(size_t)55 * p.pitch + x] = Fy[1]; }
        if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2) {
            const size_t o = (size_t)y * p.pitch + x;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                p.fold_out[(size_t)j * 3 * p.pitch + o] = f0[j];
                p.fold_out[(size_t)(9 + j) * 3 * p.pitch + o] = f1[j];
            }
        }
        chain_collide<MRT>(p, f0, f1, rho, Fx, Fy, ueqx, ueqy);
        if (p.diag) { p.diag[D_UEQ * p.plane + idx] = ueqx; p.diag[(D_UEQ + 1) * p.plane + idx] = ueqy; }
    }
    store_pairs<true>(p.fout, p.plane, idx, f0, f1);
}

// SC end-of-iteration view (D:1624-1629): streamed populations + outlet copies, rho, u with the
// force of that iteration.
template <bool INLET>
__global__ __launch_bounds__(256) void sc2d_observe(SCDev p, double *out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double f0[9], f1[9], r0, r1;
    node_state<INLET>(p, x, y, f0, f1, r0, r1);
#pragma unroll
    for (int j = 0; j < 9; ++j) { out[j * p.plane + idx] = f0[j]; out[(9 + j) * p.plane + idx] = f1[j]; }
    out[D_RHO * p.plane + idx] = r0; out[(D_RHO + 1) * p.plane + idx] = r1;
    double tx = 0., ty = 0., tr = 0.;
    tx += (mom_x(f0) + 1. / 2. * p.F[idx]); ty += (mom_y(f0) + 1. / 2. * p.F[2 * p.plane + idx]); tr += r0;
    tx += (mom_x(f1) + 1. / 2. * p.F[p.plane + idx]); ty += (mom_y(f1) + 1. / 2. * p.F[3 * p.plane + idx]); tr += r1;
    out[D_VX * p.plane + idx] = tx / tr; out[D_VY * p.plane + idx] = ty / tr;
    for (int c = 0; c < 4; ++c) out[(D_FX + c) * p.plane + idx] = p.F[c * p.plane + idx];
}

// ---------------------------------------------------------------- EFS initialisation (D:1714-1849 + first collision)
__global__ __launch_bounds__(256) void sc2d_init_psi(SCDev p)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double a = 0., b = 0.;
    for (int j = 0; j < 9; ++j) { a += p.fin[fslot(p.plane, j, idx, 0)]; b += p.fin[fslot(p.plane, j, idx, 1)]; }
    p.psi[idx] = a; p.psi[p.plane + idx] = b;
}

// psi -> F -> u_eq (from the untransformed f) -> f_eq -> F_i -> f-bar = f - F_i/2
template <bool MRT>
__global__ __launch_bounds__(256) void sc2d_init_chain(SCDev p)
{
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const unsigned sn = p.solidnbr[idx];
    double f0[9], f1[9], nb0[9], nb1[9];
    for (int j = 0; j < 9; ++j) { f0[j] = p.fin[fslot(p.plane, j, idx, 0)]; f1[j] = p.fin[fslot(p.plane, j, idx, 1)]; }
    const double rho[2] = {p.psi[idx], p.psi[p.plane + idx]};
    for (int i = 1; i < 9; ++i) {
        const size_t n = (size_t)wrapi(y + EY[i], p.ny) * p.pitch + wrapi(x + EX[i], p.nx);
        nb0[i] = p.psi[n]; nb1[i] = p.psi[p.plane + n];
    }
    double Fx[2], Fy[2];
    if (p.scheme == 4) sc_force(p, sn, rho, nb0, nb1, Fx, Fy);
    else sc_force_iso(p, x, y, rho, Fx, Fy);
    p.F[idx] = Fx[0]; p.F[p.plane + idx] = Fx[1]; p.F[2 * p.plane + idx] = Fy[0]; p.F[3 * p.plane + idx] = Fy[1];
    double *f[2] = {f0, f1};
    double mx = 0., my = 0., rt = 0.;
    for (int k = 0; k < 2; ++k) {
        const double ex = mom_x(f[k]) + 1. / 2. * Fx[k], ey = mom_y(f[k]) + 1. / 2. * Fy[k];
        if (MRT) { mx += ex * 1.; my += ey * 1.; rt += rho[k] * 1.; }
        else { mx += ex / p.tau[k]; my += ey / p.tau[k]; rt = rt + rho[k] / p.tau[k]; }
    }
    const double ux = mx / rt, uy = my / rt, usq = ux * ux + uy * uy;
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 9; ++j) {
            const double eu = (double)EX[j] * ux + (double)EY[j] * uy;
            const double feq = W[j] * rho[k] * (1. + 3. * eu + 9. / 2. * (eu * eu) - 3. / 2. * usq);
            const double ff = ((Fx[k] * ((double)EX[j] - ux)) + (Fy[k] * ((double)EY[j] - uy))) * feq / (1. / 3. * rho[k]);
            const size_t o = (size_t)(9 * k + j) * p.plane + idx;
            p.scrA[o] = feq;
            p.scrB[o] = ff;
            p.fout[fslot(p.plane, j, idx, k)] = f[k][j] - 1. / 2. * ff;          // transformPDFGPU E:278-288
        }
}

// pre-loop boundary kernels (inlet + ghost, then Dirichlet outlet + ghost; D:1772-1849) applied
// on the fly to f-bar, then the collision of loop iteration 0 with the pre-loop f_eq / F_i.
// Reads the f-bar buffer (fin here), writes fout.
template <bool MRT>
__global__ __launch_bounds__(256) void sc2d_init_collide(SCDev p)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    int ys = y;
    const bool bc = !p.nobc;
    if (bc && y >= p.ny - 1 - p.sh) ys = p.ny - 2 - p.sh;
    if (bc && p.outlet == LBMPM_OUTLET_PRESSURE && y <= p.sh) ys = 1 + p.sh;
    const size_t s = (size_t)ys * p.pitch + x;
    double f0[9], f1[9];
    for (int j = 0; j < 9; ++j) { f0[j] = p.fin[fslot(p.plane, j, s, 0)]; f1[j] = p.fin[fslot(p.plane, j, s, 1)]; }
    if (bc && ys == p.ny - 2 - p.sh) {
        inlet_row(p, x, f0, f1);                                   // (chg_in = the initial populations here, S:1642 / S:1803)
        if (p.chang && y == p.ny - 2) keep_inlet_row(p, x, f0, f1);
    }
    if (bc && p.outlet == LBMPM_OUTLET_PRESSURE && ys == 1 + p.sh) { bc_outlet(1.0, f0); bc_outlet(0.02, f1); }
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y <= 2) {
        const size_t o = (size_t)y * p.pitch + x;
        for (int j = 0; j < 9; ++j) {
            p.fold_out[(size_t)j * 3 * p.pitch + o] = f0[j];
            p.fold_out[(size_t)(9 + j) * 3 * p.pitch + o] = f1[j];
        }
    }
    if (p.outlet == LBMPM_OUTLET_CONVECTIVE && y == 3) {         // the pre-loop force of row 3 (sc2d_init_chain), read by the first pass's rows 0..2
        p.fold_out[(size_t)54 * p.pitch + x] = p.F[2 * p.plane + idx]; p.fold_out[(size_t)55 * p.pitch + x] = p.F[3 * p.plane + idx];
    }
    double *f[2] = {f0, f1};
    for (int k = 0; k < 2; ++k) {
        double d[9], ff[9];
        for (int j = 0; j < 9; ++j) {
            const size_t o = (size_t)(9 * k + j) * p.plane + idx;
            ff[j] = p.scrB[o];
            d[j] = p.scrA[o] - f[k][j] - 1. / 2. * ff[j];
        }
        if (MRT) {
            double rl[9];
            mrt_relax(d, 1. / p.tau[k], rl);
            for (int j = 0; j < 9; ++j) p.fout[fslot(p.plane, j, idx, k)] = f[k][j] + rl[j] + 1. * ff[j];
        } else {
            const double om = 1. / p.tau[k];
            for (int j = 0; j < 9; ++j) p.fout[fslot(p.plane, j, idx, k)] = f[k][j] + om * d[j] + 1. * ff[j];
        }
    }
}

// 'Chang' inlet, before the first step: the populations 4, 7, 8 of the inlet row as initialised (S:1642: deviceFluidPDFold
// starts as a copy of the initial f, which is what the pre-loop inlet kernel S:1803 reads)
__global__ void sc2d_chang_seed(SCDev p, double *out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= p.nx) return;
    const size_t idx = (size_t)(p.ny - 2) * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const int dirs[3] = {4, 7, 8};
    for (int k = 0; k < 2; ++k)
        for (int m = 0; m < 3; ++m) out[(size_t)(3 * k + m) * p.pitch + x] = p.fin[fslot(p.plane, dirs[m], idx, k)];
}

// 'Freeflow' outlet (S:1865-1884 with E:1476-1563): rows 2, 1, 0 take f-bar, F_i and f_eq of the row above before the
// collision, so all three leave it with the populations of row 3 -- applied to the post-collision buffer
__global__ void sc2d_freeflow_rows(SCDev p, double *f)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (x >= p.nx) return;
    const size_t dst = (size_t)r * p.pitch + x, src = (size_t)3 * p.pitch + x;
    if (!(p.flags[dst] & 1)) return;
    for (int j = 0; j < 9; ++j)
        for (int k = 0; k < 2; ++k) f[fslot(p.plane, j, dst, k)] = f[fslot(p.plane, j, src, k)];
}

}  // namespace

// ====================================================================== host side
struct lbmpm_sc2d {
    lbmpm_sc2d_config cfg;
    int nx, ny, pitch;
    size_t plane;
    int64_t nfluid = 0;
    hipStream_t stream = nullptr;
    uint8_t *flags = nullptr, *solidnbr = nullptr;
    double *fA = nullptr, *fB = nullptr, *F = nullptr, *foldA = nullptr, *foldB = nullptr, *diag = nullptr,
           *obs = nullptr, *chgA = nullptr, *chgB = nullptr;
    std::vector<uint8_t> h_domain;
    bool streamed = false, initialised = false, diag_valid = false, keep_force = false;
    int scheme = 4;                  // [ForceScheme] ExplicitScheme
    int iso_sweeps = 1;              // schemes 8 / 10: 1 = sc2d_iso_fused, 2 = sc2d_iso_psi + sc2d_iso_collide (LBMPM_SC2D_ISO_SWEEPS, the cross-check)
    double *psi = nullptr;           // [2][plane], schemes 8 / 10 only
    int64_t steps = 0, bytes = 0;
    hipGraphExec_t graph_exec = nullptr;              // GRAPH_STEPS captured time steps, valid while fA is where it was at capture
    const double *graph_fA = nullptr;
    bool graph_keep = false;
    int64_t graph_launches = 0, timed_steps = 0;      // hipGraph replays so far; time steps covered by the event pairs of the pool
    lbmpm::EventPool pool;
};

namespace {

SCDev make_dev(const lbmpm_sc2d *c)
{
    SCDev p{};
    p.nx = c->nx; p.ny = c->ny; p.pitch = c->pitch; p.plane = c->plane;
    p.flags = c->flags; p.solidnbr = c->solidnbr; p.fin = c->fA; p.fout = c->fB; p.F = c->F;
    p.fold_in = c->foldA; p.fold_out = c->foldB; p.diag = nullptr; p.psi = nullptr; p.scrA = p.scrB = nullptr;
    p.tau[0] = c->cfg.tau[0]; p.tau[1] = c->cfg.tau[1]; p.G = c->cfg.g_fluid;
    p.Gs[0] = c->cfg.g_solid[0]; p.Gs[1] = c->cfg.g_solid[1];
    p.vyIn[0] = c->cfg.inlet_velocity_y[0]; p.vyIn[1] = c->cfg.inlet_velocity_y[1];
    p.model = c->cfg.model; p.mrt = c->cfg.relaxation == LBMPM_RELAX_MRT; p.outlet = c->cfg.outlet_type;
    p.first = c->streamed ? 0 : 1; p.keep_force = c->keep_force ? 1 : 0;
    p.scheme = c->scheme; p.sh = c->scheme == 8 ? 1 : 0; p.nobc = (c->scheme == 10 || c->cfg.outlet_type == LBMPM_OUTLET_NONE) ? 1 : 0;
    p.psi = c->psi;
    p.chang = c->cfg.inlet_method == LBMPM_INLET_CHANG ? 1 : 0;

    p.chg_in = c->chgA; p.chg_out = c->chgB;
    return p;
}

template <typename T>
int dev_alloc(lbmpm_sc2d *c, T **ptr, size_t count)
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

// EFS: everything the reference does before its loop plus the collision of iteration 0.
int efs_initialise(lbmpm_sc2d *c)
{
    double *psi = nullptr, *scrA = nullptr, *scrB = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &psi, 2 * c->plane)) || (rc = dev_alloc(c, &scrA, 18 * c->plane)) ||
        (rc = dev_alloc(c, &scrB, 18 * c->plane))) return rc;
    SCDev p = make_dev(c);
    p.psi = psi; p.scrA = scrA; p.scrB = scrB;
    const dim3 g((c->nx + 63) / 64, (c->ny + 3) / 4), b(64, 4);
    const bool mrt = p.mrt;
    if (p.chang) sc2d_chang_seed<<<dim3((c->nx + 63) / 64), dim3(64), 0, c->stream>>>(p, c->chgB);
    sc2d_init_psi<<<g, b, 0, c->stream>>>(p);                       // fA -> psi
    if (mrt) sc2d_init_chain<true><<<g, b, 0, c->stream>>>(p);      // fA -> fB (f-bar), scrA, scrB, F
    else sc2d_init_chain<false><<<g, b, 0, c->stream>>>(p);
    SCDev q = p;
    q.fin = c->fB; q.fout = c->fA; q.fold_out = c->foldA;           // fB -> fA (post-collision), fold -> foldA
    q.chg_in = c->chgB; q.chg_out = c->chgA;                        // initial populations -> what the pre-loop inlet left
    if (mrt) sc2d_init_collide<true><<<g, b, 0, c->stream>>>(q);
    else sc2d_init_collide<false><<<g, b, 0, c->stream>>>(q);
    if (p.outlet == LBMPM_OUTLET_FREEFLOW) sc2d_freeflow_rows<<<dim3((c->nx + 63) / 64, 3), dim3(64), 0, c->stream>>>(p, c->fA);
    LBMPM_HIP_TRY(hipGetLastError());
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(psi); (void)hipFree(scrA); (void)hipFree(scrB);
    c->bytes -= (int64_t)(38 * c->plane * sizeof(double));
    c->streamed = true;
    c->initialised = true;
    return LBMPM_OK;
}

int launch_step(lbmpm_sc2d *c, bool diag, bool timed)
{
    if (c->cfg.model == LBMPM_SC_MODEL_EFS && !c->initialised) { const int rc = efs_initialise(c); if (rc) return rc; }
    SCDev p = make_dev(c);
    p.diag = diag ? c->diag : nullptr;
    // original Shan-Chen loop, first pass: deviceFluidPDFold is the initial f (S:1446) when the Chang inlet (S:1529) first reads it
    if (p.chang && !c->streamed && c->cfg.model == LBMPM_SC_MODEL_SHANCHEN)
        sc2d_chang_seed<<<dim3((c->nx + 63) / 64), dim3(64), 0, c->stream>>>(p, c->chgA);
    const int tiles_x = (c->nx + TW - 1) / TW, tiles_y = (c->ny + TH - 1) / TH;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool ev = timed && c->pool.take(&e0, &e1);
    if (ev) { LBMPM_HIP_TRY(hipEventRecord(e0, c->stream)); c->timed_steps += 1; }
    if (c->scheme != 4 && c->iso_sweeps == 1) {
        const int tx8 = (c->nx + 63) / 64, ty8 = (c->ny + 7) / 8;
        const dim3 g(tx8 * ty8), b(512);
        if (c->scheme == 10) { if (p.mrt) sc2d_iso_fused<true, true><<<g, b, 0, c->stream>>>(p, tx8); else sc2d_iso_fused<false, true><<<g, b, 0, c->stream>>>(p, tx8); }
        else { if (p.mrt) sc2d_iso_fused<true, false><<<g, b, 0, c->stream>>>(p, tx8); else sc2d_iso_fused<false, false><<<g, b, 0, c->stream>>>(p, tx8); }
    } else if (c->scheme != 4) {
        const dim3 g((c->nx + 63) / 64, (c->ny + 3) / 4), b(64, 4);
        sc2d_iso_psi<<<g, b, 0, c->stream>>>(p);
        if (p.mrt) sc2d_iso_collide<true><<<g, b, 0, c->stream>>>(p);
        else sc2d_iso_collide<false><<<g, b, 0, c->stream>>>(p);
    } else {
        // lattices small enough to live in the caches from step to step (configs[0]: 128^2) keep ordinary stores
        const dim3 g(tiles_x * tiles_y), b(THREADS);
        if (c->plane > ((size_t)1 << 18)) { if (p.mrt) sc2d_fused<true, true><<<g, b, 0, c->stream>>>(p, tiles_x); else sc2d_fused<false, true><<<g, b, 0, c->stream>>>(p, tiles_x); }
        else { if (p.mrt) sc2d_fused<true, false><<<g, b, 0, c->stream>>>(p, tiles_x); else sc2d_fused<false, false><<<g, b, 0, c->stream>>>(p, tiles_x); }
    }
    if (p.outlet == LBMPM_OUTLET_FREEFLOW) sc2d_freeflow_rows<<<dim3((c->nx + 63) / 64, 3), dim3(64), 0, c->stream>>>(p, p.fout);
    if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    LBMPM_HIP_TRY(hipGetLastError());
    std::swap(c->fA, c->fB);
    std::swap(c->foldA, c->foldB);
    std::swap(c->chgA, c->chgB);
    c->streamed = true;
    c->diag_valid = diag;
    c->steps += 1;
    return LBMPM_OK;
}

// Small lattices are launch-bound (128 x 128: 64 workgroups, 3 us of kernel behind 5 us of launch): there the time loop
// is replayed from a hipGraph of GRAPH_STEPS captured launches (an even number: the ping-pong buffers are back where they
// were), so the kernels follow each other at the graph's node-to-node gap instead of the host's launch rate.
constexpr int GRAPH_STEPS = 64;

int run_steps(lbmpm_sc2d *c, int64_t n, bool timed)
{
    int64_t k = 0;
    const bool small = c->plane <= ((size_t)1 << 18) && c->scheme == 4 && !getenv("LBMPM_NO_GRAPH");
    if (small && n >= 3 * GRAPH_STEPS) {
        // the first step may be special (EFS initialisation): run it directly
        int rc = launch_step(c, false, false);
        if (rc != LBMPM_OK) return rc;
        ++k;
        if (c->graph_exec && c->graph_keep == c->keep_force && c->graph_fA != c->fA) {      // other half of the ping-pong: one more direct step
            rc = launch_step(c, false, false);
            if (rc != LBMPM_OK) return rc;
            ++k;
        }
        bool counted = false;                               // a capture pass advances the host state by one batch
        if (!c->graph_exec || c->graph_fA != c->fA || c->graph_keep != c->keep_force) {       // (kept for the life of the context: instantiation costs ~1e4 launches)
            if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            c->graph_fA = c->fA; c->graph_keep = c->keep_force;
            LBMPM_HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            for (int g = 0; g < GRAPH_STEPS && rc == LBMPM_OK; ++g) rc = launch_step(c, false, false);     // recorded, not run
            hipError_t e = hipStreamEndCapture(c->stream, &graph);
            if (rc != LBMPM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            LBMPM_HIP_TRY(e);
            e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            LBMPM_HIP_TRY(e);
            counted = true;
        }
        while (true) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            const bool ev = timed && c->pool.take(&e0, &e1);
            if (ev) { LBMPM_HIP_TRY(hipEventRecord(e0, c->stream)); c->timed_steps += GRAPH_STEPS; }
            LBMPM_HIP_TRY(hipGraphLaunch(c->graph_exec, c->stream));
            if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
            if (!counted) c->steps += GRAPH_STEPS;
            counted = false;
            k += GRAPH_STEPS;
            c->graph_launches += 1;
            if (n - k < GRAPH_STEPS + 1) break;             // keep at least one direct step for the diagnostics of the last one
        }
        c->streamed = true;
        c->diag_valid = false;
    }
    for (; k < n; ++k) {
        const bool diag = (c->diag != nullptr) && (k == n - 1);
        // event pairs around every 8th launch only: a record on each side of every 0.1 ms kernel would
        // open a gap behind each of them and slow down the very loop that is being measured
        const int rc = launch_step(c, diag, timed && (k & 7) == 0);
        if (rc != LBMPM_OK) return rc;
    }
    return LBMPM_OK;
}

int copy_plane(lbmpm_sc2d *c, const double *dev, double *out, int ncomp)
{
    std::vector<double> h((size_t)ncomp * c->plane);
    LBMPM_HIP_TRY(hipMemcpyAsync(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const size_t k = (size_t)y * c->nx + x, d = (size_t)y * c->pitch + x;
            const bool fluid = c->h_domain[k] == 1;
            for (int i = 0; i < ncomp; ++i) out[k * ncomp + i] = fluid ? h[i * c->plane + d] : 0.0;
        }
    return LBMPM_OK;
}

}  // namespace

extern "C" int lbmpm_sc2d_create(const lbmpm_sc2d_config *cfg, const uint8_t *is_domain, lbmpm_sc2d **out)
{
    LBMPM_REQUIRE(cfg && is_domain && out, "lbmpm_sc2d_create: null argument");
    LBMPM_REQUIRE(cfg->nx >= 4 && cfg->ny >= 8 && cfg->nx < (1 << 30) && cfg->ny < (1 << 30),
                  "lbmpm_sc2d_create: domain %lld x %lld out of range", (long long)cfg->nx, (long long)cfg->ny);
    // (the kernels address a node inside a plane of 16-byte pairs with 32 bits: d2q9_device.h::pull_node)
    LBMPM_REQUIRE((size_t)((cfg->nx + 31) / 32 * 32) * (size_t)cfg->ny < ((size_t)1 << 28), "lbmpm_sc2d_create: more than 2^28 nodes per lattice plane");
    LBMPM_REQUIRE(cfg->model == LBMPM_SC_MODEL_SHANCHEN || cfg->model == LBMPM_SC_MODEL_EFS, "bad model %d", cfg->model);
    LBMPM_REQUIRE(cfg->relaxation == LBMPM_RELAX_SRT || cfg->relaxation == LBMPM_RELAX_MRT, "bad relaxation %d", cfg->relaxation);
    if (cfg->model == LBMPM_SC_MODEL_SHANCHEN && cfg->relaxation == LBMPM_RELAX_MRT) {
        set_error("the original Shan-Chen path is SRT only in the reference (ShanChenD2Q9.py:1584)");
        return LBMPM_ERR_UNSUPPORTED;
    }
    LBMPM_REQUIRE(cfg->outlet_type >= 0 && cfg->outlet_type <= 3, "bad outlet_type %d", cfg->outlet_type);
    LBMPM_REQUIRE(cfg->inlet_method == LBMPM_INLET_ZOUHE || cfg->inlet_method == LBMPM_INLET_CHANG, "bad inlet_method %d", cfg->inlet_method);
    if (cfg->outlet_type == LBMPM_OUTLET_FREEFLOW && !(cfg->model == LBMPM_SC_MODEL_EFS && cfg->relaxation == LBMPM_RELAX_SRT)) {
        // MRT: the loop transforms f-bar and F_i to moment space BEFORE it copies the rows (ShanChenD2Q9.py:1855-1884), so
        // rows 0-2 collide with a mix of their own moments and row 3's populations; the reference run turns NaN
        set_error("the 'Freeflow' outlet belongs to the explicit forcing loop with SRT "
                  "(ShanChenD2Q9.py:1865; with MRT the reference run diverges to NaN)");
        return LBMPM_ERR_UNSUPPORTED;
    }
    if (cfg->inlet_method == LBMPM_INLET_CHANG && !((cfg->force_scheme == 0 || cfg->force_scheme == 4) && cfg->outlet_type != LBMPM_OUTLET_NONE)) {
        set_error("BoundaryMethod 'Chang' exists for ExplicitScheme 4 only (ShanChenD2Q9.py:1529, :1803, :1999)");
        return LBMPM_ERR_UNSUPPORTED;
    }
    LBMPM_REQUIRE(cfg->tau[0] > 0.5 && cfg->tau[1] > 0.5, "FluidsTau must exceed 0.5");
    LBMPM_REQUIRE(cfg->variant == 0, "variant must be 0");
    LBMPM_REQUIRE(cfg->force_scheme == 0 || cfg->force_scheme == 4 || cfg->force_scheme == 8 || cfg->force_scheme == 10,
                  "ExplicitScheme must be 4, 8 or 10");
    LBMPM_REQUIRE(cfg->force_scheme == 0 || cfg->force_scheme == 4 || cfg->model == LBMPM_SC_MODEL_EFS,
                  "ExplicitScheme 8 / 10 belong to the explicit forcing scheme (EFS)");
    LBMPM_REQUIRE(!(cfg->force_scheme == 8 || cfg->force_scheme == 10) || cfg->ny >= 16, "ExplicitScheme 8 / 10 need ny >= 16");
    LBMPM_HIP_TRY(hipSetDevice(cfg->device));
    lbmpm_sc2d *c = new (std::nothrow) lbmpm_sc2d();
    if (!c) { set_error("out of host memory"); return LBMPM_ERR_NOMEM; }
    c->cfg = *cfg;
    c->nx = (int)cfg->nx; c->ny = (int)cfg->ny;
    c->pitch = (c->nx + 31) / 32 * 32;
    c->plane = (size_t)c->pitch * c->ny;
    c->h_domain.assign(is_domain, is_domain + (size_t)c->nx * c->ny);
    std::vector<uint8_t> hflags(c->plane, 0);
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const uint8_t v = is_domain[(size_t)y * c->nx + x] == 1 ? 1 : 0;
            hflags[(size_t)y * c->pitch + x] = v;
            c->nfluid += v;
        }
    if (c->nfluid == 0) { set_error("lbmpm_sc2d_create: the domain has no fluid node (is_domain == 1 marks fluid)"); delete c; return LBMPM_ERR_INVALID; }
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return LBMPM_ERR_HIP; }
    }
    int rc = LBMPM_OK;
#define TRY_RC(e) do { rc = (e); if (rc != LBMPM_OK) { lbmpm_sc2d_destroy(c); return rc; } } while (0)
    TRY_RC(dev_alloc(c, &c->flags, c->plane));
    TRY_RC(dev_alloc(c, &c->solidnbr, c->plane));
    TRY_RC(dev_alloc(c, &c->fA, 18 * c->plane));
    TRY_RC(dev_alloc(c, &c->fB, 18 * c->plane));
    TRY_RC(dev_alloc(c, &c->F, 4 * c->plane));
    c->scheme = cfg->force_scheme ? cfg->force_scheme : 4;
    if (const char *e = getenv("LBMPM_SC2D_ISO_SWEEPS")) c->iso_sweeps = atoi(e) == 2 ? 2 : 1;
    if (c->scheme != 4) TRY_RC(dev_alloc(c, &c->psi, 2 * c->plane));
    TRY_RC(dev_alloc(c, &c->foldA, (size_t)(18 * 3 + 2) * c->pitch));
    TRY_RC(dev_alloc(c, &c->foldB, (size_t)(18 * 3 + 2) * c->pitch));
    TRY_RC(dev_alloc(c, &c->chgA, (size_t)6 * c->pitch));
    TRY_RC(dev_alloc(c, &c->chgB, (size_t)6 * c->pitch));
#undef TRY_RC
    hipError_t e = hipMemcpyAsync(c->flags, hflags.data(), c->plane, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_error("flags upload failed: %s", hipGetErrorString(e)); lbmpm_sc2d_destroy(c); return LBMPM_ERR_HIP; }
    const dim3 b(64, 4), g((c->nx + 63) / 64, (c->ny + 3) / 4);
    setup_solidnbr<<<g, b, 0, c->stream>>>(c->nx, c->ny, c->pitch, c->flags, c->solidnbr);
    e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("set-up kernel failed: %s", hipGetErrorString(e)); lbmpm_sc2d_destroy(c); return LBMPM_ERR_HIP; }
    *out = c;
    return LBMPM_OK;
}

extern "C" void lbmpm_sc2d_destroy(lbmpm_sc2d *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *ptr : {(void *)c->flags, (void *)c->solidnbr, (void *)c->fA, (void *)c->fB, (void *)c->F,
                      (void *)c->foldA, (void *)c->foldB, (void *)c->diag, (void *)c->obs, (void *)c->psi,
                      (void *)c->chgA, (void *)c->chgB})
        if (ptr) (void)hipFree(ptr);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->pool.destroy();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int lbmpm_sc2d_set_pdf(lbmpm_sc2d *c, const double *pdf0, const double *pdf1)
{
    LBMPM_REQUIRE(c && pdf0 && pdf1, "lbmpm_sc2d_set_pdf: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    std::vector<double> h(18 * c->plane, 0.0);
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            if (c->h_domain[(size_t)y * c->nx + x] != 1) continue;
            const size_t s = ((size_t)y * c->nx + x) * 9, d = (size_t)y * c->pitch + x;
            for (int i = 0; i < 9; ++i) { h[fslot(c->plane, i, d, 0)] = pdf0[s + i]; h[fslot(c->plane, i, d, 1)] = pdf1[s + i]; }
        }
    LBMPM_HIP_TRY(hipMemcpyAsync(c->fA, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->F, 0, 4 * c->plane * sizeof(double), c->stream));
    // the fold buffers too: rows 54, 55 carry the outlet's F_y of lattice row 3, which the step writes for fluid nodes only -- a column
    // whose row-3 node is solid would keep the previous run's value on a re-initialised context (before round 4 it read the zeroed F)
    LBMPM_HIP_TRY(hipMemsetAsync(c->foldA, 0, (size_t)(18 * 3 + 2) * c->pitch * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->foldB, 0, (size_t)(18 * 3 + 2) * c->pitch * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    c->streamed = false; c->initialised = false; c->diag_valid = false; c->steps = 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_sc2d_set_density(lbmpm_sc2d *c, const double *rho0, const double *rho1)
{
    LBMPM_REQUIRE(c && rho0 && rho1, "lbmpm_sc2d_set_density: null argument");
    static const double Wd[9] = LBMPM_D2Q9_W;
    const size_t n = (size_t)c->nx * c->ny;
    std::vector<double> a(9 * n, 0.0), b(9 * n, 0.0);
    for (size_t k = 0; k < n; ++k)
        for (int i = 0; i < 9; ++i) { a[9 * k + i] = Wd[i] * rho0[k]; b[9 * k + i] = Wd[i] * rho1[k]; }
    return lbmpm_sc2d_set_pdf(c, a.data(), b.data());
}

extern "C" int lbmpm_sc2d_step(lbmpm_sc2d *c, int64_t nsteps)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_sc2d_step: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    return run_steps(c, nsteps, false);
}

extern "C" int lbmpm_sc2d_step_timed(lbmpm_sc2d *c, int64_t nsteps, double *ms_total, double *ms_dominant)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_sc2d_step_timed: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->cfg.model == LBMPM_SC_MODEL_EFS && !c->initialised) { const int rc = efs_initialise(c); if (rc) return rc; }
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
        *ms_dominant = c->timed_steps ? s * (double)nsteps / (double)c->timed_steps : 0.0;    // (a graph replay is timed as a whole)
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_sc2d_sync(lbmpm_sc2d *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

extern "C" int lbmpm_sc2d_enable_diagnostics(lbmpm_sc2d *c, int on)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (on && !c->diag) { const int rc = dev_alloc(c, &c->diag, D_PLANES * c->plane); if (rc) return rc; }
    if (!on && c->diag) {
        LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->diag); c->diag = nullptr; c->diag_valid = false;
        c->bytes -= (int64_t)(D_PLANES * c->plane * sizeof(double));
    }
    c->keep_force = on != 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_sc2d_get_field(lbmpm_sc2d *c, int field, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_sc2d_get_field: null argument");
    const bool rec = field >= LBMPM_SC_REC_PDF0 && field <= LBMPM_SC_REC_RHO1;
    LBMPM_REQUIRE((field >= LBMPM_SC_PDF0 && field <= LBMPM_SC_UEQY) || rec, "unknown field id %d", field);
    if (rec) {
        LBMPM_REQUIRE(c->cfg.model == LBMPM_SC_MODEL_SHANCHEN, "LBMPM_SC_REC_* describe the original Shan-Chen loop "
                      "(records are taken mid-iteration there); the EFS loop records the end-of-iteration fields");
        field -= LBMPM_SC_REC_PDF0;
    }
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->keep_force) {
        set_error("lbmpm_sc2d_get_field needs lbmpm_sc2d_enable_diagnostics(ctx, 1) before stepping");
        return LBMPM_ERR_STATE;
    }
    const double *src = nullptr;
    if (c->cfg.model == LBMPM_SC_MODEL_EFS) {
        if (!c->diag || !c->diag_valid) { set_error("no completed step with diagnostics yet"); return LBMPM_ERR_STATE; }
        src = c->diag;
    } else {
        LBMPM_REQUIRE(field < LBMPM_SC_UEQX, "u_eq is an EFS field");
        if (!c->obs) { const int rc = dev_alloc(c, &c->obs, D_PLANES * c->plane); if (rc) return rc; }
        SCDev p = make_dev(c);
        const dim3 b(64, 4), g((c->nx + 63) / 64, (c->ny + 3) / 4);
        if (rec) sc2d_observe<true><<<g, b, 0, c->stream>>>(p, c->obs);
        else sc2d_observe<false><<<g, b, 0, c->stream>>>(p, c->obs);
        LBMPM_HIP_TRY(hipGetLastError());
        LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
        src = c->obs;
    }
    switch (field) {
        case LBMPM_SC_PDF0: return copy_plane(c, src, out, 9);
        case LBMPM_SC_PDF1: return copy_plane(c, src + 9 * c->plane, out, 9);
        case LBMPM_SC_RHO0: return copy_plane(c, src + D_RHO * c->plane, out, 1);
        case LBMPM_SC_RHO1: return copy_plane(c, src + (D_RHO + 1) * c->plane, out, 1);
        case LBMPM_SC_VX: return copy_plane(c, src + D_VX * c->plane, out, 1);
        case LBMPM_SC_VY: return copy_plane(c, src + D_VY * c->plane, out, 1);
        case LBMPM_SC_FX0: return copy_plane(c, src + D_FX * c->plane, out, 1);
        case LBMPM_SC_FX1: return copy_plane(c, src + (D_FX + 1) * c->plane, out, 1);
        case LBMPM_SC_FY0: return copy_plane(c, src + D_FY * c->plane, out, 1);
        case LBMPM_SC_FY1: return copy_plane(c, src + (D_FY + 1) * c->plane, out, 1);
        case LBMPM_SC_UEQX: return copy_plane(c, src + D_UEQ * c->plane, out, 1);
        case LBMPM_SC_UEQY: return copy_plane(c, src + (D_UEQ + 1) * c->plane, out, 1);
    }
    return LBMPM_ERR_INVALID;
}

extern "C" int64_t lbmpm_sc2d_num_fluid_nodes(const lbmpm_sc2d *c) { return c ? c->nfluid : 0; }
extern "C" int64_t lbmpm_sc2d_steps_done(const lbmpm_sc2d *c) { return c ? c->steps : 0; }
extern "C" int64_t lbmpm_sc2d_device_bytes(const lbmpm_sc2d *c) { return c ? c->bytes : 0; }
extern "C" const char *lbmpm_sc2d_dominant_kernel(const lbmpm_sc2d *c) { (void)c; return "sc2d_fused"; }
