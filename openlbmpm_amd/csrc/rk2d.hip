// rk2d.hip -- colour-gradient D2Q9 (CSF) time stepper for gfx950.
//
// Replaces the 20-launch Numba-CUDA step of RKColorGradientLBM.runRKColorGradient2DCSF
// (RKCG2D/RKD2Q9.py:1295-1490, kernels in RKCG2D/AcceleratedRKGPU2D.py) -- see
// include/lbmpm.h for the kernel-by-kernel list and DESIGN.md for the data layout.
//
// Layout (HBM): dense grid, structure of arrays, direction-major:
//     f[q][y*pitch + x] = {f_R, f_B}   q in 0..8, float64 pairs, two ping-pong buffers
//     solidnbr[y*pitch + x]  uint8, bit (i-1) set <=> node + e_i is NOT fluid (periodic wrap)
//     flags[y*pitch + x]     uint8, bit0 = fluid
//     F[2][..]   CSF force of the last step (the reference's velocity lags it by one step)
//     ns[2][..]  unit normal of the solid surface on fluid nodes that touch solid
// State between steps = POST-COLLISION populations stored at their own node; a step
// starts by PULLING them (f_i(x) <- f*_i(x - e_i), or f*_opp(i)(x) when x - e_i is solid:
// identical to the reference's push + in-place half-way bounce-back,
// AcceleratedRKGPU2D.py:340-417).
#include "lbmpm_common.h"
#include "d2q9_device.h"

#include <cmath>
#include <cstdlib>

namespace {

using lbmpm::set_error;
using namespace lbmpm_dev;

// parameters of the perturbation operator (lbmpm_rk2d_set_perturbation; rk2dp_fused): an argument of their own, so that the CSF
// kernels' argument block -- and with it their scalar-register budget -- stays what it was
struct PertDev {
    double akR, akB, solidPhi, vyInR, vyInB, pOutR, pOutB;
    double *pd;        // [4][plane] rhoR, rhoB, vx, vy of the last step (nullptr = off)
};

struct RKDev {
    int nx, ny, pitch;
    size_t plane;
    const uint8_t *flags;
    const uint8_t *solidnbr;
    const double *fin;
    double *fout;
    double *F;         // [2][plane]
    const double *ns;  // [2][plane]
    double *phi;       // [plane]
    double *G;         // [2][plane]
    double *diag;      // [3][plane] vx, vy, K  (nullptr = off)
    double sigma, cosT, sinT, beta, delta, tauR, tauB, vyIn, pInB, pInR, pOut;
    int wetting, tautype, inlet, outlet;
    int first;         // 1: fin holds the initial (already post-streaming) state
    // D2Q5 tracer transport (AccelerateTransport2DRK.py), fused into phase D
    int ntr, trFree, trDirichlet;
    const double *gin;   // [ntr][5][plane]
    double *gout;
    double trCrit;
    double trM[25], trA[4][25], trBeta[4], trCb[4];
    double trRate, trJ[4];   // reaction A + B -> C between tracers 0, 1, 2 (rate 0 = off); J_0 of every tracer
};

// ---------------------------------------------------------------- boundary rows
// constantTotalVelocityInlet, A:2348-2412 (ratioB evaluated after rhoR was overwritten)
__device__ __forceinline__ void bc_inlet_velocity(double v, double fR[9], double fB[9], double &rhoR, double &rhoB)
{
    const double t0 = fR[0] + fB[0], t1 = fR[1] + fB[1], t2 = fR[2] + fB[2], t3 = fR[3] + fB[3];
    const double t5 = fR[5] + fB[5], t6 = fR[6] + fB[6];
    const double rho = (t0 + t1 + t3 + 2. * (t2 + t5 + t6)) / (1. + v);
    const double eq2 = rho * 1. / 9. * (1. + 3. * (1. * v) + 4.5 * (0. + 1. * v) * (0. + 1. * v) - 1.5 * (v * v));
    const double eq4 = rho * 1. / 9. * (1. + 3. * (-1. * v) + 4.5 * (0. + (-1.) * v) * (0. + (-1.) * v) - 1.5 * (v * v));
    const double t4 = eq4 + (t2 - eq2);
    const double eq5 = rho * 1. / 36. * (1. + 3. * (1. * v + 1. * 0.) + 4.5 * (1. * v + 1. * 0.) * (1. * v + 1. * 0.) - 1.5 * (v * v));
    const double eq7 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (-1.) * 0.) + 4.5 * ((-1.) * v + (-1.) * 0.) * ((-1.) * v + (-1.) * 0.) - 1.5 * (v * v));
    const double t7 = eq7 + (t5 - eq5);
    const double eq6 = rho * 1. / 36. * (1. + 3. * ((1.) * v + (-1.) * 0.) + 4.5 * ((1.) * v + (-1.) * 0.) * (1. * v + (-1.) * 0.) - 1.5 * (v * v));
    const double eq8 = rho * 1. / 36. * (1. + 3. * ((-1.) * v + (1.) * 0.) + 4.5 * ((-1.) * v + 1. * 0.) * ((-1.) * v + 1. * 0.) - 1.5 * (v * v));
    const double t8 = eq8 + (t6 - eq6);
    const double ratioR = rhoR / (rhoR + rhoB);
    rhoR = ratioR * rho;
    fR[4] = ratioR * t4; fR[7] = ratioR * t7; fR[8] = ratioR * t8;
    const double ratioB = rhoB / (rhoR + rhoB);     // reference quirk: uses the NEW rhoR
    rhoB = ratioB * rho;
    fB[4] = ratioB * t4; fB[7] = ratioB * t7; fB[8] = ratioB * t8;
}

// calConstPressureInletGPU, A:925-962 (Zou-He pressure per colour)
__device__ __forceinline__ void bc_inlet_pressure_one(double pc, double f[9], double &rho)
{
    const double v = -1. + (f[0] + f[1] + f[3] + 2. * (f[2] + f[5] + f[6])) / pc;
    f[4] = f[2] - 2. / 3. * pc * v;
    f[7] = f[5] + 1. / 2. * (f[1] - f[3]) - 1. / 6. * pc * v;
    f[8] = f[6] - 1. / 2. * (f[1] - f[3]) - 1. / 6. * pc * v;
    rho = pc;
}

// calConstPressureLowerGPUTotal, A:2560-2590
__device__ __forceinline__ void bc_outlet_pressure(double pL, double fR[9], double fB[9], double rhoR, double rhoB)
{
    const double t0 = fR[0] + fB[0], t1 = fR[1] + fB[1], t3 = fR[3] + fB[3], t4 = fR[4] + fB[4];
    const double t7 = fR[7] + fB[7], t8 = fR[8] + fB[8];
    const double v = 1. - 1. / pL * (t0 + t1 + t3 + 2. * (t4 + t7 + t8));
    const double t2 = t4 + 2. / 3. * (pL * v);
    const double t5 = t7 + 0.5 * (t3 - t1) + 1. / 6. * pL * v;
    const double t6 = t8 + 0.5 * (t1 - t3) + 1. / 6. * pL * v;
    const double ratioR = rhoR / (rhoR + rhoB);
    fR[2] = ratioR * t2; fR[5] = ratioR * t5; fR[6] = ratioR * t6;
    const double ratioB = rhoB / (rhoR + rhoB);
    fB[2] = ratioB * t2; fB[5] = ratioB * t5; fB[6] = ratioB * t6;
}

// State of node (x,y) as the reference holds it after the boundary kernels of a step
// (RKD2Q9.py:1299-1352): post-streaming populations + densities.
//   inlet 'Neumann'  : row ny-2 velocity BC, ghost row ny-1 = copy of it with rho re-summed
//                      (ghostPointsConstantVelocityRK, A:607-650)
//   inlet 'Dirichlet': row ny-2 Zou-He pressure, ghost row copies f and rho (A:968-1002)
//   outlet 'Dirichlet': row 1 Zou-He pressure on f_tot, ghost row 0 copies f and rho
//                      (ghostPointsConstPressureLowerRK, A:1045-1081; the reference tests the
//                       COMPACT index < nx, which is row 0 whenever row 0 is all fluid)
//   outlet 'Convective': rows 2,1,0 <- row 3, rho re-summed (A:700-784)
template <bool WITH_BC>
__device__ __forceinline__ int node_source_row(const RKDev &p, int y)
{
    int ys = y;
    if (WITH_BC) {
        if (y == p.ny - 1) ys = p.ny - 2;
        if (p.outlet == LBMPM_OUTLET_PRESSURE) { if (y == 0) ys = 1; }
        else { if (y <= 2) ys = 3; }
    }
    return ys;
}
// TR: the transport driver's order of the density sums (below) -- 1 always, 0 never, -1 decided at run time from p.ntr.  The fused
// kernels know it at compile time: the run-time test alone cost rk2d_fused 6 registers it does not have (128-register cap: 36 B of
// spills per lane, c2 82 -> 94 us per step).  For the tracer variant the form below (one summation per node, after the boundary
// rules; a second one only on the rows the rules need it) is the cheapest of four measured on c4: 0.430 ms per step against 0.452
// (run-time test, two summations everywhere) and 0.494 (second summation inside the rules' branches); round 2's order, which the
// real transport driver does not have, ran at 0.401.
template <bool WITH_BC, int TR = -1>
__device__ __forceinline__ void node_finish(const RKDev &p, int y, int ys, double fR[9], double fB[9], double &rhoR, double &rhoB);

template <bool WITH_BC, int TR = -1>
__device__ __forceinline__ void node_state(const RKDev &p, int x, int y, double fR[9], double fB[9],
                                           double &rhoR, double &rhoB)
{
    const int ys = node_source_row<WITH_BC>(p, y);
    pull_node(p, x, ys, fR, fB);
    node_finish<WITH_BC, TR>(p, y, ys, fR, fB, rhoR, rhoB);
}
// boundary rows and densities of a node whose populations pull_node(p, x, ys, ...) has fetched
template <bool WITH_BC, int TR>
__device__ __forceinline__ void node_finish(const RKDev &p, int y, int ys, double fR[9], double fB[9], double &rhoR, double &rhoB)
{
    // With tracers the step is the transport driver's (Transport2DRK.py:1177-1418): boundary rows first (:1199-1279), the densities
    // summed from the populations AFTERWARDS (:1281-1287) -- a pressure row then carries sum_i f_i, not the prescribed density.  Equal up
    // to the last bit; the wetting kernels' branch switches turn that bit into 1e-6 at single nodes (tests/test_tr_coupled.py).
    const bool after = WITH_BC && (TR == 1 || (TR < 0 && p.ntr > 0));
    const bool ruled = WITH_BC && (ys == p.ny - 2 || (p.outlet == LBMPM_OUTLET_PRESSURE && ys == 1));
    if (!after || ruled) {          // the boundary rules need the densities of the streamed populations
        rhoR = sum9(fR);
        rhoB = sum9(fB);
    }
    if (WITH_BC) {
        if (ys == p.ny - 2) {
            if (p.inlet == LBMPM_INLET_VELOCITY) {
                bc_inlet_velocity(p.vyIn, fR, fB, rhoR, rhoB);
                if (y == p.ny - 1) { rhoR = sum9(fR); rhoB = sum9(fB); }
            } else {
                bc_inlet_pressure_one(p.pInB, fB, rhoB);
                bc_inlet_pressure_one(p.pInR, fR, rhoR);
            }
        }
        if (p.outlet == LBMPM_OUTLET_PRESSURE && ys == 1) bc_outlet_pressure(p.pOut, fR, fB, rhoR, rhoB);
    }
    if (after) { rhoR = sum9(fR); rhoB = sum9(fB); }
}

// ---------------------------------------------------------------- collision pieces
// tau(phi): A:1967-1981 (same text in A:1815-1827, A:2052-2066, A:1755-1767)
__device__ __forceinline__ double tau_of(const RKDev &p, double Phi, double rR, double rB)
{
    double tau = 1.;
    if (Phi > p.delta) tau = p.tauR;
    else if (Phi < -p.delta) tau = p.tauB;
    else if (fabs(Phi) <= p.delta) {
        if (p.tautype == 1) {
            tau = 0.5 + 1. / ((1. + Phi) / (2. * (p.tauR - 0.5)) + (1. - Phi) / (2. * (p.tauB - 0.5)));
        } else if (p.tautype == 2) {
            const double ratioR = rR / (rR + rB), ratioB = rB / (rR + rB);
            const double miuR = 3. / (p.tauR - 0.5), miuB = 3. / (p.tauB - 0.5);
            const double miu = 1. / (ratioR * miuR + ratioB * miuB);
            tau = 3. * miu + 0.5;
        }
    }
    return tau;
}

// e . v = (double)ex * vx + (double)ey * vy for a lattice direction (components 0, +-1) without the products by zero and the additions of
// their results: x * 1 = x, x * -1 = -x and x + 0 * y = x hold exactly for finite operands, so the value is the reference's bit for bit (only
// the sign of an exact zero can differ); hipcc may not drop 0 * y itself.  pm(e, t) = t * (double)e for e = +-1.
__device__ __forceinline__ double pm(int e, double t) { return e > 0 ? t : -t; }
__device__ __forceinline__ double edot(int ex, int ey, double vx, double vy)
{
    if (ex == 0 && ey == 0) return 0.;
    if (ey == 0) return pm(ex, vx);
    if (ex == 0) return pm(ey, vy);
    return pm(ex, vx) + pm(ey, vy);
}

// A:170-176 calEquilibriumRK2D
__device__ __forceinline__ double feq(double rho, double w, double ex, double ey, double vx, double vy)
{
    return rho * w * (1 + (3. * (ex * vx + ey * vy) + 4.5 * (ex * vx + ey * vy) * (ex * vx + ey * vy) -
                           1.5 * (vx * vx + vy * vy)));
}

// Collision + Guo forcing on f_tot.
//   SRT: calRKCollision1TotalGPU2DSRTM A:1804-1848, calPerturbationFromForce2D A:1743-1798
//   MRT: calRKCollision1TotalGPU2DMRTM A:1938-2017, calPerturbationFromForce2DMRT A:2027-2113
//        S = diag(0,1.64,1.54,0,1.9,0,1.9,1/tau,1/tau) with tau evaluated per node.
// The reference multiplies (f - f_eq) and the source term by M, S and M^-1 as dense 9x9
// loops; here the same algebra is done in moment space: M f_eq and M src are known in
// closed form for this basis (verified against the matrices in tests/test_lattice.py),
//   m_eq  = rho (1, -2+3u^2, 1-3u^2, ux, -ux, uy, -uy, ux^2-uy^2, ux uy)
//   m_src = (0, 6u.F, -6u.F, Fx, -Fx, Fy, -Fy, 2(uxFx-uyFy), uxFy+uyFx)
// and M^-1 = M^T diag(1/|row|^2).  Differences to the dense form are rounding-level.
template <bool MRT>
__device__ __forceinline__ void collide_tau(double tau, double fT[9], double rho, double vx, double vy, double Fx, double Fy);
template <bool MRT>
__device__ __forceinline__ void collide(const RKDev &p, double fT[9], double rhoR, double rhoB, double phi,
                                        double vx, double vy, double Fx, double Fy)
{
    collide_tau<MRT>(tau_of(p, phi, rhoR, rhoB), fT, rhoR + rhoB, vx, vy, Fx, Fy);
}
template <bool MRT>
__device__ __forceinline__ void collide_tau(double tau, double fT[9], double rho, double vx, double vy, double Fx, double Fy)
{
    const double usq = vx * vx + vy * vy;
    if (!MRT) {
        constexpr double W[9] = LBMPM_D2Q9_W;
        constexpr int EXi[9] = LBMPM_D2Q9_EX, EYi[9] = LBMPM_D2Q9_EY;
        const double om = 1. / tau, sf = 1. - 1. / (2. * tau);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double ex = EXi[i], ey = EYi[i];
            const double eu = ex * vx + ey * vy;
            const double eT = rho * W[i] * (1. + (3. * eu + 4.5 * eu * eu - 1.5 * usq));
            const double src = W[i] * ((3. * (ex - vx) + 9. * ex * eu) * Fx + (3. * (ey - vy) + 9. * ey * eu) * Fy) * sf;
            fT[i] = (fT[i] - om * (fT[i] - eT)) + src;
        }
    } else {
        const double f0 = fT[0], f1 = fT[1], f2 = fT[2], f3 = fT[3], f4 = fT[4], f5 = fT[5], f6 = fT[6],
                     f7 = fT[7], f8 = fT[8];
        const double sa = (f1 + f2) + (f3 + f4), sd = (f5 + f6) + (f7 + f8);
        // non-conserved moments of f_tot (rows 1,2,4,6,7,8 of M; rows 0,3,5 carry S = 0)
        const double m1 = -4. * f0 - sa + 2. * sd;
        const double m2 = 4. * f0 - 2. * sa + sd;
        const double m4 = -2. * (f1 - f3) + ((f5 - f6) - (f7 - f8));
        const double m6 = -2. * (f2 - f4) + ((f5 + f6) - (f7 + f8));
        const double m7 = (f1 - f2) + (f3 - f4);
        const double m8 = (f5 - f6) + (f7 - f8);
        const double uF = vx * Fx + vy * Fy;
        const double s7 = 1. / tau;
        // r = -S (m - m_eq) + (1 - S/2) m_src, pre-divided by |row|^2
        const double r1 = ((1. - 0.5 * 1.64) * (6. * uF) - 1.64 * (m1 - rho * (-2. + 3. * usq))) * (1. / 36.);
        const double r2 = ((1. - 0.5 * 1.54) * (-6. * uF) - 1.54 * (m2 - rho * (1. - 3. * usq))) * (1. / 36.);
        const double r3 = Fx * (1. / 6.);
        const double r4 = ((1. - 0.5 * 1.9) * (-Fx) - 1.9 * (m4 + rho * vx)) * (1. / 12.);
        const double r5 = Fy * (1. / 6.);
        const double r6 = ((1. - 0.5 * 1.9) * (-Fy) - 1.9 * (m6 + rho * vy)) * (1. / 12.);
        const double r7 = ((1. - 0.5 * s7) * (2. * (vx * Fx - vy * Fy)) - s7 * (m7 - rho * (vx * vx - vy * vy))) * 0.25;
        const double r8 = ((1. - 0.5 * s7) * (vx * Fy + vy * Fx) - s7 * (m8 - rho * (vx * vy))) * 0.25;
        // f += M^T r
        const double a = -r1 - 2. * r2, d = 2. * r1 + r2;
        fT[0] = f0 + (-4. * r1 + 4. * r2);
        fT[1] = f1 + (a + r3 - 2. * r4 + r7);
        fT[2] = f2 + (a + r5 - 2. * r6 - r7);
        fT[3] = f3 + (a - r3 + 2. * r4 + r7);
        fT[4] = f4 + (a - r5 + 2. * r6 - r7);
        fT[5] = f5 + (d + r3 + r4 + r5 + r6 + r8);
        fT[6] = f6 + (d - r3 - r4 + r5 + r6 - r8);
        fT[7] = f7 + (d - r3 - r4 - r5 - r6 + r8);
        fT[8] = f8 + (d + r3 + r4 - r5 - r6 - r8);
    }
}

// calRecoloringProcessM, A:1857-1899.  cos(phi_i)|e_i| = (e_i . G)/|G| for every direction
// (the |e_i| factors cancel), so one reciprocal replaces the reference's eight divisions.
__device__ __forceinline__ void recolor(double beta, const double fT[9], double rhoR, double rhoB, double gx,
                                        double gy, double fR[9], double fB[9])
{
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EXi[9] = LBMPM_D2Q9_EX, EYi[9] = LBMPM_D2Q9_EY;
    const double gn = sqrt(gx * gx + gy * gy);
    const double itot = 1. / (rhoR + rhoB);
    const double kR = rhoR * itot, kB = rhoB * itot;
    const double A = (gn > 1.0e-8) ? beta * rhoR * rhoB * itot / gn : 0.;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double a = A * W[i] * edot(EXi[i], EYi[i], gx, gy);
        fR[i] = kR * fT[i] + a;
        fB[i] = kB * fT[i] - a;
    }
}

// ---------------------------------------------------------------- gradient / wetting / force
// updateColorGradientOnWetting (type 1, A:1639-1679) and ...New (type 2, A:2430-2492)
__device__ __forceinline__ void wetting_fix(const RKDev &p, double nsx, double nsy, double &gx, double &gy)
{
    const double nrm = sqrt(gx * gx + gy * gy);
    if (p.wetting == 1) {
        const double n1x = nsx * p.cosT - nsy * p.sinT, n1y = nsy * p.cosT + nsx * p.sinT;
        const double n2x = nsx * p.cosT + nsy * p.sinT, n2y = nsy * p.cosT - nsx * p.sinT;
        double ux = 0., uy = 0.;
        if (nrm > 1.0e-8) { ux = gx / nrm; uy = gy / nrm; }
        const double dx1 = ux - n1x, dy1 = uy - n1y, dx2 = ux - n2x, dy2 = uy - n2y;
        const double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
        double mx = 0., my = 0.;
        if (d1 < d2) { mx = n1x; my = n1y; }
        else if (d1 > d2) { mx = n2x; my = n2y; }
        else if (d1 == d2) { mx = nsx; my = nsy; }
        gx = nrm * mx; gy = nrm * my;
    } else if (p.wetting == 2) {
        double ux = 0., uy = 0.;
        if (nrm > 1.0e-8) { ux = -gx / nrm; uy = -gy / nrm; }
        const double ang = ux * nsx + uy * nsy;
#ifdef LBMPM_RELAXED   // measurement build only (tools/relaxed_parity.py): cos(acos a) = a, sin(acos a) = sqrt(1 - a^2), no libm calls
        const double cth = ang, sth = sqrt(1. - ang * ang);
#else
        const double th = acos(ang);
        const double sth = sin(th), cth = cos(th);
#endif
        double c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
        if (fabs(sth) > 1.0e-9) {
            c1 = p.sinT * cth / sth;
            c2 = p.sinT / sth;
            c3 = -p.sinT * cth / sth;
            c4 = -p.sinT / sth;
        }
        const double nx1 = (p.cosT - c1) * nsx + c2 * ux, ny1 = (p.cosT - c1) * nsy + c2 * uy;
        const double nx2 = (p.cosT - c3) * nsx + c4 * ux, ny2 = (p.cosT - c3) * nsy + c4 * uy;
        const double dx1 = nx1 - ux, dy1 = ny1 - uy, dx2 = nx2 - ux, dy2 = ny2 - uy;
        const double d1 = sqrt(dx1 * dx1 + dy1 * dy1), d2 = sqrt(dx2 * dx2 + dy2 * dy2);
        if (d1 < d2) { gx = -nrm * nx1; gy = -nrm * ny1; }
        else if (d1 > d2) { gx = -nrm * nx2; gy = -nrm * ny2; }
    }
}

// unit normal used by the curvature stencil: type 2 -> -G/|G| above 1e-8 (A:2512-2520),
// type 1 -> +G/|G| above 0 (A:1703-1708)
__device__ __forceinline__ void unit_normal(int wetting, double gx, double gy, double &ux, double &uy)
{
    const double n = sqrt(gx * gx + gy * gy);
    ux = 0.; uy = 0.;
    if (wetting == 2) { if (n > 1.0e-8) { const double r = -1. / n; ux = gx * r; uy = gy * r; } }
    else { if (n > 0.) { const double r = 1. / n; ux = gx * r; uy = gy * r; } }
}

// ---------------------------------------------------------------- kernels (split schedule)
constexpr int BX = 64, BY = 4;

// K1: phase field of the post-streaming, post-BC state (calPhaseFieldPhi A:1348)
__global__ __launch_bounds__(BX *BY) void rk2d_phase_field(RKDev p)
{
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[9], fB[9], rR, rB;
    node_state<true>(p, x, y, fR, fB, rR, rB);
    p.phi[idx] = (rR - rB) / (rR + rB);
}

// K2: colour gradient with wetting correction
//   calColorValueOnSolid A:1560-1581 (evaluated on the fly for solid neighbours),
//   calRKInitialGradient A:1584-1634, updateColorGradientOnWetting[New]
__global__ __launch_bounds__(BX *BY) void rk2d_gradient(RKDev p)
{
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const unsigned sn = p.solidnbr[idx];
    double gx = 0., gy = 0.;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int xn = wrapi(x + EX[i], p.nx), yn = wrapi(y + EY[i], p.ny);
        const size_t nidx = (size_t)yn * p.pitch + xn;
        double v;
        if (!((sn >> (i - 1)) & 1u)) {
            v = p.phi[nidx];
        } else {   // wetting solid: weighted mean of phi over its fluid neighbours
            const unsigned ss = p.solidnbr[nidx];
            double sum = 0., sw = 0.;
#pragma unroll
            for (int j = 1; j < 9; ++j) {
                if (!((ss >> (j - 1)) & 1u)) {
                    const int x2 = wrapi(xn + EX[j], p.nx), y2 = wrapi(yn + EY[j], p.ny);
                    sum += W[j] * p.phi[(size_t)y2 * p.pitch + x2];
                    sw += W[j];
                }
            }
            v = sum / sw;
        }
        gx += W[i] * v * (double)EX[i];
        gy += W[i] * v * (double)EY[i];
    }
    gx = 3. * gx; gy = 3. * gy;
    if (sn != 0) wetting_fix(p, p.ns[idx], p.ns[p.plane + idx], gx, gy);
    p.G[idx] = gx;
    p.G[p.plane + idx] = gy;
}

// K3: everything else of the step (dominant kernel): stream+BC, u, curvature + CSF force,
// collision + forcing, recolouring, store post-collision populations.
template <bool MRT>
__global__ __launch_bounds__(BX *BY) void rk2d_collide_stream(RKDev p)
{
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[9], fB[9], rR, rB;
    node_state<true>(p, x, y, fR, fB, rR, rB);
    double fT[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fT[i] = fR[i] + fB[i];          // calTotalFluidPDF A:1414
    // calPhysicalVelocityRKGPU2DNew1 A:2634-2654 (force of the PREVIOUS step)
    const double rs = rB + rR;
    const double vx = (fT[1] - fT[3] + fT[5] - fT[6] - fT[7] + fT[8] + 0.5 * p.F[idx]) / rs;
    const double vy = (fT[2] - fT[4] + fT[5] + fT[6] - fT[7] - fT[8] + 0.5 * p.F[p.plane + idx]) / rs;
    const double phi = (rR - rB) / (rR + rB);
    // calForceTermInColorGradient[New]2D A:1686-1736 / A:2499-2551
    const double gx = p.G[idx], gy = p.G[p.plane + idx];
    double ux, uy;
    unit_normal(p.wetting, gx, gy, ux, uy);
    const unsigned sn = p.solidnbr[idx];
    double pyx = 0., pxy = 0., px = 0., py = 0.;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        if ((sn >> (i - 1)) & 1u) continue;
        const int xn = wrapi(x + EX[i], p.nx), yn = wrapi(y + EY[i], p.ny);
        const size_t nidx = (size_t)yn * p.pitch + xn;
        double qx, qy;
        unit_normal(p.wetting, p.G[nidx], p.G[p.plane + nidx], qx, qy);
        pyx += 3. * W[i] * qy * (double)EX[i];
        pxy += 3. * W[i] * qx * (double)EY[i];
        px += 3. * W[i] * qx * (double)EX[i];
        py += 3. * W[i] * qy * (double)EY[i];
    }
    const double K = ux * uy * (pyx + pxy) - uy * uy * px - ux * ux * py;
    const double sgn = (p.wetting == 2) ? -0.5 : 0.5;
    const double Fx = sgn * p.sigma * K * gx, Fy = sgn * p.sigma * K * gy;
    p.F[idx] = Fx;
    p.F[p.plane + idx] = Fy;
    if (p.diag) { p.diag[idx] = vx; p.diag[p.plane + idx] = vy; p.diag[2 * p.plane + idx] = K; }
    collide<MRT>(p, fT, rR, rB, phi, vx, vy, Fx, Fy);
    recolor(p.beta, fT, rR, rB, gx, gy, fR, fB);
    lbmpm_dev::store_pairs<true>(p.fout, p.plane, idx, fR, fB);
}

// ---------------------------------------------------------------- fused schedule
// One kernel per time step.  A workgroup owns a TW x TH tile and recomputes the phase
// field on the tile + 3-node halo (the CSF force at a node needs n^ at distance 1, hence G,
// hence phi / phi_solid at distance 2, hence phi at distance 3 next to wetting solids):
//   A  every fluid node of tile+halo: pull + boundary rows -> rhoR, rhoB -> phi -> LDS
//      (owners of interior nodes keep f_tot, rhoR, rhoB in registers)
//   B  solid nodes of tile+2: phi_s = weighted mean over fluid neighbours -> LDS (in place)
//   C  fluid nodes of tile+1: G (+ wetting fix), unit normal n^ -> LDS
//   D  interior: curvature, CSF force, u, collision + forcing, recolouring, store.
// The halo re-reads hit the XCD's L2 (tiles are handed to XCDs in contiguous bands).
template <int TH_, int NT_>
struct FusedShape {
    static constexpr int TW = 64, TH = TH_, NT = NT_, H = 3;
    static constexpr int TY = TH / NT;            // thread rows
    static constexpr int THREADS = TW * TY;
    static constexpr int RW = TW + 2 * H, RH = TH + 2 * H;
};

// Tracer sub-step of the coupled loop (Transport2DRK.py:1341-1418) for node (x,y), given the
// flow's rhoR, physical velocity and wetting-corrected colour gradient of this step:
//   pull-stream g (calStreamingTransportGPU/2GPU T:139-194, D2Q5 order 0,E,W,N,S; free outlet
//   T:461-478 = "row 0 reads as row 1"), Inamuro inlet on the ghost row (T:682-698),
//   C = sum g (T:78-90), indicator (T:957-970), MRT collision g += A (M g - M g_eq) (T:535-590),
//   interface term (T:976-1013); store post-collision g.
__device__ __forceinline__ void tracer_substep(const RKDev &p, int x, int y, unsigned sn, double rhoR, double vx,
                                               double vy, double gx, double gy)
{
    constexpr double W5[5] = {1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.};
    constexpr int VX5[5] = {0, 1, -1, 0, 0}, VY5[5] = {0, 0, 0, 1, -1}, OPP5[5] = {0, 2, 1, 4, 3};
    constexpr int SRCBIT[5] = {0, 2, 0, 3, 1};    // D2Q9 solid bit of the node a D2Q5 population comes from
    const bool first = p.first != 0;
    const int yo = (p.trFree && y == 0 && !first) ? 1 : y;          // own-node reads
    // (byte offsets inside a plane of doubles, 32 bits; the plane bases are uniform: as the populations' pulls in d2q9_device.h)
    const unsigned own = ((unsigned)yo * (unsigned)p.pitch + (unsigned)x) * 8u, idx = ((unsigned)y * (unsigned)p.pitch + (unsigned)x) * 8u;
    const size_t plane8 = p.plane * 8u;
    const double gn = sqrt(gx * gx + gy * gy);
    double ux = 0., uy = 0., un = 0.;
    if (gn > 1.0e-8) { ux = -gx / gn; uy = -gy / gn; un = sqrt(ux * ux + uy * uy); }
    const double ind = (rhoR > p.trCrit) ? -(1. - 1.) : -(1. - 0.);
    // streamed, inlet-corrected populations of tracer t at this node
    auto pull_g = [&](int t, double g[5]) {
        const char *gi = reinterpret_cast<const char *>(p.gin) + (size_t)t * 5 * plane8;
        auto ld = [&](int j, unsigned off) { return *reinterpret_cast<const double *>(gi + (size_t)j * plane8 + off); };
        g[0] = ld(0, own);
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            if (first) { g[j] = ld(j, idx); continue; }
            if ((sn >> SRCBIT[j]) & 1u) { g[j] = ld(OPP5[j], own); continue; }     // bounce-back
            int ys = lbmpm_dev::wrapi(y - VY5[j], p.ny);
            if (p.trFree && ys == 0) ys = 1;
            g[j] = ld(j, ((unsigned)ys * (unsigned)p.pitch + (unsigned)lbmpm_dev::wrapi(x - VX5[j], p.nx)) * 8u);
        }
        if (!first && p.trDirichlet && y == p.ny - 1) {
            const double sm = g[0] + g[1] + g[2] + g[3];
            const double u = (p.trCb[t] - sm) / W5[4];
            g[4] = W5[4] * u;
        }
    };
    // reaction A + B -> C (calReactionTracersGPU T:95-111): source k C_0 C_1, needs the concentrations of
    // tracers 0 and 1 before any tracer is updated (their populations are pulled again in the loop: L1 hits)
    double src = 0.;
    if (p.trRate != 0.) {
        double ga[5], gb[5];
        pull_g(0, ga); pull_g(1, gb);
        double ca = 0., cb = 0.;
#pragma unroll
        for (int j = 0; j < 5; ++j) { ca += ga[j]; cb += gb[j]; }
        src = p.trRate * ca * cb;
    }
    for (int t = 0; t < p.ntr; ++t) {
        double g[5];
        pull_g(t, g);
        double C = 0.;
#pragma unroll
        for (int j = 0; j < 5; ++j) C += g[j];
        double diff[5], d[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            double ve = 0., vp = 0.;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                // (the D2Q5 moment matrix of lbmpm_rk2d_tracer_configure as constants: the products by 0 and 1 fold away -- adding
                // 0 * x to a finite sum changes nothing -- and 25 doubles leave the scalar registers)
                constexpr double M5[25] = {1, 1, 1, 1, 1,   0, 1, -1, 0, 0,   0, 0, 0, 1, -1,   4, -1, -1, -1, -1,   0, 1, 1, -1, -1};
                if (M5[5 * j + k] == 0.) continue;
                const double eq = C * W5[k] * (1. + 3. * edot(VX5[k], VY5[k], vx, vy));
                ve += M5[5 * j + k] * eq;
                vp += g[k] * M5[5 * j + k];
            }
            diff[j] = vp - ve;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            double v = 0.;
#pragma unroll
            for (int k = 0; k < 5; ++k) v += p.trA[t][5 * j + k] * diff[k];
            d[j] = v;
        }
        char *go = reinterpret_cast<char *>(p.gout) + (size_t)t * 5 * plane8;
        const double S = (p.trRate != 0.) ? (t == 2 ? src : -src) : 0.;        // tracers 0, 1 consumed, 2 produced
        const double J0 = p.trJ[t], J1 = (1. - p.trJ[t]) / 4.;
        // (streaming stores like the populations': read again a whole lattice later)
        __builtin_nontemporal_store(p.trRate != 0. ? (g[0] + d[0]) + J0 * S : g[0] + d[0], reinterpret_cast<double *>(go + idx));
        // cos of the angle between direction j and the interface normal, (e_j . u) / |u| (T:995-1008): e_j is a unit axis vector, so the four
        // quotients are +-u_x / |u| and +-u_y / |u| -- two divisions instead of four, the same bits ((-a) / b == -(a / b); 1 * a + 0 * b == a)
        double cax = 0., cay = 0.;
        if (un > 1.0e-8) { cax = ux / (1. * un); cay = uy / (1. * un); }
#pragma unroll
        for (int j = 1; j < 5; ++j) {
            const double c = VX5[j] != 0 ? (VX5[j] > 0 ? cax : -cax) : (VY5[j] > 0 ? cay : -cay);
            const double v = (g[j] + d[j]) + p.trBeta[t] * ind * (W5[j] * C) * c;
            __builtin_nontemporal_store((p.trRate != 0.) ? v + J1 * S : v, reinterpret_cast<double *>(go + (size_t)j * plane8 + idx));
        }
    }
}

// TR (tracer variant): the order of the density sums on the rows a boundary kernel touches (node_finish).  Rows without a boundary
// rule come out the same for TR = 0 and 1, so the tracer step takes TR = 1 (the re-summing variant) only on the tile rows whose
// region (tile + 3) holds one of the lattice rows 0, 1, ny-2, ny-1 and TR = 0 on the rest: rk2d_fused_tracer below.
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
// tools/dev/phases2d.py: cycle counter at the phase borders of rk2d_fused, kept per wave and added to one of 256 slots at the end
__device__ unsigned long long rk2d_ph[256 * 16];
#define PH2(k) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_acc[k] = now_ - ph_t; ph_t = now_; }
#else
#define PH2(k)
#endif
// the step of tile t; s_*: the workgroup's LDS arrays of RH x RW entries each
#if defined(LBMPM_DEV) && defined(RK2D_KO_BC)      // timing knock-out (tools/dev/ab2d.py): no boundary rows at all -- what an interior-tile instance could save
#define RK2D_BC false
#else
#define RK2D_BC true
#endif
template <bool MRT, bool TRACER, typename SH, int TR>
__device__ __forceinline__ void rk2d_fused_tile(const RKDev &p, int tiles_x, int t, double *s_phi, double *s_ux, double *s_uy, double *s_gx, double *s_gy,
                                                uint16_t *s_list, int *s_cnt, uint8_t *s_fluid)
{
    constexpr int TW = SH::TW, TH = SH::TH, NT = SH::NT, H = SH::H, TY = SH::TY, THREADS = SH::THREADS;
    constexpr int RW = SH::RW, RH = SH::RH;
#ifdef RK2D_NO_EAGER
    constexpr bool EAGER = false;
#else
    constexpr bool EAGER = (NT == 1);
#endif
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    const int tid = threadIdx.x, lx = tid % TW, ly = tid / TW;

#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
    unsigned long long ph_acc[12] = {};
    unsigned long long ph_t = __builtin_readcyclecounter();
#endif
    if (tid == 0) *s_cnt = 0;                  // the queue of phase C (three barriers from here)
    constexpr int NHALO = 2 * H * RW + 2 * H * TH;
    auto halo_cell = [&](int n, int &rx, int &ry) {
        int mloc = n;
        if (mloc < H * RW) { ry = mloc / RW; rx = mloc % RW; }
        else if ((mloc -= H * RW) < H * RW) { ry = RH - H + mloc / RW; rx = mloc % RW; }
        else { mloc -= H * RW; ry = H + mloc / (2 * H); const int c = mloc % (2 * H); rx = c < H ? c : RW - 2 * H + c; }
    };
    double fT[NT][9], rR[NT], rB[NT], Fpx[NT], Fpy[NT];
    unsigned sn[NT];
    bool act[NT];
    int hrx = 0, hry = 0, hx = 0, hy = 0, hys = 0;          // the first halo node of this thread (all of them when NHALO <= THREADS)
    bool hdo = false;
    double hR[9], hB[9];
    bool need3;
    if constexpr (EAGER) {
        // ---- phase A, own node first: the loads of the region's fluid mask and, with no flag known yet, the own node's solid-neighbour
        // byte, lagged force and nine population pairs go out together (a lane on a solid node reads finite junk it never uses), so the
        // volume of the own pulls moves through the CU's address path while the mask is on its way -- the halo pulls, which need the mask,
        // follow the barrier as before, with half the burst.  asm loads + hand-counted waits: written in C++ hipcc consumes the early pulls
        // one by one as they arrive to save registers (13 % slower than not hoisting them at all).
        using namespace lbmpm_dev;
        constexpr int NFL = (RH * RW + THREADS - 1) / THREADS;
        static_assert(NT == 1, "one own node per lane");
        unsigned flr[NFL];
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int n = min(tid + k * THREADS, RH * RW - 1);
            const int rx = n % RW, ry = n / RW;
            const int x = wrapm(tx0 - H + rx, p.nx), y = wrapm(ty0 - H + ry, p.ny);
            flr[k] = asm_ldu8(p.flags, (unsigned)y * (unsigned)p.pitch + (unsigned)x);
        }
        lbmpm_d2 q[9], hq[9];
        unsigned hsn = 0;
        const int x = tx0 + lx, y = ty0 + ly;
        const bool inside = (x < p.nx) && (y < p.ny);
        // nodes beyond the lattice edge of a partial tile are periodic images of real nodes and serve as halo for the valid part of the tile
        const int xw = inside ? x : wrapm(x, p.nx), yw = inside ? y : wrapm(y, p.ny);
        const int ys = node_source_row<RK2D_BC>(p, yw);
        {
            const size_t idx = (size_t)yw * p.pitch + xw;
            sn[0] = asm_ldu8(p.solidnbr, (unsigned)idx);
            Fpx[0] = asm_ld8_nt(p.F, (unsigned)idx * 8u);                  // read once, by its own node
            Fpy[0] = asm_ld8_nt(p.F + p.plane, (unsigned)idx * 8u);
            pull_issue_asm(p, xw, ys, q);
        }
        __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(12));          // the mask has landed; the own node's 12 loads may still be out
        bool any_solid = false;
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            asm volatile("" : "+v"(flr[k]));
            const int n = tid + k * THREADS;
            if (n < RH * RW) {
                const int rx = n % RW, ry = n / RW;
                const uint8_t fl = flr[k] & 1;
                s_fluid[n] = fl;
                if (!fl && rx >= 1 && rx < RW - 1 && ry >= 1 && ry < RH - 1) any_solid = true;
            }
        }
        need3 = __syncthreads_or(any_solid);   // also publishes s_fluid
        PH2(0)
        const bool first = p.first != 0;
        if (tid < NHALO) {
            halo_cell(tid, hrx, hry);
            hdo = s_fluid[hry * RW + hrx] && (need3 || !(hrx == 0 || hrx == RW - 1 || hry == 0 || hry == RH - 1));
        }
        if (__ballot(hdo) != 0ull) {               // wave-uniform: the count of loads in flight must be known
            if (hdo) {
                LBMPM_TAKEN;                       // (a lane of this wave is here: openlbmpm_amd/inflight.py drops hipcc's all-lanes-off branch)
                hx = wrapm(tx0 - H + hrx, p.nx); hy = wrapm(ty0 - H + hry, p.ny);
                hys = node_source_row<RK2D_BC>(p, hy);
                hsn = asm_ldu8(p.solidnbr, (unsigned)hys * (unsigned)p.pitch + (unsigned)hx);
                pull_issue_asm(p, hx, hys, hq);
            }
            __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(10));      // the own node's loads have landed, the halo node's 10 are out
        } else __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(0));
        asm volatile("" : "+v"(sn[0]), "+v"(Fpx[0]), "+v"(Fpy[0]));
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(q[i]));
        {
            const int ri = (H + ly) * RW + H + lx;
            const bool fluid = s_fluid[ri];
            act[0] = inside && fluid;
            if (!fluid) { sn[0] = 0; Fpx[0] = 0.; Fpy[0] = 0.; }
            if (fluid) {
                double fR[9], fB[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) { fR[i] = q[i].x; fB[i] = q[i].y; }
                const unsigned psn = first ? 0u : (ys == yw ? sn[0] : (unsigned)p.solidnbr[(size_t)ys * p.pitch + xw]);
                pull_patch(p, xw, ys, psn, fR, fB);
                node_finish<RK2D_BC, TR>(p, yw, ys, fR, fB, rR[0], rB[0]);
#pragma unroll
                for (int i = 0; i < 9; ++i) fT[0][i] = fR[i] + fB[i];
                s_phi[ri] = (rR[0] - rB[0]) / (rR[0] + rB[0]);
            }
        }
        __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(0));
        asm volatile("" : "+v"(hsn));
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(hq[i]));
        if (hdo) {
#pragma unroll
            for (int i = 0; i < 9; ++i) { hR[i] = hq[i].x; hB[i] = hq[i].y; }
            pull_patch(p, hx, hys, first ? 0u : hsn, hR, hB);
            double a, c;
            node_finish<RK2D_BC, TR>(p, hy, hys, hR, hB, a, c);
            s_phi[hry * RW + hrx] = (a - c) / (a + c);
        }
    } else {
    // fluid mask of the region (issued first so that the wait for it leaves the population
    // loads below in flight)
    bool any_solid = false;
    for (int n = tid; n < RH * RW; n += THREADS) {
        const int rx = n % RW, ry = n / RW;
        const int x = wrapm(tx0 - H + rx, p.nx), y = wrapm(ty0 - H + ry, p.ny);
        const uint8_t fl = p.flags[(size_t)y * p.pitch + x] & 1;
        s_fluid[n] = fl;
        if (!fl && rx >= 1 && rx < RW - 1 && ry >= 1 && ry < RH - 1) any_solid = true;
    }

    // ---- phase A: the node's own pull and the pull of "its" halo node are issued back to back -- one memory round trip, not two
    need3 = __syncthreads_or(any_solid);   // also publishes s_fluid
    PH2(0)
    // the first halo node of this thread (all of them when NHALO <= THREADS)
    if (tid < NHALO) {
        halo_cell(tid, hrx, hry);
        hdo = s_fluid[hry * RW + hrx] && (need3 || !(hrx == 0 || hrx == RW - 1 || hry == 0 || hry == RH - 1));
        if (hdo) {
            hx = wrapm(tx0 - H + hrx, p.nx); hy = wrapm(ty0 - H + hry, p.ny);
            hys = node_source_row<RK2D_BC>(p, hy);
            pull_node(p, hx, hys, hR, hB);
        }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        const int x = tx0 + lx, y = ty0 + ly + m * TY;
        const int ri = (H + ly + m * TY) * RW + H + lx;
        const bool inside = (x < p.nx) && (y < p.ny);
        // nodes beyond the lattice edge of a partial tile are periodic images of real nodes
        // and serve as halo for the valid part of the tile
        const int xw = inside ? x : wrapm(x, p.nx), yw = inside ? y : wrapm(y, p.ny);
        const size_t idx = (size_t)yw * p.pitch + xw;
        const bool fluid = p.flags[idx] & 1;
        act[m] = inside && fluid;
        sn[m] = 0; Fpx[m] = 0.; Fpy[m] = 0.;
        if (fluid) {
            double fR[9], fB[9];
            sn[m] = p.solidnbr[idx];
            Fpx[m] = __builtin_nontemporal_load(p.F + idx);          // read once, by its own node
            Fpy[m] = __builtin_nontemporal_load(p.F + p.plane + idx);
            node_state<RK2D_BC, TR>(p, xw, yw, fR, fB, rR[m], rB[m]);
#pragma unroll
            for (int i = 0; i < 9; ++i) fT[m][i] = fR[i] + fB[i];
            s_phi[ri] = (rR[m] - rB[m]) / (rR[m] + rB[m]);
        }
    }
    if (hdo) {
        double a, c;
        node_finish<RK2D_BC, TR>(p, hy, hys, hR, hB, a, c);
        s_phi[hry * RW + hrx] = (a - c) / (a + c);
    }
    }
    for (int n = tid + THREADS; n < NHALO; n += THREADS) {          // shapes whose halo outnumbers the threads
        int rx, ry;
        halo_cell(n, rx, ry);
        const int ri = ry * RW + rx;
        if (!s_fluid[ri]) continue;
        if (!need3 && (rx == 0 || rx == RW - 1 || ry == 0 || ry == RH - 1)) continue;
        const int x = wrapm(tx0 - H + rx, p.nx), y = wrapm(ty0 - H + ry, p.ny);
        double fR[9], fB[9], a, c;
        node_state<RK2D_BC, TR>(p, x, y, fR, fB, a, c);
        s_phi[ri] = (a - c) / (a + c);
    }
    PH2(1)
    __syncthreads();
    PH2(2)

    // ---- phase B: colour value on wetting solids (calColorValueOnSolid A:1560-1581)
    if (need3) {
        for (int n = tid; n < (RH - 2) * (RW - 2); n += THREADS) {
            const int rx = 1 + n % (RW - 2), ry = 1 + n / (RW - 2);
            const int ri = ry * RW + rx;
            if (s_fluid[ri]) continue;
            double sum = 0., sw = 0.;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                const int rn = ri + EY[i] * RW + EX[i];
                if (s_fluid[rn]) { sum += W[i] * s_phi[rn]; sw += W[i]; }
            }
            if (sw > 0.) s_phi[ri] = sum / sw;
        }
        __syncthreads();
    }
    PH2(3)

    // ---- phase C: colour gradient, wetting correction, unit normal
    // C1: every fluid node of tile + 1 gets its gradient; a node without solid neighbours gets its unit normal at once, one with is
    //     queued (raw gradient in s_gx / s_gy, its region index in s_list).
    // C2: the queue is worked off by as many lanes as it has entries.  The wetting correction is some 450 fp64 instructions (acos, sin,
    //     cos + four square roots); done in place it is paid by every wave that holds ONE node next to a solid -- in a porous medium all
    //     of them, with a third of their lanes.  Same arithmetic per node, bit for bit.
    auto raw_gradient = [&](int ri, double &gx, double &gy) -> bool {
        double ax = 0., ay = 0.;
        bool solid_nb = false;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const int rn = ri + EY[i] * RW + EX[i];
            const double v = s_phi[rn];
            solid_nb |= !s_fluid[rn];
            if (EX[i] != 0) ax += pm(EX[i], W[i] * v);         // (no "+ W v * 0": see pm / edot)
            if (EY[i] != 0) ay += pm(EY[i], W[i] * v);
        }
        gx = 3. * ax; gy = 3. * ay;
        if (solid_nb) {
            s_gx[ri] = gx; s_gy[ri] = gy;
            s_list[atomicAdd(s_cnt, 1)] = (uint16_t)ri;
        } else {
            double ux, uy;
            unit_normal(p.wetting, gx, gy, ux, uy);
            s_ux[ri] = ux; s_uy[ri] = uy;
        }
        return solid_nb;
    };
    double gx[NT], gy[NT];
    bool wet[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        const int ri = (H + ly + m * TY) * RW + H + lx;
        gx[m] = 0.; gy[m] = 0.; wet[m] = false;
        if (s_fluid[ri]) wet[m] = raw_gradient(ri, gx[m], gy[m]);
    }
    constexpr int NRING = 2 * (TW + 2) + 2 * TH;
    for (int n = tid; n < NRING; n += THREADS) {
        int rx, ry, mloc = n;
        if (mloc < TW + 2) { ry = H - 1; rx = H - 1 + mloc; }
        else if ((mloc -= TW + 2) < TW + 2) { ry = H + TH; rx = H - 1 + mloc; }
        else { mloc -= TW + 2; ry = H + mloc / 2; rx = (mloc & 1) ? H + TW : H - 1; }
        const int ri = ry * RW + rx;
        if (!s_fluid[ri]) continue;
        double a, c;
        raw_gradient(ri, a, c);
    }
    if (need3) __syncthreads();            // (a region without solids queues nothing: no barrier, an empty loop)
    for (int k = tid, nq = need3 ? *s_cnt : 0; k < nq; k += THREADS) {
        const int ri = s_list[k];
        const int x = wrapm(tx0 - H + ri % RW, p.nx), y = wrapm(ty0 - H + ri / RW, p.ny);
        const size_t idx = (size_t)y * p.pitch + x;
        double a = s_gx[ri], c = s_gy[ri], ux, uy;
        wetting_fix(p, p.ns[idx], p.ns[p.plane + idx], a, c);
        unit_normal(p.wetting, a, c, ux, uy);
        s_ux[ri] = ux; s_uy[ri] = uy;
        s_gx[ri] = a; s_gy[ri] = c;
    }
    PH2(4)
    __syncthreads();
    PH2(5)

    // ---- phase D: force, velocity, collision, recolouring, store
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        const int x = tx0 + lx, y = ty0 + ly + m * TY;
        // non-fluid lanes of a 128-byte line that holds fluid write zeros into their dead slots
        // (8 lanes per line for the 16-byte population pairs, 16 for the 8-byte force planes)
        const bool line8 = lbmpm_dev::line_has_active<8>(act[m], tid & 63);
        if (!(lbmpm_dev::line_has_active<16>(act[m], tid & 63) && x < p.nx && y < p.ny)) continue;
        const size_t idx = (size_t)y * p.pitch + x;
        double fR[9], fB[9], Fx = 0., Fy = 0.;
#pragma unroll
        for (int i = 0; i < 9; ++i) { fR[i] = 0.; fB[i] = 0.; }
        if (act[m]) {
        const int ri = (H + ly + m * TY) * RW + H + lx;
        if (wet[m]) { gx[m] = s_gx[ri]; gy[m] = s_gy[ri]; }
        const double ux = s_ux[ri], uy = s_uy[ri];
        double pyx = 0., pxy = 0., px = 0., py = 0.;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            if ((sn[m] >> (i - 1)) & 1u) continue;
            const int rn = ri + EY[i] * RW + EX[i];
            const double qx = s_ux[rn], qy = s_uy[rn];
            if (EX[i] != 0) { pyx += pm(EX[i], 3. * W[i] * qy); px += pm(EX[i], 3. * W[i] * qx); }
            if (EY[i] != 0) { pxy += pm(EY[i], 3. * W[i] * qx); py += pm(EY[i], 3. * W[i] * qy); }
        }
        const double K = ux * uy * (pyx + pxy) - uy * uy * px - ux * ux * py;
        const double sgn = (p.wetting == 2) ? -0.5 : 0.5;
        Fx = sgn * p.sigma * K * gx[m]; Fy = sgn * p.sigma * K * gy[m];
        const double rs = rB[m] + rR[m];
        double *f = fT[m];
        const double vx = (f[1] - f[3] + f[5] - f[6] - f[7] + f[8] + 0.5 * Fpx[m]) / rs;
        const double vy = (f[2] - f[4] + f[5] + f[6] - f[7] - f[8] + 0.5 * Fpy[m]) / rs;
        const double phi = (rR[m] - rB[m]) / (rR[m] + rB[m]);
        if (p.diag) {
            p.diag[idx] = vx; p.diag[p.plane + idx] = vy; p.diag[2 * p.plane + idx] = K;
            p.phi[idx] = phi; p.G[idx] = gx[m]; p.G[p.plane + idx] = gy[m];
        }
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
        unsigned long long tr_t0 = __builtin_readcyclecounter();
#endif
        if (TRACER) tracer_substep(p, x, y, sn[m], rR[m], vx, vy, gx[m], gy[m]);
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
        ph_acc[8] += __builtin_readcyclecounter() - tr_t0;
#endif
        collide<MRT>(p, f, rR[m], rB[m], phi, vx, vy, Fx, Fy);
        recolor(p.beta, f, rR[m], rB[m], gx[m], gy[m], fR, fB);
        }
        __builtin_nontemporal_store(Fx, p.F + idx);
        __builtin_nontemporal_store(Fy, p.F + p.plane + idx);
        if (line8) {
            lbmpm_dev::store_pairs<true>(p.fout, p.plane, idx, fR, fB);
        }
    }
#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
    PH2(6)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): until the stores are acknowledged
    PH2(7)
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *slot = rk2d_ph + (blockIdx.x & 255) * 16;
        for (int k = 0; k < 12; ++k) atomicAdd(slot + k, ph_acc[k]);
        atomicAdd(slot + 15, 1ull);
    }
#endif
}
#undef PH2

template <bool MRT, bool TRACER, typename SH, int TR = (TRACER ? 1 : 0)>
__global__ __launch_bounds__(SH::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rk2d_fused(RKDev p, int tiles_x, int tile0 = 0)
{
    __shared__ double s_phi[SH::RH * SH::RW];
    __shared__ double s_ux[SH::RH * SH::RW];
    __shared__ double s_uy[SH::RH * SH::RW];
    __shared__ double s_gx[SH::RH * SH::RW];
    __shared__ double s_gy[SH::RH * SH::RW];
    __shared__ uint16_t s_list[(SH::TW + 2) * (SH::TH + 2)];
    __shared__ int s_cnt;
    __shared__ uint8_t s_fluid[SH::RH * SH::RW];
    // XCD-aware tile assignment: workgroup b runs on XCD b % 8 (observed dispatch order);
    // give every XCD a contiguous band of tiles so halo rows are shared inside one L2.
    rk2d_fused_tile<MRT, TRACER, SH, TR>(p, tiles_x, tile0 + xcd_tile(blockIdx.x, gridDim.x, tiles_x), s_phi, s_ux, s_uy, s_gx, s_gy, s_list, &s_cnt, s_fluid);
}

// The tracer step as ONE launch: tiles below lo_end and from hi_begin on (the tile rows whose region holds a lattice row a boundary rule
// touches, see launch_fused_tracer) take the re-summing variant TR = 1, the others TR = 0 -- a workgroup-uniform branch over two copies of
// the step.  As three launches the two boundary ones, a few dozen tiles each, cost a whole tile's latency chain apiece (2 x 16 us of c4's
// 0.42 ms) behind the interior launch.
template <bool MRT, typename SH>
__global__ __launch_bounds__(SH::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rk2d_fused_tracer(RKDev p, int tiles_x, int lo_end, int hi_begin)
{
    __shared__ double s_phi[SH::RH * SH::RW];
    __shared__ double s_ux[SH::RH * SH::RW];
    __shared__ double s_uy[SH::RH * SH::RW];
    __shared__ double s_gx[SH::RH * SH::RW];
    __shared__ double s_gy[SH::RH * SH::RW];
    __shared__ uint16_t s_list[(SH::TW + 2) * (SH::TH + 2)];
    __shared__ int s_cnt;
    __shared__ uint8_t s_fluid[SH::RH * SH::RW];
    const int t = xcd_tile(blockIdx.x, gridDim.x, tiles_x);
    if (t < lo_end || t >= hi_begin) rk2d_fused_tile<MRT, true, SH, 1>(p, tiles_x, t, s_phi, s_ux, s_uy, s_gx, s_gy, s_list, &s_cnt, s_fluid);
    else rk2d_fused_tile<MRT, true, SH, 0>(p, tiles_x, t, s_phi, s_ux, s_uy, s_gx, s_gy, s_list, &s_cnt, s_fluid);
}

// ---------------------------------------------------------------- perturbation operator, fused
// [SurfaceTension] SurfaceTensionType = 'Perturbation': the loop of RKColorGradientLBM.runRKColorGradient2DPerturbation
// (RKD2Q9.py:978-1223; the 2-D twin of the D3Q19 model of rk3d.hip) as ONE launch per time step.  Per step, in the reference's
// order: stream both colours (A:340-417) -> Zou-He pressure outlet per colour on row 1 (calConstPressureLowerGPU A:1008-1039),
// ghost row 0 = row 1 (A:1045-1081) -> Zou-He velocity inlet per colour on row ny-2 (constantVelocityZHBoundaryHigherRK A:657-695),
// ghost row ny-1 = row ny-2 (A:607-650) -> densities (A:80-100), velocity without force (A:125-147), phase field (A:1348) ->
// collision 1: BGK per colour (calRKCollision1GPU2DSRTNew A:1125-1163) or MRT on the sum (calRKCollision1GPU2DMRTNew A:1272-1343,
// body force zero), tau harmonic in phi -> colour gradient from the neighbours' phase field (solid neighbours carry solidPhi),
// perturbation and recolouring (calRKCollision23GPUNew A:1169-1267).  The stored state is the recoloured one; the next step's pull
// streams it.  Repairs R1-R4 of the dead reference driver as in openlbmpm_amd/RKD2Q9.py (R3: f_tot is summed after collision 1
// for SRT, before it for MRT).  A workgroup owns a 64 x 8 tile and recomputes the phase field on tile + 1.

// post-streaming, post-boundary-rows state of node (x, y): both colour lattices and their densities
// in two halves (source row -- pull -- boundary rows and densities), so that a thread can have the pulls of two nodes in flight
__device__ __forceinline__ int pert_source_row(const RKDev &p, int y)
{
    // the ghost rows copy their neighbour row's state; the convective outlet (convectiveOutletGPU / Ghost2GPU / Ghost3GPU, A:700-784)
    // hands row 3's streamed state down to the rows 2, 1, 0
    int ys = y == p.ny - 1 ? p.ny - 2 : y;
    if (p.outlet == LBMPM_OUTLET_PRESSURE) { if (y == 0) ys = 1; }
    else if (y <= 2) ys = 3;
    return ys;
}
__device__ __forceinline__ void pert_node_finish(const RKDev &p, const PertDev &q, int ys, double fR[9], double fB[9], double &rhoR, double &rhoB);
__device__ __forceinline__ void pert_node_state(const RKDev &p, const PertDev &q, int x, int y, double fR[9], double fB[9], double &rhoR, double &rhoB)
{
    const int ys = pert_source_row(p, y);
    pull_node(p, x, ys, fR, fB);
    pert_node_finish(p, q, ys, fR, fB, rhoR, rhoB);
}
__device__ __forceinline__ void pert_node_finish(const RKDev &p, const PertDev &q, int ys, double fR[9], double fB[9], double &rhoR, double &rhoB)
{
    if (p.outlet == LBMPM_OUTLET_PRESSURE && ys == 1) {       // A:1008-1039 (blue first, then red)
        {
            double *b = fB;
            const double pL = q.pOutB;
            const double v = 1. - 1. / pL * (b[0] + b[1] + b[3] + 2. * (b[4] + b[7] + b[8]));
            b[2] = b[4] + 2. / 3. * (pL * v);
            b[5] = b[7] + 0.5 * (b[3] - b[1]) + 1. / 6. * pL * v;
            b[6] = b[8] + 0.5 * (b[1] - b[3]) + 1. / 6. * pL * v;
        }
        {
            double *r = fR;
            const double pL = q.pOutR;
            const double v = 1. - 1. / pL * (r[0] + r[1] + r[3] + 2. * (r[4] + r[7] + r[8]));
            r[2] = r[4] + 2. / 3. * pL * v;
            r[5] = r[7] + 0.5 * (r[3] - r[1]) + 1. / 6. * pL * v;
            r[6] = r[8] + 0.5 * (r[1] - r[3]) + 1. / 6. * pL * v;
        }
    }
    if (ys == p.ny - 2 && p.inlet != LBMPM_INLET_VELOCITY) {      // calConstPressureInletGPU A:925-962, per colour
        double d;
        bc_inlet_pressure_one(p.pInB, fB, d);
        bc_inlet_pressure_one(p.pInR, fR, d);
    }
    if (ys == p.ny - 2 && p.inlet == LBMPM_INLET_VELOCITY) {      // constantVelocityZHBoundaryHigherRK A:657-695
        {
            double *r = fR;
            const double rho = (r[0] + r[1] + r[3] + 2. * (r[2] + r[5] + r[6])) / (1. + q.vyInR);
            r[4] = r[2] - 2. / 3. * rho * q.vyInR;
            r[7] = r[5] + (r[1] - r[3]) / 2. - 1. / 6. * rho * q.vyInR;
            r[8] = r[6] - (r[1] - r[3]) / 2. - 1. / 6. * rho * q.vyInR;
        }
        {
            double *b = fB;
            const double rho = (b[0] + b[1] + b[3] + 2. * (b[2] + b[5] + b[6])) / (1. + q.vyInB);
            b[4] = b[2] - 2. / 3. * rho * q.vyInB;
            b[7] = b[5] + (b[1] - b[3]) / 2. - 1. / 6. * rho * q.vyInB;
            b[8] = b[6] - (b[1] - b[3]) / 2. - 1. / 6. * rho * q.vyInB;
        }
    }
    rhoR = sum9(fR);                      // calMacroDensityRKGPU2D runs over every node AFTER the boundary kernels
    rhoB = sum9(fB);
}

// velocity of calPhysicalVelocityRKGPU2D (A:125-147): no force term, the reference's order of the twelve terms
__device__ __forceinline__ void pert_velocity(const double r[9], const double b[9], double rhoR, double rhoB, double &vx, double &vy)
{
    const double rho = rhoB + rhoR;
    const double tx = r[1] - r[3] + r[5] - r[6] - r[7] + r[8] + b[1] - b[3] + b[5] - b[6] - b[7] + b[8];
    vx = tx / rho;
    const double ty = r[2] - r[4] + r[5] + r[6] - r[7] - r[8] + b[2] - b[4] + b[5] + b[6] - b[7] - b[8];
    vy = ty / rho;
}

template <bool MRT>
__global__ __launch_bounds__(512) void rk2dp_fused(RKDev p, PertDev q, int tiles_x)
{
    constexpr int TW = 64, TH = 8, RW = TW + 2, RH = TH + 2, THREADS = TW * TH;
    constexpr double W[9] = LBMPM_D2Q9_W;
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    __shared__ double s_phi[RH * RW];
    __shared__ uint8_t s_fluid[RH * RW];
    const int t = xcd_tile(blockIdx.x, gridDim.x, tiles_x);
    const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
    const int tid = threadIdx.x, lx = tid % TW, ly = tid / TW;
    // All flags (the region's mask, the own node's, the rim node's) and -- before any of them is known -- the own node's solid-neighbour
    // byte and nine population pairs go out together; the rim node's pulls follow when the flags are in (rk2d_fused, phase A: the volume
    // of the own pulls moves while the flags are on their way; asm loads, hand-counted waits).
    using namespace lbmpm_dev;
    constexpr int NFL = (RH * RW + THREADS - 1) / THREADS;
    constexpr int NRIM = 2 * RW + 2 * TH;
    const int x = tx0 + lx, y = ty0 + ly;
    const bool inside = x < p.nx && y < p.ny;
    const int xw = inside ? x : wrapm(x, p.nx), yw = inside ? y : wrapm(y, p.ny);       // (partial tiles: periodic images serve as rim)
    const size_t idx = (size_t)yw * p.pitch + xw;
    int hr = -1, hx = 0, hy = 0;
    if (tid < NRIM) {
        int rx, ry;
        if (tid < RW) { ry = 0; rx = tid; }
        else if (tid < 2 * RW) { ry = RH - 1; rx = tid - RW; }
        else { const int k = tid - 2 * RW; ry = 1 + k / 2; rx = (k & 1) ? RW - 1 : 0; }
        hx = wrapm(tx0 - 1 + rx, p.nx); hy = wrapm(ty0 - 1 + ry, p.ny);
        hr = ry * RW + rx;
    }
    unsigned flr[NFL];
#pragma unroll
    for (int k = 0; k < NFL; ++k) {
        const int n = min(tid + k * THREADS, RH * RW - 1);
        const int xx = wrapm(tx0 - 1 + n % RW, p.nx), yy = wrapm(ty0 - 1 + n / RW, p.ny);
        flr[k] = asm_ldu8(p.flags + (size_t)yy * p.pitch + xx);
    }
    unsigned ofl = asm_ldu8(p.flags + idx);
    unsigned hfl = asm_ldu8(p.flags + (size_t)hy * p.pitch + hx);          // (lanes without a rim node: node (0, 0), ignored)
    const int ys = pert_source_row(p, yw);
    unsigned osn = asm_ldu8(p.solidnbr + (size_t)ys * p.pitch + xw);
    lbmpm_d2 pq[9], hq[9];
    pull_issue_asm(p, xw, ys, pq);
    __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(10));              // the flags are in; the own node's 10 loads may still be out
    asm volatile("" : "+v"(ofl), "+v"(hfl));
#pragma unroll
    for (int k = 0; k < NFL; ++k) {
        asm volatile("" : "+v"(flr[k]));
        const int n = tid + k * THREADS;
        if (n < RH * RW) s_fluid[n] = flr[k] & 1;
    }
    const bool fluid = ofl & 1, act = inside && fluid;
    const bool hdo = hr >= 0 && (hfl & 1);
    const bool first = p.first != 0;
    int hys = 0;
    unsigned hsn = 0;
    if (__ballot(hdo) != 0ull) {                               // wave-uniform: the count of loads in flight must be known
        if (hdo) {
            LBMPM_TAKEN;
            hys = pert_source_row(p, hy);
            hsn = asm_ldu8(p.solidnbr + (size_t)hys * p.pitch + hx);
            pull_issue_asm(p, hx, hys, hq);
        }
        __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(10));          // the own node's loads have landed, the rim node's 10 are out
    } else __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(0));
    asm volatile("" : "+v"(osn));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(pq[i]));
    double fR[9], fB[9], rR = 1., rB = 1.;
    if (fluid) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { fR[i] = pq[i].x; fB[i] = pq[i].y; }
        pull_patch(p, xw, ys, first ? 0u : osn, fR, fB);
        pert_node_finish(p, q, ys, fR, fB, rR, rB);
    }
    __builtin_amdgcn_s_waitcnt(LBMPM_VMCNT(0));
    asm volatile("" : "+v"(hsn));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(hq[i]));
    double hphi = 0.;
    if (hdo) {
        double a[9], b[9], ra, rb;
#pragma unroll
        for (int i = 0; i < 9; ++i) { a[i] = hq[i].x; b[i] = hq[i].y; }
        pull_patch(p, hx, hys, first ? 0u : hsn, a, b);
        pert_node_finish(p, q, hys, a, b, ra, rb);
        hphi = (ra - rb) / (ra + rb);
    } else hr = -1;
    const int ri = (1 + ly) * RW + 1 + lx;
    if (fluid) s_phi[ri] = (rR - rB) / (rR + rB);
    if (hr >= 0) s_phi[hr] = hphi;
    __syncthreads();

    const bool line8 = lbmpm_dev::line_has_active<8>(act, tid & 63);
    if (!(line8 && inside)) return;
    double oR[9], oB[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { oR[i] = 0.; oB[i] = 0.; }
    if (act) {
        double vx, vy;
        pert_velocity(fR, fB, rR, rB, vx, vy);
        const double phi = (rR - rB) / (rR + rB);
        const double tau = 0.5 + 1. / ((1. + phi) / (2. * (p.tauR - 0.5)) + (1. - phi) / (2. * (p.tauB - 0.5)));   // A:1144, A:1307
        double fT[9];
        if (!MRT) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const double eR = feq(rR, W[i], (double)EX[i], (double)EY[i], vx, vy);
                const double cR = -1. / tau * (fR[i] - eR);
                const double eB = feq(rB, W[i], (double)EX[i], (double)EY[i], vx, vy);
                const double cB = -1. / tau * (fB[i] - eB);
                fR[i] = fR[i] + cR;
                fB[i] = fB[i] + cB;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) fT[i] = fR[i] + fB[i];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) fT[i] = fR[i] + fB[i];
            collide_tau<true>(tau, fT, rR + rB, vx, vy, 0., 0.);       // the moment-space form of A:1272-1343 (S = 0, 1.64, 1.54, 0, 1.9, 0, 1.9, 1/tau, 1/tau)
        }
        // calRKCollision23GPUNew, A:1169-1267
        double gx = 0., gy = 0.;
#pragma unroll
        for (int i = 1; i < 9; ++i) {
            const int rn = ri + EY[i] * RW + EX[i];
            const double ph = s_fluid[rn] ? s_phi[rn] : q.solidPhi;
            gx += 3. * W[i] * (double)EX[i] * ph;
            gy += 3. * W[i] * (double)EY[i] * ph;
        }
        constexpr double BC[9] = {-2. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 36., 1. / 36., 1. / 36., 1. / 36.};     // constantBNew, RKD2Q9.py:131-133
        const double g2 = gx * gx + gy * gy, gn = sqrt(g2);
        const double rs = rR + rB, rm = rR * rB, rs2 = rs * rs;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double c2 = 0.;
            if (g2 != 0.) {
                const double eg = (double)EX[i] * gx + (double)EY[i] * gy;
                const double part = W[i] * (eg * eg) / g2;
                c2 = (q.akR + q.akB) * 0.5 * gn * (part - BC[i]);
            }
            fT[i] += c2;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double en = sqrt((double)(EX[i] * EX[i] + EY[i] * EY[i]));
            double c = 0.;
            if (!(en == 0. || gn == 0.)) c = ((double)EX[i] * gx + (double)EY[i] * gy) / (en * gn);
            oR[i] = rR / rs * fT[i] + (p.beta * rm / rs2) * W[i] * c;
            oB[i] = rB / rs * fT[i] - (p.beta * rm / rs2) * W[i] * c;
        }
        if (q.pd) {
            q.pd[idx] = rR; q.pd[p.plane + idx] = rB; q.pd[2 * p.plane + idx] = vx; q.pd[3 * p.plane + idx] = vy;
            p.phi[idx] = phi; p.G[idx] = gx; p.G[p.plane + idx] = gy;
        }
    }
    lbmpm_dev::store_pairs<true>(p.fout, p.plane, idx, oR, oB);
}

// what the perturbation loop would record at the start of the next step (RKD2Q9.py:1121-1131: after streaming, boundary kernels,
// densities and velocity): out[22][plane] like rk2d_observe
__global__ __launch_bounds__(BX *BY) void rk2dp_observe(RKDev p, PertDev q, double *out)
{
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[9], fB[9], rR, rB, vx, vy;
    pert_node_state(p, q, x, y, fR, fB, rR, rB);
    pert_velocity(fR, fB, rR, rB, vx, vy);
#pragma unroll
    for (int i = 0; i < 9; ++i) { out[i * p.plane + idx] = fR[i]; out[(9 + i) * p.plane + idx] = fB[i]; }
    out[18 * p.plane + idx] = rR; out[19 * p.plane + idx] = rB;
    out[20 * p.plane + idx] = vx; out[21 * p.plane + idx] = vy;
}
// the stored (recoloured) populations as the device arrays hold them after a completed step: out[18][plane]
__global__ __launch_bounds__(BX *BY) void rk2dp_stored(RKDev p, double *out)
{
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const double2 *f2 = reinterpret_cast<const double2 *>(p.fin);
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double2 v = f2[i * p.plane + idx]; out[i * p.plane + idx] = v.x; out[(9 + i) * p.plane + idx] = v.y; }
}

// Observation kernels: populations/densities as the reference's device arrays hold them
// after the last completed step (WITH_BC=false), or as resultInHDF5 records them at the
// start of the next step (WITH_BC=true: + velocity, RKD2Q9.py:1382-1393).
template <bool WITH_BC>
__global__ __launch_bounds__(BX *BY) void rk2d_observe(RKDev p, double *out /*[22][plane]*/)
{
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[9], fB[9], rR, rB;
    node_state<WITH_BC>(p, x, y, fR, fB, rR, rB);
#pragma unroll
    for (int i = 0; i < 9; ++i) { out[i * p.plane + idx] = fR[i]; out[(9 + i) * p.plane + idx] = fB[i]; }
    out[18 * p.plane + idx] = rR;
    out[19 * p.plane + idx] = rB;
    double fT[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fT[i] = fR[i] + fB[i];
    const double rs = rB + rR;
    out[20 * p.plane + idx] = (fT[1] - fT[3] + fT[5] - fT[6] - fT[7] + fT[8] + 0.5 * p.F[idx]) / rs;
    out[21 * p.plane + idx] = (fT[2] - fT[4] + fT[5] + fT[6] - fT[7] - fT[8] + 0.5 * p.F[p.plane + idx]) / rs;
}

// Tracer concentration as the reference's deviceTracerConc holds it after the last completed step
// (streamed + inlet-corrected populations, calConcentrationGPU T:78-90)
__global__ __launch_bounds__(BX *BY) void rk2d_observe_tracer(RKDev p, int t, double *out)
{
    constexpr double W5[5] = {1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.};
    constexpr int VX5[5] = {0, 1, -1, 0, 0}, VY5[5] = {0, 0, 0, 1, -1}, OPP5[5] = {0, 2, 1, 4, 3};
    constexpr int SRCBIT[5] = {0, 2, 0, 3, 1};
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    const bool first = p.first != 0;
    const unsigned sn = p.solidnbr[idx];
    const int yo = (p.trFree && y == 0 && !first) ? 1 : y;
    const size_t own = (size_t)yo * p.pitch + x;
    const double *gi = p.gin + (size_t)t * 5 * p.plane;
    double g[5];
    g[0] = gi[own];
    for (int j = 1; j < 5; ++j) {
        if (first) { g[j] = gi[j * p.plane + idx]; continue; }
        if ((sn >> SRCBIT[j]) & 1u) { g[j] = gi[OPP5[j] * p.plane + own]; continue; }
        int ys = wrapi(y - VY5[j], p.ny);
        if (p.trFree && ys == 0) ys = 1;
        g[j] = gi[j * p.plane + (size_t)ys * p.pitch + wrapi(x - VX5[j], p.nx)];
    }
    if (!first && p.trDirichlet && y == p.ny - 1) {
        const double sm = g[0] + g[1] + g[2] + g[3];
        g[4] = W5[4] * ((p.trCb[t] - sm) / W5[4]);
    }
    double C = 0.;
    for (int j = 0; j < 5; ++j) C += g[j];
    out[idx] = C;
}

// ---------------------------------------------------------------- set-up kernels
// calVectorNormaltoSolid, RKD2Q9.py:768-892: 24-point iso-8 stencil over the solid mask.
__global__ void rk2d_setup_normals(int nx, int ny, int pitch, size_t plane, const uint8_t *flags,
                                   const uint8_t *solidnbr, double *ns)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= nx || y >= ny) return;
    const size_t idx = (size_t)y * pitch + x;
    if (!(flags[idx] & 1) || solidnbr[idx] == 0) return;
    // reference accumulation order (dx, dy, weight)
    constexpr int DX[24] = {1, 0, -1, 0, 1, -1, -1, 1, 2, 0, -2, 0, 2, 1, -1, -2, -2, -1, 1, 2, 2, -2, -2, 2};
    constexpr int DY[24] = {0, 1, 0, -1, 1, 1, -1, -1, 0, 2, 0, -2, 1, 2, 2, 1, -1, -2, -2, -1, 2, 2, -2, -2};
    constexpr double WW[24] = {4. / 21., 4. / 21., 4. / 21., 4. / 21., 4. / 45., 4. / 45., 4. / 45., 4. / 45.,
                               1. / 60., 1. / 60., 1. / 60., 1. / 60., 2. / 315., 2. / 315., 2. / 315., 2. / 315.,
                               2. / 315., 2. / 315., 2. / 315., 2. / 315., 1. / 5040., 1. / 5040., 1. / 5040., 1. / 5040.};
    double sx = 0., sy = 0.;
    for (int k = 0; k < 24; ++k) {
        int xn = x + DX[k], yn = y + DY[k];
        xn = xn < 0 ? xn + nx : (xn >= nx ? xn - nx : xn);
        yn = yn < 0 ? yn + ny : (yn >= ny ? yn - ny : yn);
        if (!(flags[(size_t)yn * pitch + xn] & 1)) {
            sx += WW[k] * 1. * (double)DX[k];
            sy += WW[k] * 1. * (double)DY[k];
        }
    }
    const double n = sqrt(sx * sx + sy * sy);
    ns[idx] = sx / n;
    ns[plane + idx] = sy / n;
}

}  // namespace

// ====================================================================== host side
struct lbmpm_rk2d {
    lbmpm_rk2d_config cfg;
    int nx, ny, pitch;
    size_t plane;
    int64_t nfluid = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t *flags = nullptr, *solidnbr = nullptr;
    double *fA = nullptr, *fB = nullptr;   // ping-pong [2][9][plane]; fA holds the current state
    double *F = nullptr, *ns = nullptr, *phi = nullptr, *G = nullptr, *diag = nullptr, *obs = nullptr;
    std::vector<uint8_t> h_domain;
    int ntr = 0, trFree = 0, trDirichlet = 0;
    double *gA = nullptr, *gB = nullptr;
    double trCrit = 0.5, trM[25] = {0}, trA[4][25] = {{0}}, trBeta[4] = {0}, trCb[4] = {0};
    double trRate = 0., trJ[4] = {1. / 3., 1. / 3., 1. / 3., 1. / 3.};
    int model = 0;            // 0 CSF (the create-time model), 1 perturbation operator (lbmpm_rk2d_set_perturbation)
    lbmpm_rk2d_perturbation pert{};
    double *pd = nullptr;     // perturbation model: rhoR, rhoB, vx, vy of the last step (diagnostics)
    int shape = 0;            // fused tile shape (LBMPM_RK2D_SHAPE, tuning only)
    bool streamed = false;    // false: fA holds the initial (already "post-streaming") state
    bool diag_valid = false;
    int64_t steps = 0;
    int64_t bytes = 0;
    lbmpm::EventPool pool;
};

namespace {

PertDev make_pert(const lbmpm_rk2d *c, bool diag)
{
    PertDev q;
    q.akR = c->pert.ak_r; q.akB = c->pert.ak_b; q.solidPhi = c->pert.solid_phi;
    q.vyInR = c->pert.inlet_velocity_y_r; q.vyInB = c->pert.inlet_velocity_y_b;
    q.pOutR = c->pert.outlet_rho_r; q.pOutB = c->pert.outlet_rho_b;
    q.pd = diag ? c->pd : nullptr;
    return q;
}

RKDev make_dev(const lbmpm_rk2d *c)
{
    RKDev p;
    p.nx = c->nx; p.ny = c->ny; p.pitch = c->pitch; p.plane = c->plane;
    p.flags = c->flags; p.solidnbr = c->solidnbr;
    p.fin = c->fA; p.fout = c->fB; p.F = c->F; p.ns = c->ns; p.phi = c->phi; p.G = c->G; p.diag = c->diag;
    const double th = c->cfg.contact_angle_deg / 180. * M_PI;
    p.sigma = c->cfg.surface_tension; p.cosT = cos(th); p.sinT = sin(th);
    p.beta = c->cfg.beta; p.delta = c->cfg.delta; p.tauR = c->cfg.tau_r; p.tauB = c->cfg.tau_b;
    p.vyIn = c->cfg.inlet_velocity_y; p.pInB = c->cfg.inlet_rho_b; p.pInR = c->cfg.inlet_rho_r;
    p.pOut = c->cfg.outlet_rho_total;
    p.wetting = c->cfg.wetting_type; p.tautype = c->cfg.tau_type;
    p.inlet = c->cfg.inlet_type; p.outlet = c->cfg.outlet_type;
    p.first = c->streamed ? 0 : 1;
    if (c->model == 1) p.first = 0;         // the perturbation loop streams first: the initial state is pulled like any other
    p.ntr = c->ntr; p.trFree = c->trFree; p.trDirichlet = c->trDirichlet; p.gin = c->gA; p.gout = c->gB;
    p.trCrit = c->trCrit;
    memcpy(p.trM, c->trM, sizeof(p.trM)); memcpy(p.trA, c->trA, sizeof(p.trA));
    memcpy(p.trBeta, c->trBeta, sizeof(p.trBeta)); memcpy(p.trCb, c->trCb, sizeof(p.trCb));
    p.trRate = c->trRate; memcpy(p.trJ, c->trJ, sizeof(p.trJ));
    return p;
}

dim3 grid_of(const lbmpm_rk2d *c) { return dim3((c->nx + BX - 1) / BX, (c->ny + BY - 1) / BY); }

template <typename SH>
void launch_fused_shape(lbmpm_rk2d *c, const RKDev &p)
{
    const int tiles_x = (c->nx + SH::TW - 1) / SH::TW, tiles_y = (c->ny + SH::TH - 1) / SH::TH;
    const dim3 g(tiles_x * tiles_y), b(SH::THREADS);
    if (c->cfg.relaxation == LBMPM_RELAX_MRT) rk2d_fused<true, false, SH><<<g, b, 0, c->stream>>>(p, tiles_x);
    else rk2d_fused<false, false, SH><<<g, b, 0, c->stream>>>(p, tiles_x);
}

template <typename SH>
void launch_fused_tracer(lbmpm_rk2d *c, const RKDev &p)
{
    const int tiles_x = (c->nx + SH::TW - 1) / SH::TW, tiles_y = (c->ny + SH::TH - 1) / SH::TH;
    const dim3 g(tiles_x * tiles_y), b(SH::THREADS);
    const bool mrt = c->cfg.relaxation == LBMPM_RELAX_MRT;
    // The re-summing variant (TR = 1: the transport driver's order of boundary rows and density sums, Transport2DRK.py:1199-1287) goes to
    // every tile row whose region -- its TH own rows + 3 rows of halo on either side, periodic -- holds one of the rows 0, 1, ny-2, ny-1:
    // the n_lo tile rows from the bottom with ty TH - 3 <= 1 and the n_hi from the top with ty TH + TH + 2 >= ny - 2 (one for TH = 8 when
    // ny is a multiple of 8, two otherwise: ny % 8 in 1..4 puts row ny-2 into the halo or among the own rows of tile row tiles_y - 2).
    // The tile rows in between run the variant that sums once.
    constexpr int H = 3;
    int n_lo = 0, n_hi = 0;
    for (int ty = 0; ty < tiles_y; ++ty) {
        if (ty * SH::TH - H <= 1) n_lo = ty + 1;
        if (ty * SH::TH + SH::TH - 1 + H >= c->ny - 2 && n_hi == 0) n_hi = tiles_y - ty;
    }
    if (n_lo + n_hi >= tiles_y) {
        if (mrt) rk2d_fused<true, true, SH, 1><<<g, b, 0, c->stream>>>(p, tiles_x, 0);
        else rk2d_fused<false, true, SH, 1><<<g, b, 0, c->stream>>>(p, tiles_x, 0);
        return;
    }
    if (mrt) rk2d_fused_tracer<true, SH><<<g, b, 0, c->stream>>>(p, tiles_x, tiles_x * n_lo, tiles_x * (tiles_y - n_hi));
    else rk2d_fused_tracer<false, SH><<<g, b, 0, c->stream>>>(p, tiles_x, tiles_x * n_lo, tiles_x * (tiles_y - n_hi));
}

int launch_step(lbmpm_rk2d *c, bool diag, bool timed)
{
    RKDev p = make_dev(c);
    if (!diag) p.diag = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->model == 1) {
        const PertDev q = make_pert(c, diag);
        const bool ev = timed && c->pool.take(&e0, &e1);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
        const int tiles_x = (c->nx + 63) / 64, tiles_y = (c->ny + 7) / 8;
        if (c->cfg.relaxation == LBMPM_RELAX_MRT) rk2dp_fused<true><<<dim3(tiles_x * tiles_y), dim3(512), 0, c->stream>>>(p, q, tiles_x);
        else rk2dp_fused<false><<<dim3(tiles_x * tiles_y), dim3(512), 0, c->stream>>>(p, q, tiles_x);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    } else if (c->cfg.variant == 0) {
        const bool ev = timed && c->pool.take(&e0, &e1);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
        if (c->ntr > 0) {
            // registers are capped at 128 (amdgpu_waves_per_eu on the kernel): two 64 x 8 blocks share a CU
            if (c->shape == 3) launch_fused_tracer<FusedShape<4, 1>>(c, p);
#if defined(LBMPM_DEV) && defined(LBMPM_TRACER_TALL)
            else if (c->shape == 2) launch_fused_tracer<FusedShape<16, 1>>(c, p);      // tools/dev/walk2d.sh: one 1024-thread workgroup per CU
#endif
            else launch_fused_tracer<FusedShape<8, 1>>(c, p);
        } else
        switch (c->shape) {
            // tile-shape sweep on MI355X (1024^2): 64x8 / 1 node per thread is the
            // default; the others stay selectable for tuning (LBMPM_RK2D_SHAPE)
            case 1: launch_fused_shape<FusedShape<16, 2>>(c, p); break;
            case 3: launch_fused_shape<FusedShape<4, 1>>(c, p); break;
            case 2: launch_fused_shape<FusedShape<16, 1>>(c, p); break;
            default: launch_fused_shape<FusedShape<8, 1>>(c, p); break;
        }
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    } else {
        const dim3 g = grid_of(c), b(BX, BY);
        rk2d_phase_field<<<g, b, 0, c->stream>>>(p);
        rk2d_gradient<<<g, b, 0, c->stream>>>(p);
        const bool ev = timed && c->pool.take(&e0, &e1);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e0, c->stream));
        if (c->cfg.relaxation == LBMPM_RELAX_MRT) rk2d_collide_stream<true><<<g, b, 0, c->stream>>>(p);
        else rk2d_collide_stream<false><<<g, b, 0, c->stream>>>(p);
        if (ev) LBMPM_HIP_TRY(hipEventRecord(e1, c->stream));
    }
    LBMPM_HIP_TRY(hipGetLastError());
    std::swap(c->fA, c->fB);
    if (c->ntr > 0) std::swap(c->gA, c->gB);
    c->streamed = true;
    c->diag_valid = diag;
    c->steps += 1;
    return LBMPM_OK;
}

int run_steps(lbmpm_rk2d *c, int64_t n, bool timed)
{
    for (int64_t k = 0; k < n; ++k) {
        const bool diag = (c->diag != nullptr) && (k == n - 1);
        // event pairs around every 8th launch only: a record on each side of every 0.1 ms kernel would
        // open a gap behind each of them and slow down the very loop that is being measured
        const int rc = launch_step(c, diag, timed && (k & 7) == 0);
        if (rc != LBMPM_OK) return rc;
    }
    return LBMPM_OK;
}

template <typename T>
int dev_alloc(lbmpm_rk2d *c, T **ptr, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return LBMPM_ERR_NOMEM;
    }
    // zero on the context's own stream: a null-stream memset is not ordered against the
    // non-blocking solver stream and could land after a kernel that already wrote the buffer
    e = hipMemsetAsync(v, 0, count * sizeof(T), c->stream);
    if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return LBMPM_ERR_HIP; }
    *ptr = static_cast<T *>(v);
    c->bytes += (int64_t)(count * sizeof(T));
    return LBMPM_OK;
}

}  // namespace

extern "C" int lbmpm_rk2d_create(const lbmpm_rk2d_config *cfg, const uint8_t *is_domain, lbmpm_rk2d **out)
{
    LBMPM_REQUIRE(cfg && is_domain && out, "lbmpm_rk2d_create: null argument");
    LBMPM_REQUIRE(cfg->nx >= 4 && cfg->ny >= 8 && cfg->nx < (1 << 30) && cfg->ny < (1 << 30),
                  "lbmpm_rk2d_create: domain %lld x %lld out of range", (long long)cfg->nx, (long long)cfg->ny);
    // (the fused kernels address a node inside a plane of 16-byte pairs with 32 bits: d2q9_device.h::pull_issue_asm)
    LBMPM_REQUIRE((size_t)((cfg->nx + 31) / 32 * 32) * (size_t)cfg->ny < ((size_t)1 << 28), "lbmpm_rk2d_create: more than 2^28 nodes per lattice plane");
    LBMPM_REQUIRE(cfg->wetting_type == 1 || cfg->wetting_type == 2, "WettingType must be 1 or 2 (got %d)", cfg->wetting_type);
    LBMPM_REQUIRE(cfg->tau_type == 1 || cfg->tau_type == 2, "TauType must be 1 or 2 (got %d)", cfg->tau_type);
    LBMPM_REQUIRE(cfg->relaxation == LBMPM_RELAX_SRT || cfg->relaxation == LBMPM_RELAX_MRT, "bad relaxation %d", cfg->relaxation);
    LBMPM_REQUIRE(cfg->inlet_type == 0 || cfg->inlet_type == 1, "bad inlet_type %d", cfg->inlet_type);
    LBMPM_REQUIRE(cfg->outlet_type == 0 || cfg->outlet_type == 1, "bad outlet_type %d", cfg->outlet_type);
    LBMPM_REQUIRE(cfg->tau_r > 0.5 && cfg->tau_b > 0.5, "TauR/TauB must exceed 0.5");
    LBMPM_REQUIRE(cfg->variant == 0 || cfg->variant == 1, "variant must be 0 (fused) or 1 (split), got %d", cfg->variant);
    LBMPM_HIP_TRY(hipSetDevice(cfg->device));
    lbmpm_rk2d *c = new (std::nothrow) lbmpm_rk2d();
    if (!c) { set_error("out of host memory"); return LBMPM_ERR_NOMEM; }
    c->cfg = *cfg;
#ifdef LBMPM_DEV      // tile shapes other than the default 64 x 8: development builds only (openlbmpm_amd/build.py::build_dev)
    if (const char *e = getenv("LBMPM_RK2D_SHAPE")) c->shape = atoi(e);
#endif
    c->nx = (int)cfg->nx; c->ny = (int)cfg->ny;
    c->pitch = (c->nx + 31) / 32 * 32;           // rows start on 256-byte boundaries
    c->plane = (size_t)c->pitch * c->ny;
    c->h_domain.assign(is_domain, is_domain + (size_t)c->nx * c->ny);
    std::vector<uint8_t> hflags(c->plane, 0);
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const uint8_t v = is_domain[(size_t)y * c->nx + x] == 1 ? 1 : 0;
            hflags[(size_t)y * c->pitch + x] = v;
            c->nfluid += v;
        }
    if (c->nfluid == 0) { set_error("lbmpm_rk2d_create: the domain has no fluid node (is_domain == 1 marks fluid)"); delete c; return LBMPM_ERR_INVALID; }
    int rc = LBMPM_OK;
#define TRY_RC(e) do { rc = (e); if (rc != LBMPM_OK) { lbmpm_rk2d_destroy(c); return rc; } } while (0)
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return LBMPM_ERR_HIP; }
        c->own_stream = true;
    }
    TRY_RC(dev_alloc(c, &c->flags, c->plane));
    TRY_RC(dev_alloc(c, &c->solidnbr, c->plane));
    TRY_RC(dev_alloc(c, &c->fA, 18 * c->plane));
    TRY_RC(dev_alloc(c, &c->fB, 18 * c->plane));
    TRY_RC(dev_alloc(c, &c->F, 2 * c->plane));
    TRY_RC(dev_alloc(c, &c->ns, 2 * c->plane));
    TRY_RC(dev_alloc(c, &c->phi, c->plane));
    TRY_RC(dev_alloc(c, &c->G, 2 * c->plane));
    {
        // every transfer goes through the context's stream: the null stream is not ordered
        // against it (hipStreamNonBlocking)
        hipError_t e = hipMemcpyAsync(c->flags, hflags.data(), c->plane, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error("hipMemcpy(flags) failed: %s", hipGetErrorString(e)); lbmpm_rk2d_destroy(c); return LBMPM_ERR_HIP; }
    }
    const dim3 b(64, 4), g((c->nx + 63) / 64, (c->ny + 3) / 4);
    setup_solidnbr<<<g, b, 0, c->stream>>>(c->nx, c->ny, c->pitch, c->flags, c->solidnbr);
    rk2d_setup_normals<<<g, b, 0, c->stream>>>(c->nx, c->ny, c->pitch, c->plane, c->flags, c->solidnbr, c->ns);
    {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { set_error("set-up kernels failed: %s", hipGetErrorString(e)); lbmpm_rk2d_destroy(c); return LBMPM_ERR_HIP; }
    }
#undef TRY_RC
    *out = c;
    return LBMPM_OK;
}

extern "C" void lbmpm_rk2d_destroy(lbmpm_rk2d *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *ptr : {(void *)c->flags, (void *)c->solidnbr, (void *)c->fA, (void *)c->fB, (void *)c->F,
                      (void *)c->ns, (void *)c->phi, (void *)c->G, (void *)c->diag, (void *)c->obs, (void *)c->gA, (void *)c->gB, (void *)c->pd})
        if (ptr) (void)hipFree(ptr);
    c->pool.destroy();
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int lbmpm_rk2d_set_stream(lbmpm_rk2d *c, void *hip_stream)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (hip_stream == nullptr) {
        if (!c->own_stream) {
            LBMPM_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        return LBMPM_OK;
    }
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = static_cast<hipStream_t>(hip_stream);
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_set_pdf(lbmpm_rk2d *c, const double *pdf_r, const double *pdf_b)
{
    LBMPM_REQUIRE(c && pdf_r && pdf_b, "lbmpm_rk2d_set_pdf: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    std::vector<double> h(18 * c->plane, 0.0);
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const size_t s = ((size_t)y * c->nx + x) * 9, d = (size_t)y * c->pitch + x;
            if (c->h_domain[(size_t)y * c->nx + x] != 1) continue;
            for (int i = 0; i < 9; ++i) {
                h[lbmpm_dev::fslot(c->plane, i, d, 0)] = pdf_r[s + i];
                h[lbmpm_dev::fslot(c->plane, i, d, 1)] = pdf_b[s + i];
            }
        }
    LBMPM_HIP_TRY(hipMemcpyAsync(c->fA, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->F, 0, 2 * c->plane * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    c->streamed = false;
    c->diag_valid = false;
    c->steps = 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_set_macro(lbmpm_rk2d *c, const double *rho_r, const double *rho_b, const double *vx,
                                    const double *vy)
{
    LBMPM_REQUIRE(c && rho_r && rho_b, "lbmpm_rk2d_set_macro: null argument");
    static const double EXd[9] = LBMPM_D2Q9_EX, EYd[9] = LBMPM_D2Q9_EY, Wd[9] = LBMPM_D2Q9_W;
    const size_t n = (size_t)c->nx * c->ny;
    std::vector<double> fr(9 * n, 0.0), fb(9 * n, 0.0);
    for (size_t k = 0; k < n; ++k) {
        if (c->h_domain[k] != 1) continue;
        const double ux = vx ? vx[k] : 0.0, uy = vy ? vy[k] : 0.0;
        for (int i = 0; i < 9; ++i) {   // RKD2Q9.py:577-601
            const double eu = EXd[i] * ux + EYd[i] * uy;
            const double t = 1 + (3. * eu + 4.5 * eu * eu - 1.5 * (ux * ux + uy * uy));
            fr[9 * k + i] = rho_r[k] * Wd[i] * t;
            fb[9 * k + i] = rho_b[k] * Wd[i] * t;
        }
    }
    return lbmpm_rk2d_set_pdf(c, fr.data(), fb.data());
}

extern "C" int lbmpm_rk2d_step(lbmpm_rk2d *c, int64_t nsteps)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk2d_step: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    return run_steps(c, nsteps, false);
}

extern "C" int lbmpm_rk2d_step_timed(lbmpm_rk2d *c, int64_t nsteps, double *ms_total, double *ms_dominant)
{
    LBMPM_REQUIRE(c && nsteps >= 0, "lbmpm_rk2d_step_timed: bad argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t pairs = (size_t)(nsteps < 4096 ? nsteps : 4096);
    if (c->pool.reserve(pairs + 1) != LBMPM_OK) { set_error("hipEventCreate failed"); return LBMPM_ERR_HIP; }
    c->pool.reset();
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
        const size_t timed_launches = c->pool.used / 2 - 1;
        double s = 0.0;
        for (size_t k = 2; k + 1 < c->pool.used; k += 2) {
            float m = 0.f;
            LBMPM_HIP_TRY(hipEventElapsedTime(&m, c->pool.ev[k], c->pool.ev[k + 1]));
            s += m;
        }
        // scale to all launches when more steps than pooled event pairs were run
        *ms_dominant = timed_launches ? s * (double)nsteps / (double)timed_launches : 0.0;
    }
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_sync(lbmpm_rk2d *c)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_enable_diagnostics(lbmpm_rk2d *c, int on)
{
    LBMPM_REQUIRE(c, "null context");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (on && !c->diag) { const int rc = dev_alloc(c, &c->diag, 3 * c->plane); if (rc) return rc; }
    if (on && c->model == 1 && !c->pd) { const int rc = dev_alloc(c, &c->pd, 4 * c->plane); if (rc) return rc; }
    if (!on && c->diag) {
        LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->diag); c->diag = nullptr; c->diag_valid = false;
        c->bytes -= (int64_t)(3 * c->plane * sizeof(double));
    }
    return LBMPM_OK;
}

namespace {

int copy_plane(lbmpm_rk2d *c, const double *dev, double *out, int ncomp)
{   // device SoA [ncomp][plane] -> host dense AoS [ny][nx][ncomp], zeros at solid
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

extern "C" int lbmpm_rk2d_get_field(lbmpm_rk2d *c, int field, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk2d_get_field: null argument");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    const bool rec = field >= LBMPM_RK_REC_PDF_R && field <= LBMPM_RK_REC_VY;
    const bool cur_obs = field >= LBMPM_RK_PDF_R && field <= LBMPM_RK_RHO_B;
    if (c->model == 1) {
        // perturbation loop (streams first): after a completed step the reference's arrays hold the recoloured populations and the
        // densities / velocity / phase field computed DURING that step (kept by the kernel when diagnostics are on); the "rec" fields
        // are what the next step would record after its streaming and boundary kernels
        if (rec || field == LBMPM_RK_PDF_R || field == LBMPM_RK_PDF_B) {
            if (!c->obs) { const int rc = dev_alloc(c, &c->obs, 22 * c->plane); if (rc) return rc; }
            RKDev p = make_dev(c);
            const dim3 g = grid_of(c), b(BX, BY);
            if (rec) rk2dp_observe<<<g, b, 0, c->stream>>>(p, make_pert(c, false), c->obs);
            else rk2dp_stored<<<g, b, 0, c->stream>>>(p, c->obs);
            LBMPM_HIP_TRY(hipGetLastError());
            LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
            const int f = rec ? field - LBMPM_RK_REC_PDF_R : field;
            switch (f) {
                case 0: return copy_plane(c, c->obs, out, 9);
                case 1: return copy_plane(c, c->obs + 9 * c->plane, out, 9);
                case 2: return copy_plane(c, c->obs + 18 * c->plane, out, 1);
                case 3: return copy_plane(c, c->obs + 19 * c->plane, out, 1);
                case 4: return copy_plane(c, c->obs + 20 * c->plane, out, 1);
                default: return copy_plane(c, c->obs + 21 * c->plane, out, 1);
            }
        }
        if (!(c->pd && c->diag_valid)) {
            set_error("field %d of the perturbation model needs lbmpm_rk2d_enable_diagnostics(ctx, 1) before the last lbmpm_rk2d_step", field);
            return LBMPM_ERR_STATE;
        }
        switch (field) {
            case LBMPM_RK_RHO_R: return copy_plane(c, c->pd, out, 1);
            case LBMPM_RK_RHO_B: return copy_plane(c, c->pd + c->plane, out, 1);
            case LBMPM_RK_VX: return copy_plane(c, c->pd + 2 * c->plane, out, 1);
            case LBMPM_RK_VY: return copy_plane(c, c->pd + 3 * c->plane, out, 1);
            case LBMPM_RK_PHI: return copy_plane(c, c->phi, out, 1);
            case LBMPM_RK_GX: return copy_plane(c, c->G, out, 1);
            case LBMPM_RK_GY: return copy_plane(c, c->G + c->plane, out, 1);
            default: break;
        }
        set_error("lbmpm_rk2d_get_field: the perturbation model has no field %d (no CSF force, no curvature)", field);
        return LBMPM_ERR_INVALID;
    }
    if (rec || cur_obs) {
        if (!c->obs) { const int rc = dev_alloc(c, &c->obs, 22 * c->plane); if (rc) return rc; }
        RKDev p = make_dev(c);
        const dim3 g = grid_of(c), b(BX, BY);
        if (rec) rk2d_observe<true><<<g, b, 0, c->stream>>>(p, c->obs);
        else rk2d_observe<false><<<g, b, 0, c->stream>>>(p, c->obs);
        LBMPM_HIP_TRY(hipGetLastError());
        LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
        const int f = rec ? field - LBMPM_RK_REC_PDF_R : field;
        switch (f) {
            case 0: return copy_plane(c, c->obs, out, 9);
            case 1: return copy_plane(c, c->obs + 9 * c->plane, out, 9);
            case 2: return copy_plane(c, c->obs + 18 * c->plane, out, 1);
            case 3: return copy_plane(c, c->obs + 19 * c->plane, out, 1);
            case 4: return copy_plane(c, c->obs + 20 * c->plane, out, 1);
            case 5: return copy_plane(c, c->obs + 21 * c->plane, out, 1);
        }
    }
    if (c->cfg.variant == 0 && field >= LBMPM_RK_PHI && field <= LBMPM_RK_GY && !(c->diag && c->diag_valid)) {
        set_error("field %d needs lbmpm_rk2d_enable_diagnostics(ctx, 1) before the last lbmpm_rk2d_step "
                  "(the fused schedule keeps phi and G on chip)", field);
        return LBMPM_ERR_STATE;
    }
    switch (field) {
        case LBMPM_RK_PHI: return copy_plane(c, c->phi, out, 1);
        case LBMPM_RK_GX: return copy_plane(c, c->G, out, 1);
        case LBMPM_RK_GY: return copy_plane(c, c->G + c->plane, out, 1);
        case LBMPM_RK_FX: return copy_plane(c, c->F, out, 1);
        case LBMPM_RK_FY: return copy_plane(c, c->F + c->plane, out, 1);
        case LBMPM_RK_VX: case LBMPM_RK_VY: case LBMPM_RK_K:
            if (!c->diag || !c->diag_valid) {
                set_error("field %d needs lbmpm_rk2d_enable_diagnostics(ctx, 1) before the last lbmpm_rk2d_step", field);
                return LBMPM_ERR_STATE;
            }
            return copy_plane(c, c->diag + (field == LBMPM_RK_VX ? 0 : field == LBMPM_RK_VY ? 1 : 2) * c->plane, out, 1);
        default: break;
    }
    set_error("lbmpm_rk2d_get_field: unknown field id %d", field);
    return LBMPM_ERR_INVALID;
}

namespace {
bool invert5(const double in[25], double out[25])
{   // Gauss-Jordan with partial pivoting
    double a[5][10];
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) { a[i][j] = in[5 * i + j]; a[i][5 + j] = (i == j) ? 1. : 0.; }
    for (int c = 0; c < 5; ++c) {
        int piv = c;
        for (int r = c + 1; r < 5; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (fabs(a[piv][c]) < 1e-300) return false;
        if (piv != c) for (int j = 0; j < 10; ++j) std::swap(a[piv][j], a[c][j]);
        const double d = a[c][c];
        for (int j = 0; j < 10; ++j) a[c][j] /= d;
        for (int r = 0; r < 5; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.) for (int j = 0; j < 10; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) out[5 * i + j] = a[i][5 + j];
    return true;
}
}  // namespace

extern "C" int lbmpm_rk2d_set_perturbation(lbmpm_rk2d *c, const lbmpm_rk2d_perturbation *par)
{
    LBMPM_REQUIRE(c && par, "lbmpm_rk2d_set_perturbation: null argument");
    LBMPM_REQUIRE(c->cfg.variant == 0 && c->ntr == 0, "the perturbation operator runs as the fused schedule, without tracers");
    // only the parameters the configured inlet / outlet kernels use are checked (the kernel-level loop runs an ini whose unused ones are odd)
    LBMPM_REQUIRE(std::isfinite(par->ak_r) && std::isfinite(par->ak_b) && std::isfinite(par->solid_phi), "lbmpm_rk2d_set_perturbation: A_k and solidPhi must be finite");
    LBMPM_REQUIRE(c->cfg.outlet_type != LBMPM_OUTLET_PRESSURE || (par->outlet_rho_r > 0. && par->outlet_rho_b > 0.),
                  "lbmpm_rk2d_set_perturbation: the pressure outlet needs outlet_rho_r, outlet_rho_b > 0");
    LBMPM_REQUIRE(c->cfg.inlet_type != LBMPM_INLET_VELOCITY || (par->inlet_velocity_y_r > -1. && par->inlet_velocity_y_b > -1.),
                  "lbmpm_rk2d_set_perturbation: the velocity inlet needs inlet velocities > -1");
    LBMPM_REQUIRE(c->cfg.inlet_type == LBMPM_INLET_VELOCITY || (c->cfg.inlet_rho_r > 0. && c->cfg.inlet_rho_b > 0.),
                  "lbmpm_rk2d_set_perturbation: the pressure inlet needs inlet_rho_r, inlet_rho_b > 0");
    // the reference addresses the pressure outlet's rows by COMPACT index (n < nx, nx <= n < 2 nx: A:1008-1081) and every ghost / copied
    // row takes its neighbour row through the neighbour table: both mean grid rows only when the rows at either end hold no solid node
    const int low = c->cfg.outlet_type == LBMPM_OUTLET_PRESSURE ? 2 : 4;
    for (int y = 0; y < c->ny; ++y) {
        if (y >= low && y < c->ny - 2) continue;
        for (int x = 0; x < c->nx; ++x)
            if (c->h_domain[(size_t)y * c->nx + x] != 1) {
                set_error("the fused perturbation step needs the rows 0 .. %d, ny-2 and ny-1 free of solid nodes (node (%d, %d) is not fluid): the "
                          "reference's boundary kernels reach those rows by compact node number / through the neighbour table", low - 1, x, y);
                return LBMPM_ERR_UNSUPPORTED;
            }
    }
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    c->pert = *par;
    c->model = 1;
    if (c->diag && !c->pd) { const int rc = dev_alloc(c, &c->pd, 4 * c->plane); if (rc) return rc; }
    c->diag_valid = false;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_tracer_configure(lbmpm_rk2d *c, const lbmpm_tracer_config *t)
{
    LBMPM_REQUIRE(c && t, "lbmpm_rk2d_tracer_configure: null argument");
    LBMPM_REQUIRE(t->num_tracers >= 1 && t->num_tracers <= 4, "NumberTracers must be 1..4 (got %d)", t->num_tracers);
    LBMPM_REQUIRE(c->cfg.variant == 0, "tracer transport is fused into the default (fused) schedule only");
    LBMPM_REQUIRE(c->model == 0, "tracer transport rides on the CSF step (Transport2DRK.py); the perturbation operator has no tracer loop in the reference");
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    // D2Q5 moment matrix and collision matrix -M^-1 S^-1 (Transport2DRK.py:316-347)
    const double M[25] = {1, 1, 1, 1, 1,   0, 1, -1, 0, 0,   0, 0, 0, 1, -1,   4, -1, -1, -1, -1,   0, 1, 1, -1, -1};
    double Minv[25];
    if (!invert5(M, Minv)) { set_error("singular D2Q5 matrix"); return LBMPM_ERR_INVALID; }
    memcpy(c->trM, M, sizeof(M));
    for (int k = 0; k < t->num_tracers; ++k) {
        double S[25] = {0}, Sinv[25];
        S[0] = 1.; S[18] = 1.; S[24] = 1.;
        S[6] = 0.5 + 3. * t->diffusion_x[k]; S[12] = 0.5 + 3. * t->diffusion_y[k];
        S[7] = 3. * t->diffusion_xy; S[11] = 3. * t->diffusion_yx;
        if (!invert5(S, Sinv)) { set_error("singular tracer relaxation matrix"); return LBMPM_ERR_INVALID; }
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                double v = 0.;
                for (int m = 0; m < 5; ++m) v += Minv[5 * i + m] * Sinv[5 * m + j];
                c->trA[k][5 * i + j] = -v;
            }
        c->trBeta[k] = t->beta_interface[k];
        c->trCb[k] = t->inlet_concentration[k];
        c->trJ[k] = t->diffusion_j[k];
    }
    LBMPM_REQUIRE(t->reaction_rate == 0. || t->num_tracers == 3, "the reaction couples exactly three tracers (A + B -> C)");
    c->trRate = t->reaction_rate;
    c->trCrit = t->criteria_rho; c->trFree = t->free_outlet ? 1 : 0; c->trDirichlet = t->dirichlet_inlet ? 1 : 0;
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->gA) { (void)hipFree(c->gA); (void)hipFree(c->gB); c->gA = c->gB = nullptr; c->bytes -= (int64_t)(2 * c->ntr * 5 * c->plane * sizeof(double)); }
    c->ntr = t->num_tracers;
    int rc = dev_alloc(c, &c->gA, (size_t)c->ntr * 5 * c->plane);
    if (rc == LBMPM_OK) rc = dev_alloc(c, &c->gB, (size_t)c->ntr * 5 * c->plane);
    return rc;
}

extern "C" int lbmpm_rk2d_tracer_set_concentration(lbmpm_rk2d *c, int tracer, const double *conc)
{
    LBMPM_REQUIRE(c && conc, "lbmpm_rk2d_tracer_set_concentration: null argument");
    LBMPM_REQUIRE(tracer >= 0 && tracer < c->ntr, "tracer index %d out of range (configured: %d)", tracer, c->ntr);
    LBMPM_REQUIRE(!c->streamed, "set the tracer concentration before the first time step (after lbmpm_rk2d_set_pdf/_set_macro)");
    static const double W5[5] = {1. / 3., 1. / 6., 1. / 6., 1. / 6., 1. / 6.};
    std::vector<double> h(5 * c->plane, 0.0);
    for (int y = 0; y < c->ny; ++y)
        for (int x = 0; x < c->nx; ++x) {
            const size_t k = (size_t)y * c->nx + x;
            if (c->h_domain[k] != 1) continue;
            for (int j = 0; j < 5; ++j) h[j * c->plane + (size_t)y * c->pitch + x] = conc[k] * W5[j];   // g = C w
        }
    LBMPM_HIP_TRY(hipMemcpyAsync(c->gA + (size_t)tracer * 5 * c->plane, h.data(), h.size() * sizeof(double),
                                 hipMemcpyHostToDevice, c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    return LBMPM_OK;
}

extern "C" int lbmpm_rk2d_tracer_get_concentration(lbmpm_rk2d *c, int tracer, double *out)
{
    LBMPM_REQUIRE(c && out, "lbmpm_rk2d_tracer_get_concentration: null argument");
    LBMPM_REQUIRE(tracer >= 0 && tracer < c->ntr, "tracer index %d out of range (configured: %d)", tracer, c->ntr);
    LBMPM_HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->obs) { const int rc = dev_alloc(c, &c->obs, 22 * c->plane); if (rc) return rc; }
    RKDev p = make_dev(c);
    rk2d_observe_tracer<<<grid_of(c), dim3(BX, BY), 0, c->stream>>>(p, tracer, c->obs);
    LBMPM_HIP_TRY(hipGetLastError());
    return copy_plane(c, c->obs, out, 1);
}

#if defined(LBMPM_DEV) && defined(LBMPM_PHASES2D)
extern "C" int lbmpm_dev_rk2d_phases(unsigned long long *out16)
{
    static unsigned long long h[256 * 16];
    LBMPM_HIP_TRY(hipDeviceSynchronize());
    LBMPM_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(rk2d_ph), sizeof(h)));
    for (int k = 0; k < 16; ++k) { out16[k] = 0; for (int b = 0; b < 256; ++b) out16[k] += h[b * 16 + k]; }
    memset(h, 0, sizeof(h));
    LBMPM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rk2d_ph), h, sizeof(h)));
    return LBMPM_OK;
}
#endif
extern "C" int64_t lbmpm_rk2d_num_fluid_nodes(const lbmpm_rk2d *c) { return c ? c->nfluid : 0; }
extern "C" int64_t lbmpm_rk2d_steps_done(const lbmpm_rk2d *c) { return c ? c->steps : 0; }
extern "C" int64_t lbmpm_rk2d_device_bytes(const lbmpm_rk2d *c) { return c ? c->bytes : 0; }
extern "C" const char *lbmpm_rk2d_dominant_kernel(const lbmpm_rk2d *c)
{
    if (c && c->model == 1) return "rk2dp_fused";
    return (c && c->cfg.variant == 0) ? "rk2d_fused" : "rk2d_collide_stream";
}
