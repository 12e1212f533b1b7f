// rk3d_state.h -- state in and out of the D3Q19 solver; included by rk3d.hip inside its anonymous namespace.
//
// The reference's 2-D drivers start a run from densities + velocity through f = rho w (1 + 3 e.u + 4.5 (e.u)^2 - 1.5 u^2)
// (RKD2Q9.py:577-601), restart one from recorded populations ([CyclesSetup] IsCycle = 'yes', RKD2Q9.py:491-559: fluidPDFR / fluidPDFB
// [ny][nx][9]) and record the populations (RKD2Q9.py:938-957).  Their 3-D counterparts for the three storages of this file:
//   macro  -> state      f_c,i = rho_c w_i (1 + 3 e.u + ...)                                   (lbmpm_rk3d_set_macro)
//   pdf   <-> state      [nz_local][ny][nx][19] per colour, zeros off the fluid                (lbmpm_rk3d_set_pdf / get_pdf)
//   state <-> state      the S doubles a cell stores, as stored: exact restart                 (lbmpm_rk3d_set_state / get_state)
// S = 38 for the two 38-value storages ({f_R,i}, {f_B,i}); S = 23 for the q23 storage: g_0..18, k_R, A_x, A_y, A_z (rk3dq.h).  There
// the two colour lattices are an affine image of the 23 -- f_R,i = k_R g_i + c_i e_i.A, f_B,i = g_i - f_R,i -- which get_pdf
// evaluates; set_pdf inverts it: g_i = f_R,i + f_B,i, k_R = rho_R / rho, A = (sum_i f_R,i e_i - k_R sum_i g_i e_i) / (2 c_axis + 8 c_diag)
// (exact for populations this model wrote, colours swapped or not; the least-squares projection onto the model's states otherwise).
// That round trip is good to rounding, not to the bit: a bit-exact restart goes through get_state / set_state.
//
// All kernels work on a batch of whole planes staged in device memory: one thread per lattice cell, host layout [plane][y][x][...].
enum { ST_DENSE = 0, ST_C38 = 1, ST_Q23 = 2 };
enum { IO_GET_STATE = 0, IO_SET_STATE = 1, IO_GET_PDF = 2, IO_SET_PDF = 3, IO_SET_MACRO = 4 };

template <int KIND> struct StateWidth { static constexpr int S = KIND == ST_Q23 ? QS : 2 * Q; };

// raw stored values of the fluid cell (x, y) of plane zl; j = its number inside the plane (compact storages)
template <int KIND>
__device__ __forceinline__ void state_load(const RK3Dev &p, const double *f, int zl, int x, int y, int sg, unsigned j, double *v)
{
    if (KIND == ST_Q23) {
        const unsigned long long p0 = pstart_of(p, zl);
        const size_t cnt = (size_t)(pstart_of(p, zl + 1) - p0);
        const double *pl = f + (size_t)p0 * QS;
        for (int i = 0; i < Q; ++i) v[i] = pl[(size_t)i * cnt + j];
        const unsigned flag = p.pur_in[row_index(p, zl, y, sg)];
        const double *s = pl + (size_t)Q * cnt + (size_t)j * 4;
        if (flag) { v[Q] = (flag & 1u) ? 1. : 0.; v[Q + 1] = 0.; v[Q + 2] = 0.; v[Q + 3] = 0.; }      // a flagged row keeps no records
        else { v[Q] = s[0]; v[Q + 1] = s[1]; v[Q + 2] = s[2]; v[Q + 3] = s[3]; }
    } else if (KIND == ST_C38) {
        const unsigned long long p0 = p.pstart[zl];
        const size_t cnt = (size_t)(p.pstart[zl + 1] - p0);
        const double2 *pl = reinterpret_cast<const double2 *>(f) + (size_t)p0 * Q;
        for (int i = 0; i < Q; ++i) { const double2 t = pl[(size_t)i * cnt + j]; v[i] = t.x; v[Q + i] = t.y; }
    } else {
        const double *pl = f + (size_t)zl * 2 * Q * p.plane2 + (size_t)y * p.pitch + x;
        for (int i = 0; i < 2 * Q; ++i) v[i] = pl[(size_t)i * p.plane2];
    }
}

template <int KIND>
__device__ __forceinline__ void state_store(const RK3Dev &p, double *f, int zl, int x, int y, unsigned j, const double *v)
{
    if (KIND == ST_Q23) {
        const unsigned long long p0 = pstart_of(p, zl);
        const size_t cnt = (size_t)(pstart_of(p, zl + 1) - p0);
        double *pl = f + (size_t)p0 * QS;
        for (int i = 0; i < Q; ++i) pl[(size_t)i * cnt + j] = v[i];
        double *s = pl + (size_t)Q * cnt + (size_t)j * 4;
        s[0] = v[Q]; s[1] = v[Q + 1]; s[2] = v[Q + 2]; s[3] = v[Q + 3];
    } else if (KIND == ST_C38) {
        const unsigned long long p0 = p.pstart[zl];
        const size_t cnt = (size_t)(p.pstart[zl + 1] - p0);
        double2 *pl = reinterpret_cast<double2 *>(f) + (size_t)p0 * Q;
        for (int i = 0; i < Q; ++i) { double2 t; t.x = v[i]; t.y = v[Q + i]; pl[(size_t)i * cnt + j] = t; }
    } else {
        double *pl = f + (size_t)zl * 2 * Q * p.plane2 + (size_t)y * p.pitch + x;
        for (int i = 0; i < 2 * Q; ++i) pl[(size_t)i * p.plane2] = v[i];
    }
}

// stored values <-> the two colour lattices
template <int KIND>
__device__ __forceinline__ void state_to_pdf(const RK3Dev &p, const double *v, double fR[Q], double fB[Q])
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    if (KIND != ST_Q23) { for (int i = 0; i < Q; ++i) { fR[i] = v[i]; fB[i] = v[Q + i]; } return; }
    for (int i = 0; i < Q; ++i) {
        const double ea = (double)CX[i] * v[Q + 1] + (double)CY[i] * v[Q + 2] + (double)CZ[i] * v[Q + 3];
        const double c = i == 0 ? 0. : (i < 7 ? p.rcA : p.rcD);
        fR[i] = v[Q] * v[i] + c * ea;
        fB[i] = v[i] - fR[i];
    }
}
template <int KIND>
__device__ __forceinline__ void pdf_to_state(const RK3Dev &p, const double fR[Q], const double fB[Q], double *v)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    if (KIND != ST_Q23) { for (int i = 0; i < Q; ++i) { v[i] = fR[i]; v[Q + i] = fB[i]; } return; }
    double rR = 0., rB = 0., mx = 0., my = 0., mz = 0., tx = 0., ty = 0., tz = 0.;
    for (int i = 0; i < Q; ++i) {
        const double g = fR[i] + fB[i];
        v[i] = g;
        rR += fR[i]; rB += fB[i];
        mx += (double)CX[i] * fR[i]; my += (double)CY[i] * fR[i]; mz += (double)CZ[i] * fR[i];
        tx += (double)CX[i] * g; ty += (double)CY[i] * g; tz += (double)CZ[i] * g;
    }
    const double rho = rR + rB, kR = rB == 0. ? 1. : (rR == 0. ? 0. : rR / rho);
    const double den = 2. * p.rcA + 8. * p.rcD, iden = den != 0. ? 1. / den : 0.;
    const bool pure = rB == 0. || rR == 0.;
    v[Q] = kR;
    v[Q + 1] = pure ? 0. : (mx - kR * tx) * iden;
    v[Q + 2] = pure ? 0. : (my - kR * ty) * iden;
    v[Q + 3] = pure ? 0. : (mz - kR * tz) * iden;
}

// one thread per cell of the planes zl0 .. zl0 + gridDim.z - 1; `a`, `b`: the staged host arrays of the batch
//   IO_GET_STATE / IO_SET_STATE: a = [planes][ny][nx][S]
//   IO_GET_PDF / IO_SET_PDF:     a = f_R, b = f_B, each [planes][ny][nx][19]
//   IO_SET_MACRO:                a = [5][planes][ny][nx]: rho_R, rho_B, vx, vy, vz
template <int KIND, int MODE>
__global__ __launch_bounds__(BX3 *BY3) void rk3d_state_io(RK3Dev p, double *f, int zl0, double *a, double *b)
{
    constexpr int S = StateWidth<KIND>::S;
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const int sg = blockIdx.x, y = blockIdx.y * BY3 + threadIdx.y, pz = blockIdx.z, zl = zl0 + pz;
    const int x = KIND == ST_DENSE ? (int)(blockIdx.x * 64 + threadIdx.x) : LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x);
    if (y >= p.ny || x < 0 || x >= p.nx) return;
    const size_t cell = ((size_t)pz * p.ny + y) * p.nx + x, cells = (size_t)gridDim.z * p.ny * p.nx;
    const bool fluid = p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1;
    constexpr bool GET = MODE == IO_GET_STATE || MODE == IO_GET_PDF;
    if (!fluid) {
        if (MODE == IO_GET_STATE) for (int k = 0; k < S; ++k) a[cell * S + k] = 0.;
        if (MODE == IO_GET_PDF) for (int i = 0; i < Q; ++i) { a[cell * Q + i] = 0.; b[cell * Q + i] = 0.; }
        return;
    }
    unsigned j = 0;
    if (KIND != ST_DENSE) {
        const GlobalRows rows{p, sg, y};
        const RowTab t = rows(zl, 0);
        j = t.first + bits_below<false>(t.m, threadIdx.x);
    }
    double v[S], fR[Q], fB[Q];
    if (GET) {
        state_load<KIND>(p, f, zl, x, y, sg, j, v);
        if (MODE == IO_GET_STATE) { for (int k = 0; k < S; ++k) a[cell * S + k] = v[k]; return; }
        state_to_pdf<KIND>(p, v, fR, fB);
        for (int i = 0; i < Q; ++i) { a[cell * Q + i] = fR[i]; b[cell * Q + i] = fB[i]; }
        return;
    }
    if (MODE == IO_SET_STATE) for (int k = 0; k < S; ++k) v[k] = a[cell * S + k];
    else {
        if (MODE == IO_SET_PDF) for (int i = 0; i < Q; ++i) { fR[i] = a[cell * Q + i]; fB[i] = b[cell * Q + i]; }
        else {
            const double rR = a[cell], rB = a[cells + cell], ux = a[2 * cells + cell], uy = a[3 * cells + cell], uz = a[4 * cells + cell];
            const double usq = ux * ux + uy * uy + uz * uz;
            for (int i = 0; i < Q; ++i) {
                const double eu = (double)CX[i] * ux + (double)CY[i] * uy + (double)CZ[i] * uz;
                const double poly = 1. + (3. * eu + 4.5 * eu * eu - 1.5 * usq);
                fR[i] = (rR * wq(i)) * poly;
                fB[i] = (rB * wq(i)) * poly;
            }
            if (KIND == ST_Q23) {
                // as rk3dq_init_rest: g_i = w_i (rho_R + rho_B) (...), k_R = rho_R / rho, A = 0 -- a state of the model by construction
                const double rho = rR + rB;
                for (int i = 0; i < Q; ++i) {
                    const double eu = (double)CX[i] * ux + (double)CY[i] * uy + (double)CZ[i] * uz;
                    v[i] = (wq(i) * rho) * (1. + (3. * eu + 4.5 * eu * eu - 1.5 * usq));
                }
                v[Q] = rR / rho; v[Q + 1] = 0.; v[Q + 2] = 0.; v[Q + 3] = 0.;
                state_store<KIND>(p, f, zl, x, y, j, v);
                return;
            }
        }
        pdf_to_state<KIND>(p, fR, fB, v);
    }
    state_store<KIND>(p, f, zl, x, y, j, v);
}

// q23: the row flags of the planes zl0 .. from the records just stored (every row holds records at this point): bit 0 / 1 = every
// fluid cell of the segment pure red / pure blue, 3 = no fluid cell -- what the collision writes (collide_store) and rk3dq_init_rest
__global__ __launch_bounds__(BX3 *BY3) void rk3dq_flags_from_records(RK3Dev p, const double *f, uint32_t *pur, int zl0)
{
    const int sg = blockIdx.x, x = LBMPM_SEG_LANE_X(p, sg, (int)threadIdx.x), y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + zl0;
    if (y >= p.ny) return;
    const bool fluid = x >= 0 && (p.flags[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] & 1);
    bool notred = false, notblue = false;
    if (fluid) {
        const GlobalRows rows{p, sg, y};
        const RowTab t = rows(zl, 0);
        const unsigned j = t.first + bits_below<false>(t.m, threadIdx.x);
        const unsigned long long p0 = pstart_of(p, zl);
        const size_t cnt = (size_t)(pstart_of(p, zl + 1) - p0);
        const double *s = f + (size_t)p0 * QS + (size_t)Q * cnt + (size_t)j * 4;
        const bool noA = s[1] == 0. && s[2] == 0. && s[3] == 0.;
        notred = !(s[0] == 1. && noA); notblue = !(s[0] == 0. && noA);
    }
    const unsigned code = (__ballot(notred) == 0ull ? 1u : 0u) | (__ballot(notblue) == 0ull ? 2u : 0u);
    if (threadIdx.x == 0) pur[row_index(p, zl, y, sg)] = code;
}
