// rk3d.hip -- D3Q19 colour-gradient (perturbation operator) time stepper for gfx950, written
// for z-slab decomposition across the GPUs of a node.
//
// The reference ships only an ini for this model (IniFiles/RKtwophasesetup3D.ini; the module
// RKColorGradientD3Q19 imported by main.py:22 is absent), so the model is the D3Q19 extension
// of the reference's 2-D kernels, operator by operator (list + citations in
// oracle/rk3d_oracle.c and DESIGN.md).  PARITY UNPINNED against the reference; pinned
// against the independent CPU statement oracle/rk3d_oracle.c.
//
// Layout per rank: dense SoA f[c][q][zl][y][x], zl = 0..nzl+1 where zl = 1..nzl are the owned
// planes and zl = 0 / nzl+1 are halo planes holding the neighbour rank's outermost plane
// (only the five populations that cross the cut are ever filled / read).  x, y periodic;
// z not (planes 0 and nz-1 of the global lattice are boundary ghost planes).
// One time step = [f halo exchange] -> phase_field -> [phi halo exchange] -> collide.
#include "lbmpm_common.h"
#include "d2q9_device.h"

#include <cmath>
#include <cstdlib>

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

struct RK3Dev {
    int nx, ny, nzl, pitch;
    size_t plane2, vol;
    int z0, nzg;                 // global z of local plane zl is z0 + zl - 1
    const uint8_t *flags;        // [vol]
    const uint32_t *solidnbr;    // [vol], bit (i-1) <=> node + e_i is not fluid
    const double *fin;
    double *fout;
    double *phi;                 // [vol]
    double *diag;                // [5][vol] rhoR, rhoB, vx, vy, vz or nullptr
    double ak, beta, tauR, tauB, solidPhi, vzR, vzB, rhoOutR, rhoOutB;
    int first;
};

__device__ __forceinline__ void pull3(const RK3Dev &p, int x, int y, int zl, double fR[Q], double fB[Q])
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ, OPP[Q] = LBMPM_D3Q19_OPP;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    const double *fr = p.fin, *fb = p.fin + (size_t)Q * p.vol;
    const bool first = p.first != 0;
    const unsigned sn = first ? 0u : p.solidnbr[idx];
    fR[0] = fr[idx];
    fB[0] = fb[idx];
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        const int xs = wrapi(x - CX[i], p.nx), ys = wrapi(y - CY[i], p.ny), zs = zl - CZ[i];
        const size_t s = first ? idx : (size_t)zs * p.plane2 + (size_t)ys * p.pitch + xs;
        fR[i] = fr[(size_t)i * p.vol + s];
        fB[i] = fb[(size_t)i * p.vol + s];
    }
    if (sn != 0) {
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            const int o = OPP[i];
            if ((sn >> (o - 1)) & 1u) {
                fR[i] = fr[(size_t)o * p.vol + idx];
                fB[i] = fb[(size_t)o * p.vol + idx];
            }
        }
    }
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

// post-streaming, post-boundary state of node (x, y, zl)
__device__ __forceinline__ void node_state3(const RK3Dev &p, int x, int y, int zl, double fR[Q], double fB[Q],
                                            double &rR, double &rB)
{
    const int zg = p.z0 + zl - 1;
    int zs = zl;
    if (zg == p.nzg - 1) zs = zl - 1;       // ghost plane <- inlet plane
    if (zg == 0) zs = zl + 1;               // ghost plane <- outlet plane
    const int zsg = p.z0 + zs - 1;
    pull3(p, x, y, zs, fR, fB);
    rR = sum19(fR);
    rB = sum19(fB);
    if (zsg == p.nzg - 2) {
        rR = zouhe_inlet(p.vzR, fR);
        rB = zouhe_inlet(p.vzB, fB);
        if (zg == p.nzg - 1) { rR = sum19(fR); rB = sum19(fB); }
    }
    if (zsg == 1) {
        zouhe_outlet(p.rhoOutR, fR); rR = p.rhoOutR;
        zouhe_outlet(p.rhoOutB, fB); rB = p.rhoOutB;
    }
}

constexpr int BX3 = 64, BY3 = 4;

// K1: phase field of the streamed, boundary-corrected lattice on the owned planes
__global__ __launch_bounds__(BX3 *BY3) void rk3d_phase_field(RK3Dev p)
{
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[Q], fB[Q], rR, rB;
    node_state3(p, x, y, zl, fR, fB, rR, rB);
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

// K2 (dominant): stream + boundaries again, colour gradient, BGK, perturbation, recolouring, store
__global__ __launch_bounds__(BX3 *BY3) void rk3d_collide(RK3Dev p)
{
    constexpr int CX[Q] = LBMPM_D3Q19_CX, CY[Q] = LBMPM_D3Q19_CY, CZ[Q] = LBMPM_D3Q19_CZ;
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    if (!(p.flags[idx] & 1)) return;
    double fR[Q], fB[Q], rR, rB;
    node_state3(p, x, y, zl, fR, fB, rR, rB);
    const unsigned sn = p.solidnbr[idx];
    double gx = 0., gy = 0., gz = 0., mx = 0., my = 0., mz = 0.;
#pragma unroll
    for (int i = 1; i < Q; ++i) {
        double ph = p.solidPhi;
        if (!((sn >> (i - 1)) & 1u)) {
            const size_t n = (size_t)(zl + CZ[i]) * p.plane2 + (size_t)wrapi(y + CY[i], p.ny) * p.pitch + wrapi(x + CX[i], p.nx);
            ph = p.phi[n];
        }
        gx += 3. * wq(i) * (double)CX[i] * ph;
        gy += 3. * wq(i) * (double)CY[i] * ph;
        gz += 3. * wq(i) * (double)CZ[i] * ph;
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const double t = fR[i] + fB[i];
        mx += (double)CX[i] * t; my += (double)CY[i] * t; mz += (double)CZ[i] * t;
    }
    const double rho = rR + rB;
    const double ux = mx / rho, uy = my / rho, uz = mz / rho, usq = ux * ux + uy * uy + uz * uz;
    const double phi = (rR - rB) / (rR + rB);
    const double tau = 0.5 + 1. / ((1. + phi) / (2. * (p.tauR - 0.5)) + (1. - phi) / (2. * (p.tauB - 0.5)));
    const double g2 = gx * gx + gy * gy + gz * gz, gn = sqrt(g2);
    const double kR = rR / rho, kB = rB / rho, arc = p.beta * rR * rB / (rho * rho);
    double *fr = p.fout, *fb = p.fout + (size_t)Q * p.vol;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const double eu = (double)CX[i] * ux + (double)CY[i] * uy + (double)CZ[i] * uz;
        const double feq = rho * wq(i) * (1. + 3. * eu + 4.5 * eu * eu - 1.5 * usq);
        double ft = fR[i] + fB[i];
        ft = ft - (ft - feq) / tau;
        const double eg = (double)CX[i] * gx + (double)CY[i] * gy + (double)CZ[i] * gz;
        if (g2 != 0.) ft += p.ak * gn * (wq(i) * (eg * eg) / g2 - bq(i));
        const double en = (i == 0) ? 0. : (i < 7 ? 1. : sqrt(2.));
        const double c = (en == 0. || gn == 0.) ? 0. : eg / (en * gn);
        const double a = arc * wq(i) * c;
        fr[(size_t)i * p.vol + idx] = kR * ft + a;
        fb[(size_t)i * p.vol + idx] = kB * ft - a;
    }
}

// halo packing: the five populations per colour that cross each cut
__global__ void rk3d_pack(RK3Dev p, const double *f, double *send_up, double *send_dn)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.plane2) return;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 5; ++j) {
            send_up[(size_t)(c * 5 + j) * p.plane2 + k] = f[((size_t)c * Q + UP[j]) * p.vol + (size_t)p.nzl * p.plane2 + k];
            send_dn[(size_t)(c * 5 + j) * p.plane2 + k] = f[((size_t)c * Q + DN[j]) * p.vol + p.plane2 + k];
        }
}

__global__ void rk3d_unpack(RK3Dev p, double *f, const double *recv_from_below, const double *recv_from_above,
                            int have_below, int have_above)
{
    constexpr int UP[5] = {5, 11, 14, 15, 18}, DN[5] = {6, 12, 13, 16, 17};
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.plane2) return;
    for (int c = 0; c < 2; ++c)
        for (int j = 0; j < 5; ++j) {
            if (have_below) f[((size_t)c * Q + UP[j]) * p.vol + k] = recv_from_below[(size_t)(c * 5 + j) * p.plane2 + k];
            if (have_above) f[((size_t)c * Q + DN[j]) * p.vol + (size_t)(p.nzl + 1) * p.plane2 + k] = recv_from_above[(size_t)(c * 5 + j) * p.plane2 + k];
        }
}

// f = w rho at rest on the owned planes (3-D analogue of RKD2Q9.py:577-601); rho arrays are dense
// [nzl][ny][nx] on the device
__global__ void rk3d_init_rest(RK3Dev p, const double *rho_r, const double *rho_b, double *f)
{
    const int x = blockIdx.x * BX3 + threadIdx.x, y = blockIdx.y * BY3 + threadIdx.y, zl = blockIdx.z + 1;
    if (x >= p.nx || y >= p.ny) return;
    const size_t idx = (size_t)zl * p.plane2 + (size_t)y * p.pitch + x;
    const bool fluid = p.flags[idx] & 1;
    const size_t s = ((size_t)(zl - 1) * p.ny + y) * p.nx + x;
    const double a = fluid ? rho_r[s] : 0., b = fluid ? rho_b[s] : 0.;
    for (int i = 0; i < Q; ++i) {
        f[(size_t)i * p.vol + idx] = wq(i) * a;
        f[((size_t)Q + i) * p.vol + idx] = wq(i) * b;
    }
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
    solidnbr[(size_t)zl * p.plane2 + (size_t)y * p.pitch + x] = b;
}

}  // namespace

// ====================================================================== host side
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
    double *send_up = nullptr, *send_dn = nullptr, *recv_below = nullptr, *recv_above = nullptr;
    std::vector<uint8_t> h_domain;   // owned planes only, [nzl][ny][nx]
    bool streamed = false;
    int64_t steps = 0, bytes = 0;
    lbmpm::EventPool pool;
};

namespace {

RK3Dev make_dev(const lbmpm_rk3d *c)
{
    RK3Dev p{};
    p.nx = c->nx; p.ny = c->ny; p.nzl = c->nzl; p.pitch = c->pitch; p.plane2 = c->plane2; p.vol = c->vol;
    p.z0 = (int)c->cfg.z_offset; p.nzg = (int)c->cfg.nz_global;
    p.flags = c->flags; p.solidnbr = c->solidnbr; p.fin = c->fA; p.fout = c->fB; p.phi = c->phi; p.diag = nullptr;
    p.ak = (c->cfg.ak_r + c->cfg.ak_b) * 0.5; p.beta = c->cfg.beta; p.tauR = c->cfg.tau_r; p.tauB = c->cfg.tau_b;
    p.solidPhi = c->cfg.solid_phi; p.vzR = c->cfg.inlet_vz_r; p.vzB = c->cfg.inlet_vz_b;
    p.rhoOutR = c->cfg.outlet_rho_r; p.rhoOutB = c->cfg.outlet_rho_b;
    p.first = c->streamed ? 0 : 1;
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
    LBMPM_REQUIRE((double)cfg->nx * cfg->ny * (cfg->nz_local + 2) < 2.0e9, "slab too large for 32-bit plane indices");
    LBMPM_HIP_TRY(hipSetDevice(cfg->device));
    lbmpm_rk3d *c = new (std::nothrow) lbmpm_rk3d();
    if (!c) { set_error("out of host memory"); return LBMPM_ERR_NOMEM; }
    c->cfg = *cfg;
    c->nx = (int)cfg->nx; c->ny = (int)cfg->ny; c->nzl = (int)cfg->nz_local;
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
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return LBMPM_ERR_HIP; }
        c->own_stream = true;
    }
    int rc = LBMPM_OK;
#define TRY_RC(e) do { rc = (e); if (rc != LBMPM_OK) { lbmpm_rk3d_destroy(c); return rc; } } while (0)
    TRY_RC(dev_alloc(c, &c->flags, c->vol));
    TRY_RC(dev_alloc(c, &c->solidnbr, c->vol));
    TRY_RC(dev_alloc(c, &c->fA, 2 * Q * c->vol));
    TRY_RC(dev_alloc(c, &c->fB, 2 * Q * c->vol));
    TRY_RC(dev_alloc(c, &c->phi, c->vol));
    TRY_RC(dev_alloc(c, &c->send_up, 10 * c->plane2));
    TRY_RC(dev_alloc(c, &c->send_dn, 10 * c->plane2));
    TRY_RC(dev_alloc(c, &c->recv_below, 10 * c->plane2));
    TRY_RC(dev_alloc(c, &c->recv_above, 10 * c->plane2));
#undef TRY_RC
    hipError_t e = hipMemcpyAsync(c->flags, hflags.data(), c->vol, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_error("flags upload failed: %s", hipGetErrorString(e)); lbmpm_rk3d_destroy(c); return LBMPM_ERR_HIP; }
    RK3Dev p = make_dev(c);
    rk3d_setup_solidnbr<<<grid3(c, c->nzl + 2), dim3(BX3, BY3), 0, c->stream>>>(p, c->solidnbr);
    e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("set-up kernel failed: %s", hipGetErrorString(e)); lbmpm_rk3d_destroy(c); return LBMPM_ERR_HIP; }
    *out = c;
    return LBMPM_OK;
}

extern "C" void lbmpm_rk3d_destroy(lbmpm_rk3d *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *ptr : {(void *)c->flags, (void *)c->solidnbr, (void *)c->fA, (void *)c->fB, (void *)c->phi, (void *)c->diag,
                      (void *)c->send_up, (void *)c->send_dn, (void *)c->recv_below, (void *)c->recv_above})
        if (ptr) (void)hipFree(ptr);
    c->pool.destroy();
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
    // stage the two density fields in the (idle) second population buffer, expand on the device
    const size_t n = (size_t)c->nx * c->ny * c->nzl;
    double *stage = c->fB;
    LBMPM_HIP_TRY(hipMemcpyAsync(stage, rho_r, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LBMPM_HIP_TRY(hipMemcpyAsync(stage + n, rho_b, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LBMPM_HIP_TRY(hipMemsetAsync(c->fA, 0, 2 * Q * c->vol * sizeof(double), c->stream));
    RK3Dev p = make_dev(c);
    rk3d_init_rest<<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p, stage, stage + n, c->fA);
    LBMPM_HIP_TRY(hipGetLastError());
    LBMPM_HIP_TRY(hipMemsetAsync(c->fB, 0, 2 * Q * c->vol * sizeof(double), c->stream));
    LBMPM_HIP_TRY(hipStreamSynchronize(c->stream));
    c->streamed = false;
    c->steps = 0;
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_pack_halo(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    RK3Dev p = make_dev(c);
    const int threads = 256;
    rk3d_pack<<<dim3((unsigned)((c->plane2 + threads - 1) / threads)), dim3(threads), 0, c->stream>>>(p, c->fA, c->send_up, c->send_dn);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_unpack_halo(lbmpm_rk3d *c, int have_below, int have_above)
{
    LBMPM_REQUIRE(c, "null context");
    RK3Dev p = make_dev(c);
    const int threads = 256;
    rk3d_unpack<<<dim3((unsigned)((c->plane2 + threads - 1) / threads)), dim3(threads), 0, c->stream>>>(
        p, c->fA, c->recv_below, c->recv_above, have_below, have_above);
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
    rk3d_phase_field<<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p);
    LBMPM_HIP_TRY(hipGetLastError());
    return LBMPM_OK;
}

extern "C" int lbmpm_rk3d_collide(lbmpm_rk3d *c)
{
    LBMPM_REQUIRE(c, "null context");
    RK3Dev p = make_dev(c);
    rk3d_collide<<<grid3(c, c->nzl), dim3(BX3, BY3), 0, c->stream>>>(p);
    LBMPM_HIP_TRY(hipGetLastError());
    std::swap(c->fA, c->fB);
    c->streamed = true;
    c->steps += 1;
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

extern "C" int lbmpm_rk3d_buffer(lbmpm_rk3d *c, int which, void **ptr, int64_t *bytes)
{
    LBMPM_REQUIRE(c && ptr && bytes, "lbmpm_rk3d_buffer: null argument");
    const int64_t fb = (int64_t)(10 * c->plane2 * sizeof(double)), pb = (int64_t)(c->plane2 * sizeof(double));
    switch (which) {
        case LBMPM_RK3D_BUF_F_SEND_UP: *ptr = c->send_up; *bytes = fb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_SEND_DOWN: *ptr = c->send_dn; *bytes = fb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_RECV_FROM_BELOW: *ptr = c->recv_below; *bytes = fb; return LBMPM_OK;
        case LBMPM_RK3D_BUF_F_RECV_FROM_ABOVE: *ptr = c->recv_above; *bytes = fb; return LBMPM_OK;
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

extern "C" int64_t lbmpm_rk3d_num_fluid_nodes(const lbmpm_rk3d *c) { return c ? c->nfluid : 0; }
extern "C" int64_t lbmpm_rk3d_steps_done(const lbmpm_rk3d *c) { return c ? c->steps : 0; }
extern "C" int64_t lbmpm_rk3d_device_bytes(const lbmpm_rk3d *c) { return c ? c->bytes : 0; }
extern "C" const char *lbmpm_rk3d_dominant_kernel(const lbmpm_rk3d *c) { (void)c; return "rk3d_collide"; }
