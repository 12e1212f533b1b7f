// lbmpm_common.hip -- error state, version, event pool.
#include "lbmpm_common.h"
#include "d2q9_device.h"

namespace lbmpm {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int EventPool::reserve(size_t pairs)
{
    while (ev.size() < 2 * pairs) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return LBMPM_ERR_HIP;
        ev.push_back(e);
    }
    return LBMPM_OK;
}

bool EventPool::take(hipEvent_t *a, hipEvent_t *b)
{
    if (used + 2 > ev.size()) return false;
    *a = ev[used];
    *b = ev[used + 1];
    used += 2;
    return true;
}

double EventPool::sum_ms()
{
    double s = 0.0;
    for (size_t k = 0; k + 1 < used; k += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess) s += ms;
    }
    return s;
}

void EventPool::destroy()
{
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    used = 0;
}

}  // namespace lbmpm

extern "C" const char *lbmpm_last_error(void) { return lbmpm::g_err; }

extern "C" const char *lbmpm_version(void) { return "liblbmpm_hip 0.1.0 gfx950"; }

extern "C" int lbmpm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        lbmpm::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return LBMPM_ERR_HIP;
    }
    return n;
}

namespace lbmpm_dev {

__global__ void setup_solidnbr(int nx, int ny, int pitch, const uint8_t *flags, uint8_t *solidnbr)
{
    constexpr int EX[9] = LBMPM_D2Q9_EX, EY[9] = LBMPM_D2Q9_EY;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= nx || y >= ny) return;
    unsigned b = 0;
    for (int i = 1; i < 9; ++i) {
        const int xn = wrapi(x + EX[i], nx), yn = wrapi(y + EY[i], ny);
        if (!(flags[(size_t)yn * pitch + xn] & 1)) b |= 1u << (i - 1);
    }
    solidnbr[(size_t)y * pitch + x] = (uint8_t)b;
}

}  // namespace lbmpm_dev
