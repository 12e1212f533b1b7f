// lbmpm_common.hip -- error state, version, event pool.
#include "lbmpm_common.h"

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
