// lbmpm_common.h -- shared host/device helpers of liblbmpm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/lbmpm.h"

namespace lbmpm {

void set_error(const char *fmt, ...);

#define LBMPM_HIP_TRY(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::lbmpm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                               __FILE__, __LINE__);                                       \
            return LBMPM_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define LBMPM_REQUIRE(cond, ...)                                                          \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            ::lbmpm::set_error(__VA_ARGS__);                                              \
            return LBMPM_ERR_INVALID;                                                     \
        }                                                                                 \
    } while (0)

// D2Q9 lattice, ordering of RKCG2D/RKD2Q9.py:300-303 / ShanChen2D/SimpleD2Q9.py:226
//   i : 0      1     2      3      4      5     6      7       8
//   e : (0,0) (1,0) (0,1) (-1,0) (0,-1) (1,1) (-1,1) (-1,-1) (1,-1)
#define LBMPM_D2Q9_EX {0, 1, 0, -1, 0, 1, -1, -1, 1}
#define LBMPM_D2Q9_EY {0, 0, 1, 0, -1, 1, 1, -1, -1}
#define LBMPM_D2Q9_OPP {0, 3, 4, 1, 2, 7, 8, 5, 6}
#define LBMPM_D2Q9_W {4. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 9., 1. / 36., 1. / 36., 1. / 36., 1. / 36.}

// Lattice-constant arrays of the kernel-level entry points (include/lbmpm_kernels.h).  The reference passes its direction vectors and
// weights to every kernel as device arrays; the kernels here have them built in.  An array that does not hold the built-in values would
// silently be ignored, so each such argument is checked the first time its device pointer is seen (one small device-to-host copy; the
// verdict is cached per pointer): LBMPM_ERR_UNSUPPORTED with the first differing entry in the message.
enum LatticeTable { LT_D2Q9_EX = 0, LT_D2Q9_EY, LT_D2Q9_W, LT_D2Q5_VX, LT_D2Q5_VY };
int check_lattice_constant(hipStream_t st, const char *kernel, const char *arg, const double *device_array, int table);
void forget_lattice_constant(const void *device_ptr);   // the facade's free / host-to-device copy: the cached verdict no longer holds
                                                         // (memory that does not go through lbmpm_device_free / lbmpm_memcpy_h2d and is
                                                         // rewritten in place keeps its first verdict)

// Event pool used by the *_step_timed entry points: one (start, stop) pair per launch of
// the dominant kernel, all recorded on the stream that kernel runs on.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    int reserve(size_t pairs);
    void reset() { used = 0; }
    bool take(hipEvent_t *a, hipEvent_t *b);
    double sum_ms();   // after the stream was synchronised
    void destroy();
};

}  // namespace lbmpm
