// Shared host-side plumbing for the C-ABI implementation (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mi_dspu.h"

namespace mi
{
    // Thread-local message behind mi_dspu_last_error().
    char       *error_buffer();
    int         fail(int code, const char *fmt, ...);

    inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

    // Events armed by mi_dspu_profile_next_launch(); the next hot-path kernel launch of this thread
    // consumes them (hipExtLaunchKernelGGL records them at the kernel's own begin/end).
    void        take_profile_events(hipEvent_t *start, hipEvent_t *stop);

    // Device twiddle table exp(-2 pi i j / twn), one per device, created on first use (convolver.hip).
    int         fft_twiddles(const float2 **tw, int *twn);
} // namespace mi

#define MI_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t mi_err__ = (expr);                                                   \
        if (mi_err__ != hipSuccess)                                                     \
            return ::mi::fail((mi_err__ == hipErrorNoDevice || mi_err__ == hipErrorInvalidDevice) \
                                  ? MI_ENODEV : MI_EHIP,                                \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(mi_err__), \
                              __FILE__, __LINE__);                                      \
    } while (0)

#define MI_REQUIRE(cond, code, ...)                                                     \
    do { if (!(cond)) return ::mi::fail((code), __VA_ARGS__); } while (0)
