// Test infrastructure: mi::fail() for the host-only sanitizer build (tests/test_host_sanitizers.py).  The product's
// version lives in csrc/runtime.hip together with the HIP runtime glue, which has no place in a CPU-only library.
#include <cstdarg>
#include <cstdio>

namespace mi
{
    static thread_local char g_msg[512];

    int fail(int code, const char *fmt, ...)
    {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(g_msg, sizeof(g_msg), fmt, ap);
        va_end(ap);
        return code;
    }
}

extern "C" const char *mi_dspu_last_error(void) { return mi::g_msg; }
