// Error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace pngpd {
// Records the message for pngpd_last_error() on this thread and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace pngpd

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return pngpd::fail(PNGPD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                                  \
    } while (0)
