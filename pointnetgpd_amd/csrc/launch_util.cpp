// Per-device launch facts that do not change while the process lives, looked up once instead of on every launch.
#include <hip/hip_runtime.h>

#include <atomic>

#include "kernels.h"

namespace pngpd {

namespace {
constexpr int MAX_DEV = 64;
std::atomic<int> g_cu_count[MAX_DEV];

struct AttrSlot {
    std::atomic<const void*> kern{nullptr};
    std::atomic<unsigned long long> devices{0};     // bit d: the attribute is set on device d
};
constexpr int MAX_KERNELS = 64;
AttrSlot g_attr[MAX_KERNELS];
}  // namespace

hipError_t device_cu_count(int* n_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
    int n = g_cu_count[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (n < 1) return hipErrorInvalidDevice;
        g_cu_count[dev].store(n, std::memory_order_relaxed);
    }
    *n_cu = n;
    return hipSuccess;
}

hipError_t set_max_lds_once(const void* kern, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
    AttrSlot* slot = nullptr;
    for (int i = 0; i < MAX_KERNELS && !slot; ++i) {
        const void* k = g_attr[i].kern.load(std::memory_order_acquire);
        if (k == kern) {
            slot = &g_attr[i];
        } else if (k == nullptr) {
            const void* expected = nullptr;
            if (g_attr[i].kern.compare_exchange_strong(expected, kern, std::memory_order_acq_rel) || expected == kern)
                slot = &g_attr[i];
        }
    }
    if (slot && ((slot->devices.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (slot) slot->devices.fetch_or(1ull << dev, std::memory_order_release);     // no slot left: set it every time
    return hipSuccess;
}

}  // namespace pngpd
