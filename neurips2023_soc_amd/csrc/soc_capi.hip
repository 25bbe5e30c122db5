// ABI version + error strings of libsoc_hip.so.
#include "soc_common.h"

extern "C" int soc_hip_abi_version(void) { return SOC_HIP_ABI_VERSION; }

extern "C" const char* soc_hip_error_string(int code) {
    switch (code) {
        case SOC_OK: return "ok";
        case SOC_EINVAL: return "invalid argument (null pointer or non-positive dimension)";
        case SOC_EUNSUPPORTED: return "shape not supported by this kernel build";
        case SOC_ELAUNCH: return "HIP kernel launch failed";
        case SOC_EWORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

// CUs the persistent kernels (one workgroup per CU for a whole launch: K13 / K13b, K20, K23, K24) size their grids for: the
// device's count minus a reserve the host may ask for, so that the short launches of a side stream (the query chain of the
// previous clip, the text branch) find a free CU at once instead of waiting for a 100-300 us workgroup to retire.
#include <atomic>
static std::atomic<int> g_reserved_cus{0};

extern "C" void soc_set_reserved_cus(int n) { g_reserved_cus.store(n < 0 ? 0 : n, std::memory_order_relaxed); }
extern "C" int soc_get_reserved_cus(void) { return g_reserved_cus.load(std::memory_order_relaxed); }

int soc_num_cus() {
    static std::atomic<int> cached[SOC_MAX_DEVICES];      // 0 = not queried yet; per device
    const int dev = soc_current_device();
    int n = 256;
    if (dev >= 0) {
        n = cached[dev].load(std::memory_order_relaxed);
        if (n == 0) {
            n = 256;
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
            cached[dev].store(n, std::memory_order_relaxed);
        }
    }
    const int r = g_reserved_cus.load(std::memory_order_relaxed);
    return n - r >= n / 2 ? n - r : n / 2;                 // never less than half the chip
}
