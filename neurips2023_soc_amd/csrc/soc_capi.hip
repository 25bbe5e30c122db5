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
// current device's count (a cache of a device attribute -- no caller-visible state).  A host-settable reserve for the side
// stream's short launches was measured null in round 4 and left the ABI with ABI 16.
#include <atomic>

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
    return n;
}
