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

// CUs a launch on `st` can use: the device's count, cut down to the stream's CU mask when the caller created the stream with
// hipExtStreamCreateWithCUMask (tools/experiments/cu_mask_probe.py, partition_probe.py: a head on one CU set, a tail on another --
// measured slower at every split and not shipped, DESIGN.md section 3 "Launch structure"; the query stays so that a caller who
// does partition the chip gets grids of the right size).  A one-workgroup-per-CU grid sized for the whole chip on a stream that
// owns 240 CUs runs in two rounds (measured: K23 320 -> 513 us), so every persistent kernel sizes itself from here.  The mask is a
// property of the stream argument -- nothing the library remembers; the words queried follow the device's CU count (ADVICE r5:
// a fixed 8 words sent devices with more than 256 CUs down the error path on every launch).
int soc_num_cus(hipStream_t st) {
    const int n = soc_device_cus();
    constexpr int MAXW = 32;                                   // up to 1024 CUs
    const int words = (n + 31) / 32 < MAXW ? (n + 31) / 32 : MAXW;
    uint32_t mask[MAXW] = {0};
    if (hipExtStreamGetCUMask(st, (uint32_t)words, mask) != hipSuccess) {
        (void)hipGetLastError();               // not a launch error: leave nothing behind for soc_check_launch()
        return n;
    }
    int c = 0;
    for (int i = 0; i < words; ++i) c += __builtin_popcount(mask[i]);
    return c > 0 && c < n ? c : n;
}

int soc_device_cus() {
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

extern "C" int soc_stream_cus(void* stream) { return soc_num_cus((hipStream_t)stream); }
