// Shared helpers for the gfx950 kernels of libsoc_hip.so (see include/soc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "soc_hip.h"

#define SOC_WAVE 64

static inline int soc_check_launch() {
    return hipGetLastError() == hipSuccess ? SOC_OK : SOC_ELAUNCH;
}

static inline int soc_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Per-device one-time state (a process may drive several GPUs: clip-parallel ranks are one process per GPU,
// but nothing in the C ABI forbids a host that switches devices).  Kernel attributes such as
// hipFuncAttributeMaxDynamicSharedMemorySize are per device, so "done once" must be keyed by the current device.
#define SOC_MAX_DEVICES 64
static inline int soc_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SOC_MAX_DEVICES) return -1;
    return dev;
}

// defined in soc_capi.hip: the CU count persistent kernels size their grids for (of the current device)
#define SOC_CU_MASK_WORDS 8             // 256 CU bits
int soc_device_cus();
int soc_num_cus(hipStream_t st);
