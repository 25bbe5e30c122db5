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
