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
