// Which CU does a workgroup land on?  One record per workgroup: (XCC id, SE id, SH id, CU id) from the hardware id registers.
// Used by cu_mask_probe.py to see what a hipExtStreamCreateWithCUMask mask really selects on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GETREG(id, offset, size) __builtin_amdgcn_s_getreg((id) | ((offset) << 6) | (((size) - 1) << 11))

extern "C" __global__ void cu_id_kernel(uint32_t* out, int spin) {
    const uint32_t hw = GETREG(4, 0, 32);       // HW_REG_HW_ID
    const uint32_t xcc = GETREG(20, 0, 4);      // HW_REG_XCC_ID
    // keep the workgroup resident for a while so that a grid of one workgroup per CU really spreads over the CUs
    uint64_t t0 = __builtin_readcyclecounter();
    while ((int64_t)(__builtin_readcyclecounter() - t0) < (int64_t)spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

extern "C" int cu_id_launch(uint32_t* out, int blocks, int threads, int lds_bytes, int spin, void* stream) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(cu_id_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(cu_id_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, out, spin);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
