// Microbenchmark: how do bf16 MFMAs (v_mfma_f32_16x16x32_bf16) and vector ALU work overlap on a gfx950 SIMD?
//   mode 1: waves 0-3 issue MFMAs only (4 independent accumulators), waves 4-7 idle
//   mode 2: waves 4-7 issue VALU only (full-rate v_fma_f32 / v_pk / cvt mix like the three-way split), waves 0-3 idle
//   mode 3: both (wave w and w + 4 share a SIMD): co-execution ACROSS waves
//   mode 4: every wave interleaves 1 MFMA with NV independent VALU instructions: co-execution INSIDE a wave (1 wave per SIMD)
//   mode 5: the same with 2 waves per SIMD
//   mode 6: waves 0-3 MFMA only but each MFMA DEPENDS on the previous one (one accumulator)
// Prints cycles per MFMA / per VALU instruction for each mode.   hipcc -O3 --offload-arch=gfx950 -o /tmp/ov mfma_bf16_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__device__ __forceinline__ void valu_block(float& x0, float& x1, float& x2, float& x3) {
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(x3));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(x0));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(x2));
    }
}

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int mode, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
    float x0 = 1.0f + lane * 1e-3f, x1 = 0.999f, x2 = 1.001f, x3 = 0.998f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool mf = (mode == 1 || mode == 3) && wave < 4;
    const bool va = (mode == 2 || mode == 3) && wave >= 4;
    if (mf) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            }
        }
    } else if (va) {
        for (int i = 0; i < iters; ++i) valu_block<128>(x0, x1, x2, x3);
    } else if (mode == 6 && wave < 4) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        }
    } else if ((mode == 4 && wave < 4) || mode == 5 || (mode >= 10 && (mode < 20 ? wave < 4 : true))) {
        const int nv = mode >= 10 ? mode % 10 : 4;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#define STEP(C)                                                                  \
                C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, C, 0, 0, 0);   \
                if (nv == 2) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(x3)); } \
                else if (nv == 3) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(x3)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(x0)); } \
                else if (nv == 6) { valu_block<4>(x0, x1, x2, x3); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(x3)); } \
                else if (nv == 8) { valu_block<8>(x0, x1, x2, x3); }            \
                else valu_block<4>(x0, x1, x2, x3);
                STEP(c0) STEP(c1) STEP(c2) STEP(c3)
#undef STEP
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    static unsigned long long h[256 * 8];
    const int iters = 2000;
    const int modes[] = {1, 2, 3, 6, 4, 5, 12, 13, 16, 18, 22, 23, 26, 28};
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, mode, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; v += h[b * 8 + 4]; }
        m /= 256; v /= 256;
        const int nv = mode >= 10 ? mode % 10 : 4;
        if (mode <= 3 || mode == 6)
            printf("mode %2d: MFMA wave %8.0f clk (%.2f per MFMA), VALU wave %8.0f clk (%.2f per VALU instr)\n", mode, m,
                   m / (iters * 32.0), v, v / (iters * 128.0));
        else
            printf("mode %2d: %d wave(s)/SIMD, 1 MFMA + %d VALU interleaved: wave0 %8.0f clk = %.2f per (MFMA + %d VALU); wave4 %8.0f\n",
                   mode, (mode == 5 || mode >= 20) ? 2 : 1, nv, m, m / (iters * 32.0), nv, v);
    }
    return 0;
}
