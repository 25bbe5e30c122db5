// Microbenchmark: can a VALU-only wave and an f32-MFMA-only wave that share a SIMD overlap?
// 512-thread workgroups (waves w and w+4 share a SIMD).  mode bit0: waves 0-3 run MFMA, bit1: waves
// 4-7 run VALU (v_exp + v_fma mix like the K1 softmax).  Prints cycles per role alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int mode, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (mode & 2) {
            float x0 = a, x1 = a * 1.1f, x2 = a * 1.2f, x3 = a * 1.3f, s = 0.f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // 8 x (sub, exp, add, max): 32 VALU of which 8 transcendental
                    x0 = __builtin_amdgcn_exp2f(x0 - 1.0f); s += x0; x1 = fmaxf(x1, x0);
                    x2 = __builtin_amdgcn_exp2f(x2 - 1.0f); s += x2; x3 = fmaxf(x3, x2);
                    x0 = __builtin_amdgcn_exp2f(x1 - 2.0f); s += x0; x1 = fmaxf(x3, x0);
                    x2 = __builtin_amdgcn_exp2f(x3 - 2.0f); s += x2; x3 = fmaxf(x1, x2);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = s + x0 + x1 + x2 + x3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, mode, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; v += h[b * 8 + 4]; }
        printf("mode %d: MFMA wave %.0f cycles (%.1f per MFMA), VALU wave %.0f cycles (%.2f per VALU op)\n", mode,
               m / 256, m / 256 / (iters * 32.0), v / 256, v / 256 / (iters * 96.0));
    }
    return 0;
}
