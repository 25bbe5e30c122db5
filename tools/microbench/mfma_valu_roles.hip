// Microbenchmark: wave w and wave w + 4 of a 512-thread workgroup share a SIMD.  One of them issues only bf16 MFMAs (4 chains),
// the other only independent full-rate VALU (8 chains of v_fma_f32): do the two pipes run side by side, and does it matter
// which wave is the older one?      hipcc -O3 --offload-arch=gfx950 -o _roles mfma_valu_roles.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// role: 0 idle, 1 MFMA stream, 2 VALU stream, 3 VALU stream with every 4th instruction a v_exp_f32
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int role_lo, int role_hi, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 1e-3f + i * 1e-2f;
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int role = wave < 4 ? role_lo : role_hi;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 24; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j & 3], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 96; ++j) x[j & 7] = __builtin_fmaf(x[j & 7], 0.999f, 1e-3f);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 96; ++j)
                x[j & 7] = (j & 3) == 3 ? __builtin_amdgcn_exp2f(x[j & 7]) : __builtin_fmaf(x[j & 7], 0.999f, 1e-3f);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    static unsigned long long h[256 * 8];
    const int iters = 2000;
    const char* names[] = {"idle", "MFMA", "VALU", "VALU+exp"};
    const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {2, 1}, {1, 3}, {3, 1}};
    for (auto& cs : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, cs[0], cs[1], iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m0 = 0, m4 = 0;
        for (int b = 0; b < 256; ++b) { m0 += h[b * 8 + 0]; m4 += h[b * 8 + 4]; }
        m0 /= 256.0 * iters; m4 /= 256.0 * iters;
        printf("waves 0-3 %-8s waves 4-7 %-8s: wave 0 %7.1f clk per step (24 MFMA | 96 VALU), wave 4 %7.1f\n", names[cs[0]], names[cs[1]], m0, m4);
    }
    return 0;
}
