// Microbenchmark: a wave that has 24 bf16 MFMAs (4 chains of 6) and NV independent full-rate VALU instructions per step --
// issued as two phases (all MFMAs, then all VALU: K1's shape) or interleaved (1 MFMA, NV / 24 VALU) -- with 1 or 2 waves per SIMD.
// Ideal per step = max(24 x 16, NV x 4) cycles per SIMD.    hipcc -O3 --offload-arch=gfx950 -o _mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool MIX>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int waves, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 1e-3f + i * 1e-2f;
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 24; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) x[j & 7] = __builtin_fmaf(x[j & 7], 0.999f, 1e-3f);
            if (MIX) {
#pragma unroll
                for (int j = 0; j < 24; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV / 24, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NV, bool MIX>
void run(float* out, unsigned long long* cyc, int waves) {
    static unsigned long long h[256 * 8];
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NV, MIX>), dim3(256), dim3(512), 0, 0, out, cyc, waves, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, m4 = 0;
    for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; m4 += h[b * 8 + 4]; }
    m /= 256; m4 /= 256;
    printf("  [wave 4: %.1f] ", m4 / iters);
    printf("NV %3d %s %d wave(s)/SIMD: %7.1f clk per step per wave (ideal per SIMD %d)\n", NV, MIX ? "interleaved" : "two phases ",
           waves / 4, m / iters, (24 * 16 > NV * 4 ? 24 * 16 : NV * 4));
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0, false>(out, cyc, 1); run<0, false>(out, cyc, 4); run<0, false>(out, cyc, 8);
    run<48, false>(out, cyc, 4); run<48, true>(out, cyc, 4); run<48, false>(out, cyc, 8); run<48, true>(out, cyc, 8);
    run<72, false>(out, cyc, 4); run<72, true>(out, cyc, 4); run<72, false>(out, cyc, 8); run<72, true>(out, cyc, 8);
    run<96, false>(out, cyc, 4); run<96, true>(out, cyc, 4); run<96, false>(out, cyc, 8); run<96, true>(out, cyc, 8);
    return 0;
}
