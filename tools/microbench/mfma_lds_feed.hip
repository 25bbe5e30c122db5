// Microbenchmark: what does feeding v_mfma_f32_16x16x4_f32 from LDS cost, with one or two waves per SIMD and with the
// accumulators in VGPRs or AGPRs?  1024-thread workgroups, one per CU; waves w, w + 4, w + 8, w + 12 share a SIMD.
//   mode 0  operands in registers (the 32-cycle floor)
//   mode 1  A operand = ds_read_b128 per 4 MFMAs, ping-pong register sets (what K1 / K13 do)
//   mode 2  as 1, accumulators in AGPRs (inline asm)
//   mode 3  as 1, reads as 2 x ds_read_b64
// waves: 4 / 8 / 12 / 16 = one .. four per SIMD (1024-thread workgroups).   build: hipcc -O3 --offload-arch=gfx950 mfma_lds_feed.hip -o mfma_lds_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(C, A, B) C = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)
#define MFA(C, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B))

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, int waves) {
    __shared__ __attribute__((aligned(16))) float4 img[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 1024) img[i] = make_float4(i * 1e-4f, 1.f, 2.f, 3.f);
    __syncthreads();
    if (wave >= waves) return;
    float b = 0.5f + lane * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float4* p = img + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        float a = 1.0f + lane * 1e-3f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { MF(c0, a, b); MF(c1, a, b); MF(c2, a, b); MF(c3, a, b); }
        }
    } else {
        float4 A0 = p[0], A1 = p[64], B0, B1;
        for (int i = 0; i < iters; ++i) {
            const float4* q = p + ((i & 7) * 512);
            // 32 MFMAs per iteration, 8 fragment reads (1 per 4 MFMAs), ping-pong A / B sets
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (MODE == 3) {
                    const float2* q2 = reinterpret_cast<const float2*>(q + (4 * h + 2) * 64);
                    float2 u0 = q2[0], u1 = q2[1], u2 = q2[128], u3 = q2[129];
                    B0 = make_float4(u0.x, u0.y, u1.x, u1.y); B1 = make_float4(u2.x, u2.y, u3.x, u3.y);
                } else { B0 = q[(4 * h + 2) * 64]; B1 = q[(4 * h + 3) * 64]; }
                if (MODE == 2) { MFA(c0, A0.x, b); MFA(c1, A1.x, b); MFA(c2, A0.y, b); MFA(c3, A1.y, b);
                                 MFA(c0, A0.z, b); MFA(c1, A1.z, b); MFA(c2, A0.w, b); MFA(c3, A1.w, b); }
                else { MF(c0, A0.x, b); MF(c1, A1.x, b); MF(c2, A0.y, b); MF(c3, A1.y, b);
                       MF(c0, A0.z, b); MF(c1, A1.z, b); MF(c2, A0.w, b); MF(c3, A1.w, b); }
                __builtin_amdgcn_sched_barrier(0);
                A0 = q[(4 * h + 4) * 64]; A1 = q[(4 * h + 5) * 64];
                if (MODE == 2) { MFA(c0, B0.x, b); MFA(c1, B1.x, b); MFA(c2, B0.y, b); MFA(c3, B1.y, b);
                                 MFA(c0, B0.z, b); MFA(c1, B1.z, b); MFA(c2, B0.w, b); MFA(c3, B1.w, b); }
                else { MF(c0, B0.x, b); MF(c1, B1.x, b); MF(c2, B0.y, b); MF(c3, B1.y, b);
                       MF(c0, B0.z, b); MF(c1, B1.z, b); MF(c2, B0.w, b); MF(c3, B1.w, b); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    unsigned long long h[256 * 16];
    const int iters = 4000;
    for (int waves = 4; waves <= 16; waves += 4)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, cyc, iters, waves);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, out, cyc, iters, waves);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, out, cyc, iters, waves);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, out, cyc, iters, waves);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0;
            for (int b = 0; b < 256; ++b) m += h[b * 16 + 0];
            const double per_wave = m / 256 / (iters * 32.0);
            const double tflops = 256.0 * waves * iters * 32.0 * 2048.0 / (ms * 1e-3) / 1e12;
            printf("waves/SIMD %d mode %d: %.1f cycles per MFMA per wave => %.1f cycles of SIMD time per MFMA; wall %.3f ms = %.1f TFLOP/s\n",
                   waves / 4, mode, per_wave, per_wave / (waves / 4), ms, tflops);
        }
    return 0;
}
