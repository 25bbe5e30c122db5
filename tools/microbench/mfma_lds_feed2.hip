// Microbenchmark 2: the second product of a fused FFN / the P.V of an attention tile as an MFMA stream --
// 16 accumulator tiles, B operand = 8 registers, A operand = 4 x ds_read_b128 per 16 MFMAs, two waves per SIMD.
//   mode 0: accumulators alternate in pairs (o[ct], o[ct+1]) x 8 MFMAs   (the K14 form)
//   mode 1: accumulators rotate over 4 tiles
//   mode 2: as 0 but B operand is ONE register for all MFMAs
//   mode 3: as 0 but only 4 accumulator tiles are used (reused for every pair)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(C, A, B) C = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float4 img[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) img[i] = make_float4(i * 1e-4f, 1.f, 2.f, 3.f);
    __syncthreads();
    float ha[4], hb[4];
    for (int i = 0; i < 4; ++i) { ha[i] = 0.5f + lane * 1e-3f + i; hb[i] = 0.25f + lane * 2e-3f + i; }
    f32x4 o[16];
    for (int i = 0; i < 16; ++i) o[i] = (f32x4){0, 0, 0, 0};
    const float4* W = img + lane;
    for (int it = 0; it < iters; ++it) {
        const float4* q = W + (it & 1) * 4096;
        float4 p0 = q[0], p1 = q[64], p2 = q[1024], p3 = q[1024 + 64];
        float4 r0, r1, r2, r3;
#define G2(V0, V1, V2, V3, A, B)                                                                                 \
    MF(o[A], V0.x, ha[0]); MF(o[B], V1.x, MODE == 2 ? ha[0] : ha[0]); MF(o[A], V0.y, MODE == 2 ? ha[0] : ha[1]); \
    MF(o[B], V1.y, MODE == 2 ? ha[0] : ha[1]); MF(o[A], V0.z, MODE == 2 ? ha[0] : ha[2]);                        \
    MF(o[B], V1.z, MODE == 2 ? ha[0] : ha[2]); MF(o[A], V0.w, MODE == 2 ? ha[0] : ha[3]);                        \
    MF(o[B], V1.w, MODE == 2 ? ha[0] : ha[3]); MF(o[A], V2.x, MODE == 2 ? ha[0] : hb[0]);                        \
    MF(o[B], V3.x, MODE == 2 ? ha[0] : hb[0]); MF(o[A], V2.y, MODE == 2 ? ha[0] : hb[1]);                        \
    MF(o[B], V3.y, MODE == 2 ? ha[0] : hb[1]); MF(o[A], V2.z, MODE == 2 ? ha[0] : hb[2]);                        \
    MF(o[B], V3.z, MODE == 2 ? ha[0] : hb[2]); MF(o[A], V2.w, MODE == 2 ? ha[0] : hb[3]);                        \
    MF(o[B], V3.w, MODE == 2 ? ha[0] : hb[3]);
#define G4(V0, V1, V2, V3, A)                                                                       \
    MF(o[A], V0.x, ha[0]); MF(o[A + 1], V1.x, ha[0]); MF(o[A + 2], V2.x, hb[0]); MF(o[A + 3], V3.x, hb[0]); \
    MF(o[A], V0.y, ha[1]); MF(o[A + 1], V1.y, ha[1]); MF(o[A + 2], V2.y, hb[1]); MF(o[A + 3], V3.y, hb[1]); \
    MF(o[A], V0.z, ha[2]); MF(o[A + 1], V1.z, ha[2]); MF(o[A + 2], V2.z, hb[2]); MF(o[A + 3], V3.z, hb[2]); \
    MF(o[A], V0.w, ha[3]); MF(o[A + 1], V1.w, ha[3]); MF(o[A + 2], V2.w, hb[3]); MF(o[A + 3], V3.w, hb[3]);
#pragma unroll
        for (int ct = 0; ct < 16; ct += 4) {
            r0 = q[(ct + 2) * 64]; r1 = q[(ct + 3) * 64]; r2 = q[1024 + (ct + 2) * 64]; r3 = q[1024 + (ct + 3) * 64];
            if (MODE == 1) { G4(p0, p1, p2, p3, ct) }
            else if (MODE == 3) { G2(p0, p1, p2, p3, 0, 1) }
            else { G2(p0, p1, p2, p3, ct, ct + 1) }
            __builtin_amdgcn_sched_barrier(0);
            if (ct + 4 < 16) { p0 = q[(ct + 4) * 64]; p1 = q[(ct + 5) * 64]; p2 = q[1024 + (ct + 4) * 64]; p3 = q[1024 + (ct + 5) * 64]; }
            if (MODE == 1) { G4(r0, r1, r2, r3, ct) }
            else if (MODE == 3) { G2(r0, r1, r2, r3, 2, 3) }
            else { G2(r0, r1, r2, r3, ct + 2, ct + 3) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += o[i][0] + o[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("mode %d: %.3f ms = %.1f TFLOP/s\n", mode, ms, 256.0 * 8 * iters * 128.0 * 2048.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
