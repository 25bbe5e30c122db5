// Microbenchmark 3: build K14's inner loop up from the 97 % MFMA stream of mfma_lds_feed2 and see which piece costs what.
//   -DSTEP=0  GEMM2 stream only (16 acc tiles)                        -DSTEP=1  + GEMM1 stream (x fragments in 64 VGPRs) + ReLU
//   -DSTEP=2  + 8 global loads per unit, committed to LDS + barrier   -DSTEP=3  + per-unit bias reads from LDS and loop bookkeeping
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(C, A, B) C = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)
#ifndef STEP
#define STEP 0
#endif

__global__ __launch_bounds__(512, 2) void k(float* out, const float* __restrict__ gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float4 img[];   // 2 x 4096 float4
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef RANDOM_DATA
    for (int i = tid; i < 8192; i += 512) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (blockIdx.x * 40503u);
        img[i] = make_float4(__uint_as_float(0x3f000000u | (h & 0x7fffffu)) - 0.75f, __uint_as_float(0x3f000000u | ((h * 31u) & 0x7fffffu)) - 0.75f,
                             __uint_as_float(0x3f000000u | ((h * 131u) & 0x7fffffu)) - 0.75f, __uint_as_float(0x3f000000u | ((h * 977u) & 0x7fffffu)) - 0.75f);
    }
#else
    for (int i = tid; i < 8192; i += 512) img[i] = make_float4(i * 1e-4f, 1.f, 2.f, 3.f);
#endif
    __syncthreads();
    float4 xf[16];
#ifdef RANDOM_DATA
    for (int j = 0; j < 16; ++j) {
        unsigned h = (unsigned)((tid * 16 + j) * 2246822519u);
        xf[j] = make_float4(__uint_as_float(0x3f000000u | (h & 0x7fffffu)) - 0.75f, __uint_as_float(0x3f000000u | ((h * 7u) & 0x7fffffu)) - 0.75f,
                            __uint_as_float(0x3f000000u | ((h * 13u) & 0x7fffffu)) - 0.75f, __uint_as_float(0x3f000000u | ((h * 29u) & 0x7fffffu)) - 0.75f);
    }
#else
    for (int j = 0; j < 16; ++j) xf[j] = make_float4(lane * 1e-3f + j, 1.f, 0.5f, 0.25f);
#endif
    f32x4 o[16];
    for (int i = 0; i < 16; ++i) o[i] = (f32x4){0, 0, 0, 0};
    f32x4 ha = {0.5f, 1.5f, 2.5f, 3.5f}, hb = {0.25f, 1.25f, 2.25f, 3.25f};
    float4 s0, s1, s2, s3, s4, s5, s6, s7;
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#if STEP >= 2
        {   const float4* gp = reinterpret_cast<const float4*>(gsrc) + (long)((it * 37 + blockIdx.x) & 255) * 4096 + tid;
            s0 = gp[0]; s1 = gp[512]; s2 = gp[1024]; s3 = gp[1536]; s4 = gp[2048]; s5 = gp[2560]; s6 = gp[3072]; s7 = gp[3584]; }
#endif
        const float4* W1i = img + buf * 4096 + lane;
        const float4* W2i = W1i + 2048;
#if STEP >= 1
#if STEP >= 3
        const float4 ba = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(img) + ((it * 32) & 2047) + 4 * (lane >> 4));
        ha = (f32x4){ba.x, ba.y, ba.z, ba.w}; hb = (f32x4){ba.y, ba.x, ba.w, ba.z};
#else
        ha = (f32x4){0.5f, 1.5f, 2.5f, 3.5f}; hb = (f32x4){0.25f, 1.25f, 2.25f, 3.25f};
#endif
#define G1(WA, WB, J) MF(ha, WA.x, xf[J].x); MF(hb, WB.x, xf[J].x); MF(ha, WA.y, xf[J].y); MF(hb, WB.y, xf[J].y); \
                      MF(ha, WA.z, xf[J].z); MF(hb, WB.z, xf[J].z); MF(ha, WA.w, xf[J].w); MF(hb, WB.w, xf[J].w);
        {
            float4 a0 = W1i[0], b0 = W1i[1024], a1 = W1i[64], b1 = W1i[1024 + 64], c0, d0, c1, d1;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                c0 = W1i[(j + 2) * 64]; d0 = W1i[1024 + (j + 2) * 64]; c1 = W1i[(j + 3) * 64]; d1 = W1i[1024 + (j + 3) * 64];
                G1(a0, b0, j) G1(a1, b1, j + 1)
                __builtin_amdgcn_sched_barrier(0);
                if (j + 4 < 16) { a0 = W1i[(j + 4) * 64]; b0 = W1i[1024 + (j + 4) * 64]; a1 = W1i[(j + 5) * 64]; b1 = W1i[1024 + (j + 5) * 64]; }
                G1(c0, d0, j + 2) G1(c1, d1, j + 3)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 4; ++i) { ha[i] = fmaxf(ha[i], 0.f); hb[i] = fmaxf(hb[i], 0.f); }
#endif
#define G2(V0, V1, V2, V3, A, B)                                                                                  \
    MF(o[A], V0.x, ha[0]); MF(o[B], V1.x, ha[0]); MF(o[A], V0.y, ha[1]); MF(o[B], V1.y, ha[1]);                   \
    MF(o[A], V0.z, ha[2]); MF(o[B], V1.z, ha[2]); MF(o[A], V0.w, ha[3]); MF(o[B], V1.w, ha[3]);                   \
    MF(o[A], V2.x, hb[0]); MF(o[B], V3.x, hb[0]); MF(o[A], V2.y, hb[1]); MF(o[B], V3.y, hb[1]);                   \
    MF(o[A], V2.z, hb[2]); MF(o[B], V3.z, hb[2]); MF(o[A], V2.w, hb[3]); MF(o[B], V3.w, hb[3]);
        {
            float4 p0 = W2i[0], p1 = W2i[64], p2 = W2i[1024], p3 = W2i[1024 + 64], r0, r1, r2, r3;
#pragma unroll
            for (int ct = 0; ct < 16; ct += 4) {
                r0 = W2i[(ct + 2) * 64]; r1 = W2i[(ct + 3) * 64]; r2 = W2i[1024 + (ct + 2) * 64]; r3 = W2i[1024 + (ct + 3) * 64];
                G2(p0, p1, p2, p3, ct, ct + 1)
                __builtin_amdgcn_sched_barrier(0);
                if (ct + 4 < 16) { p0 = W2i[(ct + 4) * 64]; p1 = W2i[(ct + 5) * 64]; p2 = W2i[1024 + (ct + 4) * 64]; p3 = W2i[1024 + (ct + 5) * 64]; }
                G2(r0, r1, r2, r3, ct + 2, ct + 3)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if STEP >= 2
        {   float4* dst = img + (buf ^ 1) * 4096 + tid;
            dst[0] = s0; dst[512] = s1; dst[1024] = s2; dst[1536] = s3; dst[2048] = s4; dst[2560] = s5; dst[3072] = s6; dst[3584] = s7; }
        __syncthreads();
        buf ^= 1;
#endif
    }
    float s = ha[0] + hb[1];
    for (int i = 0; i < 16; ++i) s += o[i][0] + o[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float *out, *gsrc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&gsrc, 256L * 4096 * 16);
#ifdef RANDOM_DATA
    {   float* h = (float*)malloc(256L * 4096 * 16);
        unsigned st = 12345u;
        for (long i = 0; i < 256L * 4096 * 4; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.5f; }
        hipMemcpy(gsrc, h, 256L * 4096 * 16, hipMemcpyHostToDevice); free(h); }
#else
    hipMemset(gsrc, 0, 256L * 4096 * 16);
#endif
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, out, gsrc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = STEP >= 1 ? 256.0 : 128.0;
    printf("STEP %d: %.3f ms = %.1f TFLOP/s\n", STEP, ms, 256.0 * 8 * iters * mf * 2048.0 / (ms * 1e-3) / 1e12);
    return 0;
}
