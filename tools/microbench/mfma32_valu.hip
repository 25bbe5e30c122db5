// Microbenchmark (round 6): does vector work run beside v_mfma_f32_32x32x16_bf16 where it does not beside the 16x16x32 form?
// (tools/microbench/mfma_valu_roles.hip / mfma_valu_mix.hip found NO co-issue beside 16x16x32 streams.)
//   roles: waves 0-3 stream MFMAs, waves 4-7 (their SIMD partners) stream independent VALU;
//   mix:   every wave does NM MFMAs + NV vector instructions per step, in two phases or interleaved, 1 or 2 waves per SIMD;
//   the vector stream is plain v_fma (KIND 0) or a softmax-like mix exp / add / cvt_pk / sub (KIND 1).
// hipcc -O3 --offload-arch=gfx950 -o _m32 mfma32_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <bool BIG>
struct Acc;
template <>
struct Acc<true> {
    f32x16 c[2];
    __device__ void init() { for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f; }
#ifdef CHAIN_BLOCKS      // six dependent MFMAs on one accumulator, then six on the other (K1's phases) instead of alternating
    __device__ void mfma(int j, bf16x8 a, bf16x8 b) { c[(j / 6) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[(j / 6) & 1], 0, 0, 0); }
#else
    __device__ void mfma(int j, bf16x8 a, bf16x8 b) { c[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j & 1], 0, 0, 0); }
#endif
    __device__ float sum() { return c[0][0] + c[1][5]; }
};
template <>
struct Acc<false> {
    f32x4 c[4];
    __device__ void init() { for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) c[j][i] = 0.f; }
    __device__ void mfma(int j, bf16x8 a, bf16x8 b) { c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j & 3], 0, 0, 0); }
    __device__ float sum() { return c[0][0] + c[1][1] + c[2][2] + c[3][3]; }
};

template <int KIND>
__device__ __forceinline__ void vwork(float (&x)[8], unsigned (&pk)[4], int j) {
    if (KIND == 0) {
        x[j & 7] = __builtin_fmaf(x[j & 7], 0.999f, 1e-3f);
    } else {
        // softmax-like: per 6 instructions 1 exp, 1 add, 1.5 cvt_pk ... approximated by a rotating pattern
        const int r = j % 6;
        if (r == 0) x[j & 7] = __builtin_amdgcn_exp2f(x[j & 7]);
        else if (r == 1) x[(j + 1) & 7] += x[j & 7];
        else if (r == 2) { bf16x2 h = {(__bf16)x[j & 7], (__bf16)x[(j + 3) & 7]}; pk[j & 3] ^= __builtin_bit_cast(unsigned, h); }
        else if (r == 3) x[j & 7] = x[j & 7] - __builtin_bit_cast(float, pk[j & 3] << 16);
        else if (r == 4) x[j & 7] = x[j & 7] - __builtin_bit_cast(float, pk[j & 3] & 0xffff0000u);
        else x[j & 7] = __builtin_fmaf(x[j & 7], 0.999f, 1e-3f);
    }
}

// MODE 0: two phases (all MFMAs, then all vector work); 1: interleaved by sched_group_barrier; 2: roles (waves 0-3 MFMA only, 4-7 VALU only)
template <bool BIG, int NM, int NV, int MODE, int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int waves, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + lane * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
    float x[8];
    unsigned pk[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + lane * 1e-3f + i * 1e-2f;
    Acc<BIG> acc;
    acc.init();
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves) {
        if (MODE == 2) {
            if (wave < 4) {
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int j = 0; j < NM; ++j) acc.mfma(j, a, b);
                }
            } else {
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) vwork<KIND>(x, pk, j);
                }
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NM; ++j) acc.mfma(j, a, b);
#pragma unroll
                for (int j = 0; j < NV; ++j) vwork<KIND>(x, pk, j);
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < NM; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV / NM, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
                    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc.sum() + s + (float)(pk[0] ^ pk[1] ^ pk[2] ^ pk[3]);
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;

template <bool BIG, int NM, int NV, int MODE, int KIND>
void run(int waves) {
    static unsigned long long h[256 * 8];
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<BIG, NM, NV, MODE, KIND>), dim3(256), dim3(512), 0, 0, g_out, g_cyc, waves, iters);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, m4 = 0;
    for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; m4 += h[b * 8 + 4]; }
    m /= 256.0 * iters; m4 /= 256.0 * iters;
    const char* mode = MODE == 0 ? "two phases " : MODE == 1 ? "interleaved" : "roles      ";
    printf("%s NM %2d NV %3d %s %s %d wave(s)/SIMD: wave0 %7.1f  wave4 %7.1f clk per step  (MFMA alone %d)\n",
           BIG ? "32x32x16" : "16x16x32", NM, NV, KIND ? "softmax-mix" : "v_fma      ", mode, waves > 4 ? 2 : 1, m, m4,
           NM * (BIG ? 32 : 16));
}

int main() {
    (void)hipMalloc(&g_out, 256 * 512 * 4);
    (void)hipMalloc(&g_cyc, 256 * 8 * 8);
    // baselines: MFMA stream alone, 1 and 2 waves per SIMD
    run<true, 12, 0, 0, 0>(4); run<true, 12, 0, 0, 0>(8);
    run<false, 24, 0, 0, 0>(4); run<false, 24, 0, 0, 0>(8);
    // roles: partner wave streams VALU
    run<true, 12, 96, 2, 0>(8); run<false, 24, 96, 2, 0>(8);
    run<true, 12, 96, 2, 1>(8); run<false, 24, 96, 2, 1>(8);
    // in-wave: 1 wave per SIMD
    run<true, 12, 48, 0, 0>(4); run<true, 12, 48, 1, 0>(4);
    run<true, 12, 72, 0, 0>(4); run<true, 12, 72, 1, 0>(4);
    run<true, 12, 96, 0, 0>(4); run<true, 12, 96, 1, 0>(4);
    run<true, 12, 48, 1, 1>(4); run<true, 12, 72, 1, 1>(4); run<true, 12, 96, 1, 1>(4);
    run<false, 24, 48, 1, 0>(4); run<false, 24, 96, 1, 0>(4); run<false, 24, 96, 1, 1>(4);
    // in-wave: 2 waves per SIMD
    run<true, 12, 48, 0, 0>(8); run<true, 12, 48, 1, 0>(8);
    run<true, 12, 72, 0, 0>(8); run<true, 12, 72, 1, 0>(8);
    run<true, 12, 96, 0, 0>(8); run<true, 12, 96, 1, 0>(8);
    run<true, 12, 48, 1, 1>(8); run<true, 12, 72, 1, 1>(8); run<true, 12, 96, 1, 1>(8);
    run<true, 12, 96, 0, 1>(8);
    run<false, 24, 48, 1, 0>(8); run<false, 24, 96, 1, 0>(8); run<false, 24, 96, 1, 1>(8); run<false, 24, 96, 0, 1>(8);
    return 0;
}
