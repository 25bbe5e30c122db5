// K12: LDS-tiled fp32 MFMA GEMM with a fused epilogue for gfx950 -- out = act(x W^T + bias), used where
// fusing the activation beats the library GEMM + a separate elementwise pass: the Video-Swin MLP
// up-projection fc1 + exact (erf) GELU on the tall, small-K shapes of stages 0-1
// (reference models/video_swin_transformer.py:24-37 Mlp.forward: fc1 -> nn.GELU -> fc2).  hipBLASLt's
// GELU epilogue is the tanh approximation, so with the library the GELU is its own 354 MB read+write pass
// at stage 0; here it is applied to the accumulators.
//
// 128 x 128 output tile per 256-thread workgroup (2 x 2 waves, 64 x 64 each = 16 MFMA accumulators), K in
// steps of 16.  Both operands are K-contiguous, so a lane's 16-B load is 4 consecutive k of one row:
// exactly the per-lane operand of four v_mfma_f32_16x16x4_f32.  LDS holds the tile as [k plane][row]
// float4 (plane stride padded by 16 B: conflict-free b128 writes and reads), double-buffered, one
// barrier per K-step, next tile's global loads in flight during the MFMAs (branch-free: the last step
// re-loads its own tile).  The product is formed as W x^T so that a lane ends up with 4 consecutive
// output columns of one row -> 16-B stores.  Measured on MI355X (tools/gemm_probe.py): 115200 x 96 x 384
// 89.6 us vs 96.6 us library, 114 us vs 160 us with GELU; the kernel is not used where the library
// wins (large K, or grids that do not fill the chip with 128 x 128 tiles).
#include "soc_common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int PLANE = BM * 4 + 4;     // floats per kq plane (+4: shifts banks between the planes)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// up to four layers that read the same input: segment i owns the column tiles [tile0[i], tile0[i+1])
constexpr int MAX_SEG = 4;
struct Seg2 {
    const float* w[MAX_SEG];
    const float* bias[MAX_SEG];
    float* out[MAX_SEG];
    int N[MAX_SEG];
    int tile0[MAX_SEG];
    int nseg;
};

// ACT: 0 none, 1 relu, 2 gelu(erf);  KP: 4-wide k planes per step (BK = 4*KP);  HAS_ADD: A = x + x_add
template <int ACT, int KP, bool HAS_ADD>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ ADD,
                                                         const Seg2 sg, int M, int K) {
    int seg = 0;
#pragma unroll
    for (int i = 1; i < MAX_SEG; ++i)
        if (i < sg.nseg && (int)blockIdx.x >= sg.tile0[i]) seg = i;
    const float* __restrict__ W = sg.w[seg];
    const float* __restrict__ bias = sg.bias[seg];
    float* __restrict__ C = sg.out[seg];
    const int N = sg.N[seg];
    constexpr int BK = 4 * KP;
    constexpr int RPP = 256 / KP;              // rows covered per load pass
    constexpr int NP = BM / RPP;               // load passes per operand
    __shared__ __attribute__((aligned(16))) float lds[2][2][KP * PLANE];   // [buf][A|W][k plane][row][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = ((int)blockIdx.x - sg.tile0[seg]) * BN;
    const int r = lane & 15, kq = lane >> 4;
    const int lrow = tid / KP, lkq = tid % KP;
    const float* ap[NP];
    const float* pp[NP];
    const float* wp[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int ra = min(m0 + lrow + RPP * i, M - 1), rw = min(n0 + lrow + RPP * i, N - 1);
        ap[i] = A + (long)ra * K + 4 * lkq;
        pp[i] = HAS_ADD ? ADD + (long)ra * K + 4 * lkq : nullptr;
        wp[i] = W + (long)rw * K + 4 * lkq;
    }
    float4 ga[NP], gw[NP];
    f32x4 acc[4][4];   // [n block][m block]: acc = mfma(w_frag, a_frag): lane holds C[m = r][n = 4*kq + reg]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        ga[i] = *reinterpret_cast<const float4*>(ap[i]);
        if (HAS_ADD) {
            const float4 t = *reinterpret_cast<const float4*>(pp[i]);
            ga[i].x += t.x; ga[i].y += t.y; ga[i].z += t.z; ga[i].w += t.w;
        }
        gw[i] = *reinterpret_cast<const float4*>(wp[i]);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        *reinterpret_cast<float4*>(&lds[0][0][lkq * PLANE + (lrow + RPP * i) * 4]) = ga[i];
        *reinterpret_cast<float4*>(&lds[0][1][lkq * PLANE + (lrow + RPP * i) * 4]) = gw[i];
    }
    __syncthreads();
    const int steps = K / BK;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const int knext = min((s + 1) * BK, K - BK);     // last step re-loads its own tile: no branch
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            ga[i] = *reinterpret_cast<const float4*>(ap[i] + knext);
            if (HAS_ADD) {
                const float4 t = *reinterpret_cast<const float4*>(pp[i] + knext);
                ga[i].x += t.x; ga[i].y += t.y; ga[i].z += t.z; ga[i].w += t.w;
            }
            gw[i] = *reinterpret_cast<const float4*>(wp[i] + knext);
        }
#pragma unroll
        for (int h = 0; h < KP / 4; ++h) {
            float4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const float4*>(&lds[buf][0][(4 * h + kq) * PLANE + (wm * 64 + 16 * i + r) * 4]);
                wf[i] = *reinterpret_cast<const float4*>(&lds[buf][1][(4 * h + kq) * PLANE + (wn * 64 + 16 * i + r) * 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].x, af[i].x, acc[j][i], 0, 0, 0);
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].y, af[i].y, acc[j][i], 0, 0, 0);
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].z, af[i].z, acc[j][i], 0, 0, 0);
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].w, af[i].w, acc[j][i], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            *reinterpret_cast<float4*>(&lds[buf ^ 1][0][lkq * PLANE + (lrow + RPP * i) * 4]) = ga[i];
            *reinterpret_cast<float4*>(&lds[buf ^ 1][1][lkq * PLANE + (lrow + RPP * i) * 4]) = gw[i];
        }
        __syncthreads();
    }
    // epilogue: lane holds C[m][n..n+3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * kq;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && n < N) b4 = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + 16 * i + r;
            if (m < M && n < N) {
                float4 v = make_float4(acc[j][i][0] + b4.x, acc[j][i][1] + b4.y, acc[j][i][2] + b4.z, acc[j][i][3] + b4.w);
                if (ACT == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (ACT == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
                *reinterpret_cast<float4*>(C + (long)m * N + n) = v;
            }
        }
    }
}


template <bool HAS_ADD>
int launch(const float* x, const float* x_add, const Seg2& sg, int tiles, int M, int K, int act, hipStream_t st) {
    dim3 grid(tiles, soc_ceil_div(M, BM));
    if (act == 0) hipLaunchKernelGGL((gemm_nt_kernel<0, 4, HAS_ADD>), grid, dim3(256), 0, st, x, x_add, sg, M, K);
    else if (act == 1) hipLaunchKernelGGL((gemm_nt_kernel<1, 4, HAS_ADD>), grid, dim3(256), 0, st, x, x_add, sg, M, K);
    else hipLaunchKernelGGL((gemm_nt_kernel<2, 4, HAS_ADD>), grid, dim3(256), 0, st, x, x_add, sg, M, K);
    return soc_check_launch();
}

}  // namespace

extern "C" int soc_linear_act_multi_f32(const float* x, const float* x_add, int nseg, const float* const* w,
                                        const float* const* bias, float* const* out, const int* N, int M, int K,
                                        int act, void* stream) {
    if (M < 0 || K <= 0 || act < 0 || act > 2 || nseg < 1 || !w || !out || !N) return SOC_EINVAL;
    if (nseg > MAX_SEG) return SOC_EUNSUPPORTED;
    if (M == 0) return SOC_OK;
    if (!x) return SOC_EINVAL;
    if (K % 16 != 0) return SOC_EUNSUPPORTED;
    Seg2 sg;
    int tiles = 0;
    for (int i = 0; i < MAX_SEG; ++i) {
        const int j = i < nseg ? i : 0;
        if (!w[j] || !out[j] || N[j] <= 0) return SOC_EINVAL;
        if (N[j] % 4 != 0) return SOC_EUNSUPPORTED;
        const float* b = bias ? bias[j] : nullptr;
        if ((((uintptr_t)w[j] | (uintptr_t)out[j] | (uintptr_t)b) & 15) != 0) return SOC_EUNSUPPORTED;
        sg.w[i] = w[j]; sg.bias[i] = b; sg.out[i] = out[j]; sg.N[i] = N[j];
        sg.tile0[i] = tiles;
        if (i < nseg) tiles += soc_ceil_div(N[j], BN);
    }
    sg.nseg = nseg;
    if ((((uintptr_t)x | (uintptr_t)x_add) & 15) != 0) return SOC_EUNSUPPORTED;
    if (x_add) return launch<true>(x, x_add, sg, tiles, M, K, act, (hipStream_t)stream);
    return launch<false>(x, x_add, sg, tiles, M, K, act, (hipStream_t)stream);
}

extern "C" int soc_linear_act_f32(const float* x, const float* w, const float* bias, float* out, int M, int N, int K,
                                  int act, void* stream) {
    if (N <= 0) return SOC_EINVAL;
    return soc_linear_act_multi_f32(x, nullptr, 1, &w, &bias, &out, &N, M, K, act, stream);
}
