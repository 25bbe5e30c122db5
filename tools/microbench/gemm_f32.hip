// Experiment: LDS-tiled fp32 MFMA GEMM  C[M,N] = act(A[M,K] W[N,K]^T + bias)  for the tall, small-K
// shapes of Video-Swin (M = 115200 tokens, K = 96..384).  Built stand-alone by tools/gemm_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int PLANE = BM * 4 + 4;     // floats per kq plane (+4: shifts banks between the planes)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int ACT, int KP>   // ACT: 0 none, 1 relu, 2 gelu(erf);  KP: 4-wide k planes per step (BK = 4*KP)
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         int M, int N, int K) {
    constexpr int BK = 4 * KP;
    constexpr int RPP = 256 / KP;              // rows covered per load pass
    constexpr int NP = BM / RPP;               // load passes per operand
    __shared__ __attribute__((aligned(16))) float lds[2][2][KP * PLANE];   // [buf][A|W][k plane][row][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 15, kq = lane >> 4;
    const int lrow = tid / KP, lkq = tid % KP;
    const float* ap[NP];
    const float* wp[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int ra = min(m0 + lrow + RPP * i, M - 1), rw = min(n0 + lrow + RPP * i, N - 1);
        ap[i] = A + (long)ra * K + 4 * lkq;
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

template <int KP>
static int launch(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, hipStream_t st) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    if (act == 0) hipLaunchKernelGGL((gemm_nt_kernel<0, KP>), grid, dim3(256), 0, st, A, W, bias, C, M, N, K);
    else if (act == 1) hipLaunchKernelGGL((gemm_nt_kernel<1, KP>), grid, dim3(256), 0, st, A, W, bias, C, M, N, K);
    else hipLaunchKernelGGL((gemm_nt_kernel<2, KP>), grid, dim3(256), 0, st, A, W, bias, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int probe_gemm_nt(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act,
                             int bk, void* stream) {
    if (N % 4 != 0 || K % bk != 0) return -2;
    if (bk == 16) return launch<4>(A, W, bias, C, M, N, K, act, (hipStream_t)stream);
    if (bk == 32) return launch<8>(A, W, bias, C, M, N, K, act, (hipStream_t)stream);
    return -2;
}
