// K7: small-M linear layer for gfx950 -- out = act((x [+ x_add]) W^T + bias) with M <= a few hundred
// rows (the 160 frame queries / 20 video queries / ~10 words of SOC's decoder, VOC, heads and the
// text side of the VLA blocks).
//
// Why not the library GEMM: at M = 160, N = K = 256 the best hipBLASLt/rocBLAS kernel takes 12-16 us
// (macro-tiles of 32x256 give 5 workgroups on a 256-CU part; profiles/r01_bench_kernel_stats.csv),
// and ~90 such launches sit back to back on the decoder -> VOC critical path.  The work is 21 MFLOP:
// it is latency, not throughput, that matters.  Mapping:
//   * one workgroup = one 16x16 output tile, 4 waves = 4-way interleaved split of K;
//   * a lane (r = lane%16, kq = lane/16) loads 16 B of row r of x and of row r of W per K-step of 16,
//     which are exactly the A / B operands of four v_mfma_f32_16x16x4_f32 (the K order inside a step
//     is permuted identically on both sides, which a dot product does not see);
//   * all loads of a wave's share are issued before the first MFMA when K <= 512; nothing goes
//     through LDS except the final 4-partial reduction (4 KB);
//   * bias, the optional ReLU and the optional "x + x_add" (positional embedding added to the input,
//     rows of x_add broadcast as (m / add_div) % add_mod) are fused.
//   * up to 4 layers that read the same input (q / k / v projections, sampling offsets + attention
//     weights) run as column segments of ONE launch, each with its own weight, bias and output
//     tensor, and each either with or without the positional add -- every launch on this chain costs
//     ~4.4 us whatever it does.
// M*N/256 workgroups: 160 for the 256x256 case, 1280 for the FFN up-projection.
#include "soc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_SEG = 4;

struct Segments {
    const float* w[MAX_SEG];
    const float* bias[MAX_SEG];
    float* out[MAX_SEG];
    int N[MAX_SEG];
    int tile0[MAX_SEG];      // first column tile of the segment
    int use_add[MAX_SEG];
    int nseg;
};

template <bool HAS_ADD>
__global__ __launch_bounds__(256) void linear_small_kernel(
    const float* __restrict__ x, const float* __restrict__ xadd, Segments sg, int M, int K, int add_div,
    int add_mod, int relu) {
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    int seg = 0;                               // block-uniform
#pragma unroll
    for (int i = 1; i < MAX_SEG; ++i)
        if (i < sg.nseg && (int)blockIdx.x >= sg.tile0[i]) seg = i;
    const float* __restrict__ w = sg.w[seg];
    const float* __restrict__ bias = sg.bias[seg];
    float* __restrict__ out = sg.out[seg];
    const int N = sg.N[seg];
    const bool add_here = HAS_ADD && sg.use_add[seg];
    const int m0 = blockIdx.y * 16, n0 = ((int)blockIdx.x - sg.tile0[seg]) * 16;
    const int arow = min(m0 + r, M - 1);       // rows past the edge re-read the last row; never stored
    const int brow = min(n0 + r, N - 1);
    const float4* ap = reinterpret_cast<const float4*>(x + (long)arow * K) + kq;
    const float4* bp = reinterpret_cast<const float4*>(w + (long)brow * K) + kq;
    const float4* pp = nullptr;
    if (HAS_ADD) pp = reinterpret_cast<const float4*>(xadd + (long)((arow / add_div) % add_mod) * K) + kq;

    const int steps = K >> 4;                  // K-steps of 16; wave w takes steps w, w+4, ...
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = wave;
    // 8 steps (32 float4 loads) in flight per iteration
    for (; s + 28 < steps; s += 32) {
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = ap[(s + 4 * u) * 4];
            b[u] = bp[(s + 4 * u) * 4];
            if (add_here) {
                const float4 p = pp[(s + 4 * u) * 4];
                a[u].x += p.x; a[u].y += p.y; a[u].z += p.z; a[u].w += p.w;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
        }
    }
    for (; s < steps; s += 4) {
        float4 a = ap[s * 4];
        const float4 b = bp[s * 4];
        if (add_here) {
            const float4 p = pp[s * 4];
            a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    // acc[i] = D[row 4*kq + i][col r] of this wave's K share
#pragma unroll
    for (int i = 0; i < 4; ++i) part[wave][i * 64 + lane] = acc[i];
    __syncthreads();
    const int t = threadIdx.x;                 // element (i = t/64, lane' = t%64)
    const int i = t >> 6, l2 = t & 63;
    const int row = m0 + 4 * (l2 >> 4) + i, col = n0 + (l2 & 15);
    if (row < M && col < N) {
        float v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
        if (bias) v += bias[col];
        if (relu == 1) v = fmaxf(v, 0.f);
        else if (relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // exact GELU, as nn.GELU()
        out[(long)row * N + col] = v;
    }
}

}  // namespace

extern "C" int soc_linear_small_multi_f32(const float* x, const float* x_add, int add_div, int add_mod,
                                          int nseg, const float* const* w, const float* const* bias,
                                          float* const* out, const int* N, const int* use_add, int M,
                                          int K, int relu, void* stream) {
    if (M < 0 || K <= 0 || nseg <= 0 || !w || !out || !N) return SOC_EINVAL;
    if (nseg > MAX_SEG) return SOC_EUNSUPPORTED;
    if (x_add && (add_div <= 0 || add_mod <= 0)) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    Segments sg;
    int tiles = 0;
    bool any_add = false;
    for (int i = 0; i < MAX_SEG; ++i) {
        const int j = i < nseg ? i : 0;
        if (!w[j] || !out[j] || N[j] <= 0) return SOC_EINVAL;
        if (((uintptr_t)w[j] & 15) != 0) return SOC_EUNSUPPORTED;
        sg.w[i] = w[j]; sg.bias[i] = bias ? bias[j] : nullptr; sg.out[i] = out[j]; sg.N[i] = N[j];
        sg.use_add[i] = (x_add && (!use_add || use_add[j])) ? 1 : 0;
        sg.tile0[i] = tiles;
        if (i < nseg) {
            tiles += soc_ceil_div(N[j], 16);
            any_add |= sg.use_add[i] != 0;
        }
    }
    sg.nseg = nseg;
    if (!x) return SOC_EINVAL;
    if (K % 16 != 0 || M > 4096) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)x_add) & 15) != 0) return SOC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(tiles, soc_ceil_div(M, 16));
    if (any_add)
        hipLaunchKernelGGL(linear_small_kernel<true>, grid, dim3(256), 0, st, x, x_add, sg, M, K, add_div,
                           add_mod, relu);
    else
        hipLaunchKernelGGL(linear_small_kernel<false>, grid, dim3(256), 0, st, x, x_add, sg, M, K, 1, 1, relu);
    return soc_check_launch();
}

extern "C" int soc_linear_small_f32(const float* x, const float* x_add, int add_div, int add_mod,
                                    const float* w, const float* bias, float* out, int M, int N,
                                    int K, int relu, void* stream) {
    return soc_linear_small_multi_f32(x, x_add, add_div, add_mod, 1, &w, &bias, &out, &N, nullptr, M, K, relu,
                                      stream);
}
