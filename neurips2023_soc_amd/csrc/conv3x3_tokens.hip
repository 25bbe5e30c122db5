// K19: 3x3 / stride 1 / pad 1 convolution on token-major (channels-last) maps as an implicit GEMM on f32 MFMA -- the
// convolutions of the FPN spatial decoder (reference models/segmentation.py:41-74: lay1..lay5, out_lay).
//
//   out[(n, y, x), co] = bias[co] + sum_{ky, kx, ci} in[n, y + ky - 1, x + kx - 1, ci] * w[co, ci, ky, kx]
//
// Why not the library: MIOpen runs these six layers (6.9 GFLOP per clip, 1 920 .. 115 200 pixels, 256 .. 8 output channels)
// at 10-50 TFLOP/s in NCHW (0.22 ms of convolution kernels per clip plus layout copies), and in the software pipeline the
// FPN runs beside the next clip's head, which is charged the CU time (0.26 ms per clip, tools/experiments/tail_ablation.py).
// The encoder output and the backbone map are token-major already, so nothing is transposed on the way in.
//
// Mapping (K7's operand scheme): GEMM rows = pixels, K = 9 * Cin ordered (tap, ci) -- the host passes the weight as
// [Cout][9 * Cin] -- so a K-step of 16 lies inside one tap and a lane's A operand is 16 contiguous bytes of one
// neighbouring pixel (zero outside the map).  A workgroup = 16 * RT pixels x 16 * NT output channels; its 4 waves split
// the K-steps round-robin (lay1: 144 steps) and their partial tiles are summed through LDS.  Per step a lane loads RT A +
// NT B float4 and issues 4 * RT * NT v_mfma_f32_16x16x4_f32.  RT = 2 for the small maps (enough workgroups), 4 for the
// large ones (the weights are re-read once per workgroup: a quarter of the L2 traffic of 16-pixel tiles).
#include "soc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* in;     // [N][H*W][Cin], frame stride in_fs elements
    const float* w;      // [Cout][9*Cin]
    const float* bias;   // [Cout] or null
    float* out;          // token-major [N*H*W][Cout], or NCHW [N][Cout][H][W]
    long in_fs;
    int N, H, W, Cin, Cout, out_nchw, relu;
};

template <int RT, int NT>
__global__ __launch_bounds__(256) void conv3x3_tokens_kernel(const ConvArgs a) {
    __shared__ float part[4][RT * NT][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int HW = a.H * a.W;
    const long M = (long)a.N * HW;
    const long m0 = (long)blockIdx.x * (16 * RT);
    const int n0 = blockIdx.y * 16 * NT;
    const int K9 = 9 * a.Cin;

    const float* in_n[RT];
    int py[RT], px[RT];
    bool live[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const long m = m0 + rt * 16 + r;
        live[rt] = m < M;
        const long mc = live[rt] ? m : M - 1;
        const int n = (int)(mc / HW), rem = (int)(mc - (long)n * HW);
        py[rt] = rem / a.W;
        px[rt] = rem - py[rt] * a.W;
        in_n[rt] = a.in + (long)n * a.in_fs + kq * 4;
    }
    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.w + (long)min(n0 + t * 16 + r, a.Cout - 1) * K9 + kq * 4;

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load_a = [&](int rt, int tap, int ci) -> float4 {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;          // wave-uniform
        const int yy = py[rt] + dy, xx = px[rt] + dx;
        if (live[rt] && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
            return *reinterpret_cast<const float4*>(in_n[rt] + (long)(yy * a.W + xx) * a.Cin + ci);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mma = [&](const float4 (&av)[RT], const float4 (&bv)[NT]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].x, bv[t].x, acc[rt][t], 0, 0, 0);
                acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].y, bv[t].y, acc[rt][t], 0, 0, 0);
                acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].z, bv[t].z, acc[rt][t], 0, 0, 0);
                acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt].w, bv[t].w, acc[rt][t], 0, 0, 0);
            }
    };

    const int steps = K9 >> 4;             // K-steps of 16 (Cin % 16 == 0: a step lies inside one tap)
    int s = wave;
    constexpr int U = 8 / RT;              // steps in flight per iteration
    for (; s + 4 * (U - 1) < steps; s += 4 * U) {
        float4 av[U][RT], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = (s + 4 * u) << 4;
            const int tap = kk / a.Cin, ci = kk - tap * a.Cin;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[u][rt] = load_a(rt, tap, ci);
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[u][t] = *reinterpret_cast<const float4*>(wp[t] + kk);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) mma(av[u], bv[u]);
    }
    for (; s < steps; s += 4) {
        const int kk = s << 4;
        const int tap = kk / a.Cin, ci = kk - tap * a.Cin;
        float4 av[RT], bv[NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[rt] = load_a(rt, tap, ci);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const float4*>(wp[t] + kk);
        mma(av, bv);
    }

    // acc[rt][t][i] = D[row 4*kq + i][col r] of this wave's share of K
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[wave][rt * NT + t][i * 64 + lane] = acc[rt][t][i];
    __syncthreads();
    const int tid = threadIdx.x;
    const int i = tid >> 6, l2 = tid & 63;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const long row = m0 + rt * 16 + 4 * (l2 >> 4) + i;
            const int col = n0 + t * 16 + (l2 & 15);
            if (row < M && col < a.Cout) {
                const int q = rt * NT + t;
                float v = (part[0][q][tid] + part[1][q][tid]) + (part[2][q][tid] + part[3][q][tid]);
                if (a.bias) v += a.bias[col];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.out_nchw) {
                    const int n = (int)(row / HW), rem = (int)(row - (long)n * HW);
                    a.out[((long)n * a.Cout + col) * HW + rem] = v;
                } else {
                    a.out[row * a.Cout + col] = v;
                }
            }
        }
}

}  // namespace

extern "C" int soc_conv3x3_tokens_f32(const float* in, long in_frame_stride, const float* w_taps, const float* bias,
                                      float* out, int N, int H, int W, int Cin, int Cout, int out_nchw, int relu,
                                      void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!in || !w_taps || !out || in_frame_stride < (long)H * W * Cin) return SOC_EINVAL;
    if (Cin % 16 || (((uintptr_t)in | (uintptr_t)w_taps) & 15) || (in_frame_stride & 3)) return SOC_EUNSUPPORTED;
    ConvArgs a;
    a.in = in; a.w = w_taps; a.bias = bias; a.out = out; a.in_fs = in_frame_stride;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.out_nchw = out_nchw; a.relu = relu;
    const long M = (long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    const bool big = M >= 16384;                 // enough 64-pixel tiles to fill the chip
    const unsigned gx = (unsigned)((M + (big ? 63 : 31)) / (big ? 64 : 32));
    const unsigned gy = Cout > 16 ? soc_ceil_div(Cout, 32) : 1;
    if (Cout > 16) {
        if (big) hipLaunchKernelGGL((conv3x3_tokens_kernel<4, 2>), dim3(gx, gy), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_tokens_kernel<2, 2>), dim3(gx, gy), dim3(256), 0, st, a);
    } else {
        if (big) hipLaunchKernelGGL((conv3x3_tokens_kernel<4, 1>), dim3(gx, gy), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_tokens_kernel<2, 1>), dim3(gx, gy), dim3(256), 0, st, a);
    }
    return soc_check_launch();
}
