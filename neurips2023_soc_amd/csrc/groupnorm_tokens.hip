// K10: GroupNorm over token-major activations for gfx950 (HBM-bound).
//
//   x [N, S, C] (N frames, S = h*w pixels, C channels)   ->   out [N, S, C]
//   out = (x - mean[n, g]) * rsqrt(var[n, g] + eps) * gamma[c] + beta[c],  g = c / (C / G),
//   statistics over the S * (C / G) elements of (frame n, group g), biased variance (torch.nn.GroupNorm).
//
// Why: Video-Swin emits token-major maps and the vision-language fusion consumes token-major
// sequences, but the reference's input_proj = Conv2d(1x1) + GroupNorm(32) is written for NCHW
// (models/soc.py:107-125, 226-230), which costs a layout copy on either side of the GroupNorm.  With this
// kernel the 1x1 convolution is a plain GEMM over tokens and nothing is ever transposed.
//
// Two launches, both streaming x once with 16-B accesses (a row of C = 256 floats is one wave-wide
// float4 load; a lane's 4 channels lie in one group when (C/G) % 4 == 0):
//   stats:     grid (chunks, N); per (frame, row chunk, group) partial sum / sum of squares of
//              (x - pivot), pivot = first element of the group -- shifted sums, so no cancellation when
//              |mean| >> std -- written to the workspace; no atomics, so the result is run-to-run
//              bit-identical;
//   normalise: every workgroup first folds the chunk partials of its frame (double precision) into
//              mean / rstd in LDS, then streams its rows.
#include "soc_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 64;   // rows of S handled by one workgroup (4 waves x 16 rows)

// partial[(n * chunks + chunk) * G + g] = (sum, sumsq) of (x - pivot[n, g])
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float2* __restrict__ partial,
                                                       int S, int C, int G, int chunks) {
    __shared__ float2 red[4][64];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;                    // float4 per row (<= 64)
    const int cpg4 = (C / G) >> 2;              // float4 per group
    const float* xn = x + (long)n * S * C;
    float s = 0.f, q = 0.f;
    if (lane < nvec) {
        const float pivot = xn[(lane / cpg4) * (C / G)];
        const int r0 = chunk * ROWS_PER_BLOCK + wave;
        const int r1 = min((chunk + 1) * ROWS_PER_BLOCK, S);
#pragma unroll 4
        for (int r = r0; r < r1; r += 4) {
            const float4 v = reinterpret_cast<const float4*>(xn + (long)r * C)[lane];
            const float a = v.x - pivot, b = v.y - pivot, c = v.z - pivot, d = v.w - pivot;
            s += (a + b) + (c + d);
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    // lanes of one group are adjacent (cpg4 of them, a power of two <= 64)
    for (int o = 1; o < cpg4; o <<= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    red[wave][lane] = make_float2(s, q);
    __syncthreads();
    if (threadIdx.x < G) {
        const int l0 = threadIdx.x * cpg4;
        float2 a = red[0][l0], b = red[1][l0], c = red[2][l0], d = red[3][l0];
        partial[((long)n * chunks + chunk) * G + threadIdx.x] =
            make_float2((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y));
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float2* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int S, int C, int G, int chunks,
                                                       float eps) {
    __shared__ float mean_s[64], rstd_s[64];
    __shared__ double part_s[4][64], part_q[4][64];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2, cpg = C / G, cpg4 = cpg >> 2;
    const float* xn = x + (long)n * S * C;
    // fold the chunk partials of this frame: lane = group, the 4 waves take every 4th chunk (independent loads,
    // a fixed order -> bit-identical from run to run), then one thread per group adds the 4 slices
    {
        double s = 0.0, q = 0.0;
        if (lane < G) {
            for (int c = wave; c < chunks; c += 4) {
                const float2 p = partial[((long)n * chunks + c) * G + lane];
                s += p.x; q += p.y;
            }
        }
        part_s[wave][lane] = s; part_q[wave][lane] = q;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int t = threadIdx.x;
        const double s = (part_s[0][t] + part_s[1][t]) + (part_s[2][t] + part_s[3][t]);
        const double q = (part_q[0][t] + part_q[1][t]) + (part_q[2][t] + part_q[3][t]);
        const double cnt = (double)S * cpg;
        const double md = s / cnt;                                  // mean of (x - pivot)
        const double var = fmax(q / cnt - md * md, 0.0);
        mean_s[t] = (float)((double)xn[t * cpg] + md);
        rstd_s[t] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (lane >= nvec) return;
    const int g = lane / cpg4;
    const float mu = mean_s[g], rs = rstd_s[g];
    const float4 gm = reinterpret_cast<const float4*>(gamma)[lane];
    const float4 bt = reinterpret_cast<const float4*>(beta)[lane];
    const float4 sc = make_float4(gm.x * rs, gm.y * rs, gm.z * rs, gm.w * rs);
    const float4 sh = make_float4(bt.x - mu * sc.x, bt.y - mu * sc.y, bt.z - mu * sc.z, bt.w - mu * sc.w);
    float* on = out + (long)n * S * C;
    const int r1 = min((chunk + 1) * ROWS_PER_BLOCK, S);
#pragma unroll 4
    for (int r = chunk * ROWS_PER_BLOCK + wave; r < r1; r += 4) {
        const float4 v = reinterpret_cast<const float4*>(xn + (long)r * C)[lane];
        reinterpret_cast<float4*>(on + (long)r * C)[lane] =
            make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    }
}

}  // namespace

extern "C" size_t soc_groupnorm_tokens_workspace_bytes(int N, int S, int C, int G) {
    (void)C;
    if (N <= 0 || S <= 0 || G <= 0) return 0;
    return (size_t)N * soc_ceil_div(S, ROWS_PER_BLOCK) * G * sizeof(float2);
}

extern "C" int soc_groupnorm_tokens_f32(const float* x, const float* gamma, const float* beta, float* out, int N,
                                        int S, int C, int G, float eps, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    if (N < 0 || S <= 0 || C <= 0 || G <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!x || !gamma || !beta || !out) return SOC_EINVAL;
    // one wave-wide float4 load per row; a lane's float4 inside one group; adjacent-lane reduction
    const int cpg = C / G;
    if (C % G != 0 || C > 256 || C % 4 != 0 || cpg % 4 != 0 || ((cpg / 4) & (cpg / 4 - 1)) != 0 || G > 64)
        return SOC_EUNSUPPORTED;
    if (!workspace || workspace_bytes < soc_groupnorm_tokens_workspace_bytes(N, S, C, G)) return SOC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = soc_ceil_div(S, ROWS_PER_BLOCK);
    float2* partial = (float2*)workspace;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, N), dim3(256), 0, st, x, partial, S, C, G, chunks);
    if (soc_check_launch() != SOC_OK) return SOC_ELAUNCH;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, N), dim3(256), 0, st, x, partial, gamma, beta, out, S, C, G,
                       chunks, eps);
    return soc_check_launch();
}
