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
// Two launches, both streaming x once with 16-B accesses.  A row of C floats is C/4 lanes wide (a power of two <= 64), so a
// wave covers 64 / (C/4) rows per load: one for the C = 256 of input_proj, 16 for the C = 16 of the FPN's last GroupNorm.
// A lane's 4 channels lie in one group when (C/G) % 4 == 0; C/G == 2 (two groups per lane) is the other supported form:
//   stats:     grid (chunks, N); per (frame, row chunk, group) partial sum / sum of squares of
//              (x - pivot), pivot = first element of the group -- shifted sums, so no cancellation when
//              |mean| >> std -- written to the workspace; no atomics, so the result is run-to-run
//              bit-identical;
//   normalise: every workgroup first folds the chunk partials of its frame (double precision) into
//              mean / rstd in LDS, then streams its rows.
#include "soc_common.h"

namespace {

// Rows of S handled by one workgroup: 64 rows of C = 256, more for narrower rows (16 KB of input per workgroup at least),
// so that the number of chunk partials every workgroup of the second launch folds stays small (S = 14 400, C = 16: 15
// chunks instead of 225).
__host__ __device__ inline int rows_per_block(int C) { return 64 * (256 / C > 0 ? 256 / C : 1); }

// Sums / parameters of the (up to) two groups a lane's float4 belongs to: components x, y -> [0], z, w -> [1]; with
// (C/G) % 4 == 0 both are the same group and only [0] is used.
struct Pair { float a, b; };

// partial[(n * chunks + chunk) * G + g] = (sum, sumsq) of (x - pivot[n, g]), pivot = first element of the group
template <bool CPG2>
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float2* __restrict__ partial,
                                                       int S, int C, int G, int chunks) {
    __shared__ float4 red[4][64];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;                    // float4 per row: a power of two <= 64
    const int rpi = 64 / nvec;                  // rows per wave-wide load
    const int vec = lane & (nvec - 1), rsub = lane / nvec;
    const int cpg = C / G;
    const float* xn = x + (long)n * S * C;
    const int g0 = CPG2 ? 2 * vec : (4 * vec) / cpg;
    const float piv0 = xn[g0 * cpg], piv1 = CPG2 ? xn[(g0 + 1) * cpg] : piv0;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    const int rpb = rows_per_block(C);
    const int r1 = min((chunk + 1) * rpb, S);
#pragma unroll 4
    for (int r = chunk * rpb + wave * rpi + rsub; r < r1; r += 4 * rpi) {
        const float4 v = reinterpret_cast<const float4*>(xn + (long)r * C)[vec];
        const float a = v.x - piv0, b = v.y - piv0, c = v.z - piv1, d = v.w - piv1;
        if (CPG2) {
            s0 += a + b; q0 += a * a + b * b;
            s1 += c + d; q1 += c * c + d * d;
        } else {
            s0 += (a + b) + (c + d);
            q0 += (a * a + b * b) + (c * c + d * d);
        }
    }
    // lanes of one group within a row are adjacent (cpg / 4 of them), the rows of a wave-wide load are nvec lanes apart
    if (!CPG2)
        for (int o = 1; o < (cpg >> 2); o <<= 1) {
            s0 += __shfl_xor(s0, o);
            q0 += __shfl_xor(q0, o);
        }
    for (int o = nvec; o < 64; o <<= 1) {
        s0 += __shfl_xor(s0, o); q0 += __shfl_xor(q0, o);
        if (CPG2) { s1 += __shfl_xor(s1, o); q1 += __shfl_xor(q1, o); }
    }
    red[wave][lane] = make_float4(s0, q0, s1, q1);
    __syncthreads();
    if (threadIdx.x < G) {
        const int t = threadIdx.x;
        const int l0 = CPG2 ? t >> 1 : t * (cpg >> 2);      // a lane of row-subgroup 0 that holds the group
        const float4 a = red[0][l0], b = red[1][l0], c = red[2][l0], d = red[3][l0];
        const bool hi = CPG2 && (t & 1);
        const float ss = hi ? (a.z + b.z) + (c.z + d.z) : (a.x + b.x) + (c.x + d.x);
        const float qq = hi ? (a.w + b.w) + (c.w + d.w) : (a.y + b.y) + (c.y + d.y);
        partial[((long)n * chunks + chunk) * G + t] = make_float2(ss, qq);
    }
}

template <bool CPG2>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float2* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int S, int C, int G, int chunks,
                                                       float eps, int relu) {
    __shared__ float mean_s[64], rstd_s[64];
    __shared__ double part_s[4][64], part_q[4][64];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2, cpg = C / G;
    const int rpi = 64 / nvec;
    const int vec = lane & (nvec - 1), rsub = lane / nvec;
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
    const int g0 = CPG2 ? 2 * vec : (4 * vec) / cpg, g1 = CPG2 ? g0 + 1 : g0;
    const float mu0 = mean_s[g0], rs0 = rstd_s[g0], mu1 = mean_s[g1], rs1 = rstd_s[g1];
    const float4 gm = reinterpret_cast<const float4*>(gamma)[vec];
    const float4 bt = reinterpret_cast<const float4*>(beta)[vec];
    const float4 sc = make_float4(gm.x * rs0, gm.y * rs0, gm.z * rs1, gm.w * rs1);
    const float4 sh = make_float4(bt.x - mu0 * sc.x, bt.y - mu0 * sc.y, bt.z - mu1 * sc.z, bt.w - mu1 * sc.w);
    float* on = out + (long)n * S * C;
    const int rpb = rows_per_block(C);
    const int r1 = min((chunk + 1) * rpb, S);
#pragma unroll 4
    for (int r = chunk * rpb + wave * rpi + rsub; r < r1; r += 4 * rpi) {
        const float4 v = reinterpret_cast<const float4*>(xn + (long)r * C)[vec];
        float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(on + (long)r * C)[vec] = o;
    }
}

}  // namespace

extern "C" size_t soc_groupnorm_tokens_workspace_bytes(int N, int S, int C, int G) {
    if (N <= 0 || S <= 0 || G <= 0 || C <= 0) return 0;
    return (size_t)N * soc_ceil_div(S, rows_per_block(C)) * G * sizeof(float2);
}

extern "C" int soc_groupnorm_tokens_f32(const float* x, const float* gamma, const float* beta, float* out, int N,
                                        int S, int C, int G, float eps, int relu, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (N < 0 || S <= 0 || C <= 0 || G <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!x || !gamma || !beta || !out) return SOC_EINVAL;
    // a row = C/4 lanes (a power of two <= 64); a lane's float4 inside one group, or exactly two groups per lane
    const int cpg = C / G, nvec = C / 4;
    const bool whole = cpg % 4 == 0 && ((cpg / 4) & (cpg / 4 - 1)) == 0;
    if (C % G != 0 || C > 256 || C % 4 != 0 || (nvec & (nvec - 1)) != 0 || !(whole || cpg == 2) || G > 64)
        return SOC_EUNSUPPORTED;
    if (!workspace || workspace_bytes < soc_groupnorm_tokens_workspace_bytes(N, S, C, G)) return SOC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = soc_ceil_div(S, rows_per_block(C));
    float2* partial = (float2*)workspace;
    if (whole) hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(chunks, N), dim3(256), 0, st, x, partial, S, C, G, chunks);
    else hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(chunks, N), dim3(256), 0, st, x, partial, S, C, G, chunks);
    if (soc_check_launch() != SOC_OK) return SOC_ELAUNCH;
    if (whole)
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(chunks, N), dim3(256), 0, st, x, partial, gamma, beta, out, S, C, G,
                           chunks, eps, relu);
    else
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(chunks, N), dim3(256), 0, st, x, partial, gamma, beta, out, S, C, G,
                           chunks, eps, relu);
    return soc_check_launch();
}
