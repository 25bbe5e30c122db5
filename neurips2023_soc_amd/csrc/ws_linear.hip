// K13: weight-stationary fp32 MFMA linear layer for the tall, short-K GEMMs of Video-Swin stages 0-1
// (reference models/video_swin_transformer.py:219-274: norm1 -> qkv, proj + residual, norm2 -> fc1 -> GELU,
// fc2 + residual; 115 200 tokens x C = 96 at stage 0 of the BASELINE config).
//
// Why: with K = 96..384 a 128 x 128-tile GEMM spends its time in tile prologues / epilogues, re-loads the
// same few weights for each of its thousands of tiles and pads N to whole tiles -- the tuned library kernels
// and K12 both run these layers at 59-67 TFLOP/s (tools/gemm_probe.py), and the LayerNorm in front and the
// residual add behind are separate 4-pass kernels (K5).  Here the WEIGHTS stay put:
//   * a workgroup (8 waves) loads W [N x K] (<= 152 KB) ONCE into LDS, laid out as ready-made MFMA operands
//     (one conflict-free ds_read_b128 per lane and 16-wide k group);
//   * each wave then streams 16-row tiles of x straight from global memory in MFMA operand layout (16 B per
//     lane and k group; nothing of x goes through LDS), the next tile's rows in flight during the current tile;
//   * LayerNorm of the input rows (the whole row is in the 4 lanes that share it: two xor-shuffles) is
//     applied in registers in front of the MFMAs; bias, exact-erf GELU / ReLU and the residual add are applied
//     to the accumulators; outputs leave as 16-B stores, two 16-column tiles at a time, so no output tile
//     ever waits in registers;
//   * N needs no padding beyond 16; layers whose W exceeds the LDS are split over column ranges (each range
//     its own set of workgroups, x re-read from L2).
// out = act(LN(x) W^T + b) + residual, every part optional.
#include "soc_common.h"
#include <atomic>
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512;
constexpr int LDS_FLOATS = 38912;     // 152 KB of weights per workgroup

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. ~1 ulp of the 1 + erf it is added to): 15 vector
// instructions instead of libm erff's ~30 -- the GELU sits on the accumulators of an MFMA kernel, where every
// VALU instruction is matrix-pipe time (f32 MFMA and VALU do not overlap on gfx950).
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);          // erf(|x| / sqrt 2)
    const float half = 0.5f * x;
    return fmaf(half, copysignf(erf_abs, x), half);        // 0.5 x (1 + erf(x / sqrt 2))
}

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return gelu_erf(v);
    return v;
}

template <int K, int ACT, bool HAS_LN, bool HAS_RES, int CT>
__global__ __launch_bounds__(THREADS, 2) void ws_linear_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ res,
    float* __restrict__ out, long M, int N, int nc_per_split, int wg_per_split) {
    constexpr int KG = K / 16;            // 16-wide k groups: one float4 per lane and group
    constexpr bool PREFETCH = K <= 192;   // next tile's rows in registers while this one computes
    extern __shared__ __attribute__((aligned(16))) float4 wimg[];   // [col tile][k group][lane], then the bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x / wg_per_split, wgi = blockIdx.x % wg_per_split;
    const int n_begin = split * nc_per_split;
    const int nc = min(nc_per_split, N - n_begin);
    const int nct = nc >> 4;
    // ---- weights -> LDS as MFMA operands: lane (r, kq) of column tile ct, k group j holds W[16 ct + r][16 j + 4 kq ..+3]
    for (int idx = tid; idx < nc * (K / 4); idx += THREADS) {
        const int n = idx / (K / 4), kk = idx - n * (K / 4);
        wimg[((n >> 4) * KG + (kk >> 2)) * 64 + (kk & 3) * 16 + (n & 15)] =
            *reinterpret_cast<const float4*>(w + (long)(n_begin + n) * K + kk * 4);
    }
    float4* bimg = wimg + (nc_per_split >> 4) * KG * 64;            // [nc / 4] bias (zeros without one)
    for (int idx = tid; idx < (nc >> 2); idx += THREADS)
        bimg[idx] = bias ? *reinterpret_cast<const float4*>(bias + n_begin + 4 * idx) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int r = lane & 15, kq = lane >> 4;
    float4 g4[HAS_LN ? KG : 1], b4[HAS_LN ? KG : 1];
    if (HAS_LN) {
#pragma unroll
        for (int j = 0; j < KG; ++j) {
            g4[j] = *reinterpret_cast<const float4*>(gamma + 16 * j + 4 * kq);
            b4[j] = *reinterpret_cast<const float4*>(beta + 16 * j + 4 * kq);
        }
    }
    // Work = (row tile, group of CT column tiles) units, dealt to the waves of this column range as CONTIGUOUS, equal
    // shares (+-1 unit): whole tiles would leave 7200 tiles on 2048 waves at 4 rounds for 3.5 rounds of work.  A tile
    // cut by a share boundary is loaded (and LayerNorm-ed) by both neighbours, each doing its part of the columns.
    const long ntiles = (M + 15) >> 4;
    const int gpt = nct / CT;                                   // groups per tile (the host picks CT | nct)
    const long units = ntiles * gpt, nwaves = (long)wg_per_split * (THREADS / 64);
    const long wv = (long)wgi * (THREADS / 64) + wave;
    const long g0 = wv * units / nwaves, g1 = (wv + 1) * units / nwaves;
    long t = g0 / gpt;
    float4 xn[PREFETCH ? KG : 1];
    auto load_rows = [&](long tile, float4 (&dst)[KG]) {
        const long m = min(tile * 16 + r, M - 1);
        const float4* xp = reinterpret_cast<const float4*>(x + m * K + 4 * kq);
#pragma unroll
        for (int j = 0; j < KG; ++j) dst[j] = xp[4 * j];
    };
    if (PREFETCH && g0 < g1) load_rows(t, reinterpret_cast<float4(&)[KG]>(xn));   // in flight behind the weight staging
    __syncthreads();
    for (; t * gpt < g1; ++t) {
        const int ct_lo = t * gpt < g0 ? (int)(g0 - t * gpt) * CT : 0;
        const int ct_hi = (t + 1) * gpt > g1 ? (int)(g1 - t * gpt) * CT : nct;
        float4 xf[KG];
        if (PREFETCH) {
#pragma unroll
            for (int j = 0; j < KG; ++j) xf[j] = xn[j];
            if ((t + 1) * gpt < g1) load_rows(t + 1, reinterpret_cast<float4(&)[KG]>(xn));
        } else {
            load_rows(t, xf);
        }
        if (HAS_LN) {   // row m = tile*16 + r lives in the 4 lanes (r, kq = 0..3): two-pass mean / variance
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KG; ++j) s += (xf[j].x + xf[j].y) + (xf[j].z + xf[j].w);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / K);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                xf[j].x -= mean; xf[j].y -= mean; xf[j].z -= mean; xf[j].w -= mean;
                v += (xf[j].x * xf[j].x + xf[j].y * xf[j].y) + (xf[j].z * xf[j].z + xf[j].w * xf[j].w);
            }
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const float rstd = rsqrtf(v * (1.0f / K) + eps);
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                xf[j].x = xf[j].x * rstd * g4[j].x + b4[j].x; xf[j].y = xf[j].y * rstd * g4[j].y + b4[j].y;
                xf[j].z = xf[j].z * rstd * g4[j].z + b4[j].z; xf[j].w = xf[j].w * rstd * g4[j].w + b4[j].w;
            }
        }
        const long m = t * 16 + r;
        const bool live = m < M;
        // lane (r, kq) ends up with out[m][n0 + 4 kq .. +3] of every column tile: 16-B stores
        float* orow = out + m * N + n_begin + 4 * kq;
        const float* rrow = HAS_RES ? res + m * N + n_begin + 4 * kq : nullptr;
        auto finish = [&](int ct, const f32x4& acc) {
            float4 o = make_float4(activate<ACT>(acc[0]), activate<ACT>(acc[1]), activate<ACT>(acc[2]), activate<ACT>(acc[3]));
            if (live) {
                if (HAS_RES) {
                    const float4 rr = *reinterpret_cast<const float4*>(rrow + ct * 16);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
                *reinterpret_cast<float4*>(orow + ct * 16) = o;
            }
        };
        // CT column tiles at a time (the host picks a CT that divides the tile count): their MFMAs alternate
        // accumulators (no dependent-issue stalls); the accumulators start from the bias
        for (int ct = ct_lo; ct < ct_hi; ct += CT) {
            f32x4 acc[CT];
            const float4* wp = wimg + (ct * KG) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const float4 bb = bimg[(ct + i) * 4 + kq];
                acc[i] = (f32x4){bb.x, bb.y, bb.z, bb.w};
            }
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                float4 wf[CT];
#pragma unroll
                for (int i = 0; i < CT; ++i) wf[i] = wp[(i * KG + j) * 64];
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].x, xf[j].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].y, xf[j].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].z, xf[j].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].w, xf[j].w, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) finish(ct + i, acc[i]);
        }
    }
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)

// columns of W per workgroup: even ranges, multiples of 16, that fit the LDS; 0 if N cannot be split that way
int split_columns(int N, int K) {
    const int nc_max = (LDS_FLOATS / (K + 1)) & ~15;
    for (int nsplit = (N + nc_max - 1) / nc_max; nsplit <= N / 16; ++nsplit)
        if ((N / 16) % nsplit == 0) return N / nsplit;
    return 0;
}

template <int K, int ACT, bool HAS_LN, bool HAS_RES, int CT>
int launch_ct(const float* x, const float* gamma, const float* beta, float eps, const float* w, const float* bias,
               const float* res, float* out, long M, int N, hipStream_t st) {
    const void* fn = reinterpret_cast<const void*>(ws_linear_kernel<K, ACT, HAS_LN, HAS_RES, CT>);
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];      // per instantiation and per device
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int nc_per_split = split_columns(N, K);
    const int nsplit = (N + nc_per_split - 1) / nc_per_split;
    const int cus = num_cus(st);
    const long ntiles = (M + 15) >> 4;
    long per_split = cus / nsplit > 0 ? cus / nsplit : 1;
    if (per_split * 8 > ntiles) per_split = (ntiles + 7) / 8;         // never more waves than row tiles
    const size_t lds = (size_t)nc_per_split * (K + 1) * sizeof(float);     // weights + bias
    hipLaunchKernelGGL((ws_linear_kernel<K, ACT, HAS_LN, HAS_RES, CT>), dim3((unsigned)(per_split * nsplit)), dim3(THREADS),
                       lds, st, x, gamma, beta, eps, w, bias, res, out, M, N, nc_per_split, (int)per_split);
    return soc_check_launch();
}

// column tiles per MFMA group: must divide every workgroup's tile count
template <int K, int ACT, bool HAS_LN, bool HAS_RES>
int launch_one(const float* x, const float* gamma, const float* beta, float eps, const float* w, const float* bias,
               const float* res, float* out, long M, int N, hipStream_t st) {
    const int nc = split_columns(N, K);
    if (nc == 0) return SOC_EUNSUPPORTED;
    const int nct = nc / 16;
    if (nct % 3 == 0) return launch_ct<K, ACT, HAS_LN, HAS_RES, 3>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
    if (nct % 2 == 0) return launch_ct<K, ACT, HAS_LN, HAS_RES, 2>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
    return launch_ct<K, ACT, HAS_LN, HAS_RES, 1>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
}

template <int K, int ACT>
int launch_k(const float* x, const float* gamma, const float* beta, float eps, const float* w, const float* bias,
             const float* res, float* out, long M, int N, hipStream_t st) {
    if (gamma) {
        if constexpr (K > 256) {       // a LayerNorm-ed input row is a model width (<= 256 here); gamma / beta live in registers
            return SOC_EUNSUPPORTED;
        } else {
            if (res) return launch_one<K, ACT, true, true>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
            return launch_one<K, ACT, true, false>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
        }
    }
    if (res) return launch_one<K, ACT, false, true>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
    return launch_one<K, ACT, false, false>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
}

template <int K>
int launch_act(int act, const float* x, const float* gamma, const float* beta, float eps, const float* w,
               const float* bias, const float* res, float* out, long M, int N, hipStream_t st) {
    if (act == 0) return launch_k<K, 0>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
    if (act == 1) return launch_k<K, 1>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
    return launch_k<K, 2>(x, gamma, beta, eps, w, bias, res, out, M, N, st);
}

}  // namespace

// K13b (ws_linear_split.hip): the same layer on the bf16 matrix cores, exact three-way split
int soc_ws_linear_split_dispatch(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* w,
                                 const float* bias, const float* residual, float* out, long M, int N, int K, int act,
                                 hipStream_t st);

extern "C" int soc_ws_linear_f32(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                 const float* w, const float* bias, const float* residual, float* out, long M,
                                 int N, int K, int act, int split_arith, void* stream) {
    if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return SOC_EINVAL;
    if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return SOC_EINVAL;
    if (M == 0) return SOC_OK;                        // empty input: nothing to launch (pointers may be null)
    if (!x || !w || !out) return SOC_EINVAL;
    if (N % 16 != 0) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)ln_gamma |
          (uintptr_t)ln_beta) & 15) != 0)
        return SOC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (split_arith != 0) {
        const int rc = soc_ws_linear_split_dispatch(x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, K, act, st);
        if (rc != SOC_EUNSUPPORTED) return rc;
    }
    switch (K) {
        case 96: return launch_act<96>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        case 128: return launch_act<128>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        case 192: return launch_act<192>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        case 256: return launch_act<256>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        case 384: return launch_act<384>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        case 512: return launch_act<512>(act, x, ln_gamma, ln_beta, ln_eps, w, bias, residual, out, M, N, st);
        default: return SOC_EUNSUPPORTED;
    }
}
