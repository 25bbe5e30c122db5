// K13b: the weight-stationary linear layer (K13, ws_linear.hip) on the bf16 matrix cores, exact three-way operand split.
//
// Same contract and the same data flow as K13 -- out = act(LN(x) W^T + b) + residual for the tall, short-K layers of
// Video-Swin stage 0 (reference models/video_swin_transformer.py:219-274; 115 200 tokens x C = 96 at the BASELINE config):
// the weights stay in LDS as ready-made MFMA operands, every wave streams row tiles of x straight from global memory in
// operand layout, LayerNorm runs in registers on the four lanes that share a row, bias / GELU / residual sit on the
// accumulators -- but the products run as v_mfma_f32_16x16x32_bf16 on exact bf16 splits (linear_split.hip explains the
// arithmetic: a = a0 + a1 + a2 exactly, six products, dropped terms <= 2^-23 |a b|).  On the f32-input MFMA these layers are
// matrix-pipe-bound at 0.39-0.57 of its peak; six bf16 MFMAs take 6/16 of that time and leave the vector
// ALU free, which moves the layers to the HBM side of their roofline (qkv: 177 MB per launch).
//   * LDS image: [column tile][k step of 32][plane][lane] 16-B pieces = W[16 ct + (lane & 15)][32 s + 8 (lane >> 4) .. + 7]
//     of plane pl, split from the f32 weights while they are staged (6 bytes per weight: wider layers use more column ranges);
//   * a lane (r, kq) loads x[m = 16 tile + r][32 s + 8 kq .. + 7]: two 16-B loads per step, the four kq lanes of a row cover
//     128 contiguous bytes; LayerNorm as in K13; the normalised values are split in registers;
//   * RT row tiles share every weight fragment read (RT = 2 for K <= 96), halving the LDS traffic per MFMA.
// The kernel mixes bf16 MFMAs with LDS traffic, so it follows the co-residence rule of DESIGN.md section 3: it claims all
// 256 VGPRs, its waves retire behind a barrier, its packed f32 arithmetic uses VGPR operands only (tests/test_isa_rules.py).
#include "soc_common.h"
#include <atomic>
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512;
constexpr int LDS_BYTES = 152 * 1024;

__device__ __forceinline__ float in_vgpr(float c) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(c));
    return r;
}

struct GeluK { float rs2, a0, one, c5, c4, c3, c2, c1, nlog2e, half; };
__device__ __forceinline__ GeluK gelu_k() {
    return {in_vgpr(0.70710678118654752f), in_vgpr(0.3275911f), in_vgpr(1.0f), in_vgpr(1.061405429f), in_vgpr(-1.453152027f),
            in_vgpr(1.421413741f), in_vgpr(-0.284496736f), in_vgpr(0.254829592f), in_vgpr(-1.4426950408889634f),
            in_vgpr(0.5f)};
}
// exact (erf) GELU, erf by Abramowitz & Stegun 7.1.26 as in K13 / K20; constants in VGPRs
__device__ __forceinline__ float gelu_erf(float x, const GeluK& k) {
    const float z = fabsf(x) * k.rs2;
    const float t = __builtin_amdgcn_rcpf(fmaf(k.a0, z, k.one));
    float p = fmaf(k.c5, t, k.c4);
    p = fmaf(p, t, k.c3);
    p = fmaf(p, t, k.c2);
    p = fmaf(p, t, k.c1);
    const float e = __builtin_amdgcn_exp2f(z * z * k.nlog2e);
    const float erf_abs = fmaf(-p * t, e, k.one);
    const float half = k.half * x;
    return fmaf(half, copysignf(erf_abs, x), half);
}

// f32 x 8 -> three bf16 x 8 with a0 + a1 + a2 == a exactly
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        h0[i] = a0; h1[i] = a1; h2[i] = (__bf16)r2;
    }
}

// FLY (K > 256, no LayerNorm, one column group per row group): the rows are split step by step inside the MFMA loop instead
// of up front -- K / 32 x 12 VGPRs of split operands would not fit beside the K / 4 raw words of a K = 384 / 512 row
template <int K, int ACT, bool HAS_LN, bool HAS_RES, int CT, int RT, bool FLY>
__global__ __launch_bounds__(THREADS, 2) void ws_linear_split_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ res,
    float* __restrict__ out, long M, int N, int nc_per_split, int wg_per_split) {
    constexpr int KS = K / 32;            // k steps of 32: two float4 per lane and step
    extern __shared__ __attribute__((aligned(16))) u32x4 wimg[];    // [col tile][k step][plane][lane], then the bias
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                   // own the CU (see the header)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x / wg_per_split, wgi = blockIdx.x % wg_per_split;
    const int n_begin = split * nc_per_split;
    const int nc = min(nc_per_split, N - n_begin);
    const int nct = nc >> 4;
    // ---- weights -> LDS: item (n, s, kq) = W[n][32 s + 8 kq .. + 7], split into the three planes
    for (int idx = tid; idx < nc * (K / 8); idx += THREADS) {
        const int n = idx / (K / 8), c = idx - n * (K / 8);
        const int s = c >> 2, kq = c & 3;
        const float4* src = reinterpret_cast<const float4*>(w + (long)(n_begin + n) * K + 8 * c);
        const float4 lo = src[0], hi = src[1];
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        bf16x8 h0, h1, h2;
        split8(v, h0, h1, h2);
        u32x4* dst = wimg + (((n >> 4) * KS + s) * 3) * 64 + kq * 16 + (n & 15);
        dst[0] = __builtin_bit_cast(u32x4, h0);
        dst[64] = __builtin_bit_cast(u32x4, h1);
        dst[128] = __builtin_bit_cast(u32x4, h2);
    }
    float4* bimg = reinterpret_cast<float4*>(wimg + (nc_per_split >> 4) * KS * 3 * 64);   // [nc / 4] bias (zeros without one)
    for (int idx = tid; idx < (nc >> 2); idx += THREADS)
        bimg[idx] = bias ? *reinterpret_cast<const float4*>(bias + n_begin + 4 * idx) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int r = lane & 15, kq = lane >> 4;
    // gamma / beta of the LayerNorm live in LDS behind the bias (in registers they would be 4 K / 8 VGPRs the MFMA loop
    // does not have): [K / 4] float4 each
    float4* gimg = bimg + (nc_per_split >> 2);
    float4* eimg = gimg + K / 4;
    if (HAS_LN) {
        for (int idx = tid; idx < K / 4; idx += THREADS) {
            gimg[idx] = *reinterpret_cast<const float4*>(gamma + 4 * idx);
            eimg[idx] = *reinterpret_cast<const float4*>(beta + 4 * idx);
        }
    }
    const float inv_k = in_vgpr(1.0f / K), eps_v = in_vgpr(eps);
    // Work = (group of RT row tiles, group of CT column tiles) units, dealt to the waves of this column range as contiguous,
    // equal shares (as K13): a row group cut by a share boundary is loaded by both neighbours
    const long ngroups = (M + 16 * RT - 1) / (16 * RT);
    const int gpt = nct / CT;                                   // column groups per row group (the host picks CT | nct)
    const long units = ngroups * gpt, nwaves = (long)wg_per_split * (THREADS / 64);
    const long wv = (long)wgi * (THREADS / 64) + wave;
    const long g0 = wv * units / nwaves, g1 = (wv + 1) * units / nwaves;
    float4 xn[RT][KS][2];
    auto load_rows = [&](long grp) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long m = min((grp * RT + rt) * 16 + r, M - 1);
            const float4* xp = reinterpret_cast<const float4*>(x + m * K + 8 * kq);
#pragma unroll
            for (int s = 0; s < KS; ++s) { xn[rt][s][0] = xp[8 * s]; xn[rt][s][1] = xp[8 * s + 1]; }
        }
    };
    long t = g0 / gpt;
    if (g0 < g1) load_rows(t);                                  // in flight behind the weight staging
    __syncthreads();
    for (; t * gpt < g1; ++t) {
        const int ct_lo = t * gpt < g0 ? (int)(g0 - t * gpt) * CT : 0;
        const int ct_hi = (t + 1) * gpt > g1 ? (int)(g1 - t * gpt) * CT : nct;
        bf16x8 xb[FLY ? 1 : RT][FLY ? 1 : KS][3];
#pragma unroll
        for (int rt = 0; rt < (FLY ? 0 : RT); ++rt) {
            float v[KS][8];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                v[s][0] = xn[rt][s][0].x; v[s][1] = xn[rt][s][0].y; v[s][2] = xn[rt][s][0].z; v[s][3] = xn[rt][s][0].w;
                v[s][4] = xn[rt][s][1].x; v[s][5] = xn[rt][s][1].y; v[s][6] = xn[rt][s][1].z; v[s][7] = xn[rt][s][1].w;
            }
            if (HAS_LN) {   // row m lives in the 4 lanes (r, kq = 0..3): two-pass mean / variance, as K13
                float sm = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int i = 0; i < 8; ++i) sm += v[s][i];
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                const float mean = sm * inv_k;
                float q = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[s][i] -= mean; q = fmaf(v[s][i], v[s][i], q); }
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                const float rstd = rsqrtf(fmaf(q, inv_k, eps_v));
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float4 ga = gimg[8 * s + 2 * kq], gb = gimg[8 * s + 2 * kq + 1];
                    const float4 ea = eimg[8 * s + 2 * kq], eb = eimg[8 * s + 2 * kq + 1];
                    const float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                    const float bb[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[s][i] = fmaf(v[s][i] * rstd, gg[i], bb[i]);
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) split8(v[s], xb[rt][s][0], xb[rt][s][1], xb[rt][s][2]);
        }
        if (!FLY && (t + 1) * gpt < g1) load_rows(t + 1);       // the next group's rows fly during this group's MFMAs
        if (FLY && t != g0 / gpt) load_rows(t);                 // (the first group's rows were loaded behind the weight staging)
        auto column_group = [&](int ct) {
            f32x4 acc[RT][CT];
            const u32x4* wp = wimg + (ct * KS * 3) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const float4 bq = bimg[(ct + i) * 4 + kq];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][i] = (f32x4){bq.x, bq.y, bq.z, bq.w};
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bf16x8 wf[CT][3];
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wf[i][pl] = __builtin_bit_cast(bf16x8, wp[((i * KS + s) * 3 + pl) * 64]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    bf16x8 xs[3];
                    if (FLY) {
                        float v[8] = {xn[rt][s][0].x, xn[rt][s][0].y, xn[rt][s][0].z, xn[rt][s][0].w,
                                      xn[rt][s][1].x, xn[rt][s][1].y, xn[rt][s][1].z, xn[rt][s][1].w};
                        split8(v, xs[0], xs[1], xs[2]);
                    } else {
                        xs[0] = xb[FLY ? 0 : rt][FLY ? 0 : s][0]; xs[1] = xb[FLY ? 0 : rt][FLY ? 0 : s][1];
                        xs[2] = xb[FLY ? 0 : rt][FLY ? 0 : s][2];
                    }
#pragma unroll
                    for (int i = 0; i < CT; ++i) {      // smallest terms first
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][2], xs[0], acc[rt][i], 0, 0, 0);
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][1], xs[1], acc[rt][i], 0, 0, 0);
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][0], xs[2], acc[rt][i], 0, 0, 0);
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][1], xs[0], acc[rt][i], 0, 0, 0);
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][0], xs[1], acc[rt][i], 0, 0, 0);
                        acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][0], xs[0], acc[rt][i], 0, 0, 0);
                    }
                }
                if (FLY && (s & 1)) __builtin_amdgcn_sched_barrier(0);   // keep the weight reads of later steps from piling up
            }
            // lane (r, kq) holds out[m][n0 + 4 kq .. + 3] of every column tile: 16-B stores
            const GeluK gk = gelu_k();      // materialised here: ten registers the MFMA loop does not carry
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const long m = (t * RT + rt) * 16 + r;
                if (m < M) {
                    float* orow = out + m * N + n_begin + 4 * kq;
#pragma unroll
                    for (int i = 0; i < CT; ++i) {
                        float4 o = make_float4(acc[rt][i][0], acc[rt][i][1], acc[rt][i][2], acc[rt][i][3]);
                        if (ACT == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                        if (ACT == 2) { o.x = gelu_erf(o.x, gk); o.y = gelu_erf(o.y, gk); o.z = gelu_erf(o.z, gk); o.w = gelu_erf(o.w, gk); }
                        if (HAS_RES) {
                            const float4 rr = *reinterpret_cast<const float4*>(res + m * N + n_begin + 4 * kq + (ct + i) * 16);
                            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                        }
                        *reinterpret_cast<float4*>(orow + (ct + i) * 16) = o;
                    }
                }
            }
        };
        if (FLY) {          // one column group per row group (the host checks): no loop the split could be hoisted out of
            column_group(0);
        } else {
            for (int ct = ct_lo; ct < ct_hi; ct += CT) column_group(ct);
        }
    }
    __syncthreads();        // the waves retire together: no foreign wave beside a partner that still issues MFMAs
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)

// columns of W per workgroup: even ranges, multiples of 16, whose three planes (+ bias) fit the LDS; 0 if impossible.
// K > 256 (rows split step by step): a range is ONE group of 3, 2 or 1 column tiles.
int split_columns(int N, int K) {
    const int nc_max = ((LDS_BYTES - 8 * K) / (6 * K + 4)) & ~15;          // 8 K bytes: gamma / beta
    if (nc_max <= 0) return 0;
    if (K > 256) {
        for (int nc = 48; nc >= 16; nc -= 16)
            if (nc <= nc_max && N % nc == 0) return nc;
        return 0;
    }
    for (int nsplit = (N + nc_max - 1) / nc_max; nsplit <= N / 16; ++nsplit)
        if ((N / 16) % nsplit == 0) return N / nsplit;
    return 0;
}

template <int K, int ACT, bool HAS_LN, bool HAS_RES, int CT>
int launch_ct(const float* x, const float* gamma, const float* beta, float eps, const float* w, const float* bias,
              const float* res, float* out, long M, int N, hipStream_t st) {
    constexpr int RT = K <= 96 ? 2 : 1;
    constexpr bool FLY = K > 256;
    if (FLY && (HAS_LN || split_columns(N, K) / 16 != CT)) return SOC_EUNSUPPORTED;     // K13 takes those
    const void* fn = reinterpret_cast<const void*>(ws_linear_split_kernel<K, ACT, HAS_LN, HAS_RES, CT, RT, FLY>);
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int nc_per_split = split_columns(N, K);
    const int nsplit = (N + nc_per_split - 1) / nc_per_split;
    const int cus = num_cus(st);
    const long ngroups = (M + 16 * RT - 1) / (16 * RT);
    long per_split = cus / nsplit > 0 ? cus / nsplit : 1;
    if (per_split * 8 > ngroups) per_split = (ngroups + 7) / 8;       // never more waves than row groups
    const size_t lds = (size_t)nc_per_split * (6 * K + 4) + 8 * K;
    hipLaunchKernelGGL((ws_linear_split_kernel<K, ACT, HAS_LN, HAS_RES, CT, RT, FLY>), dim3((unsigned)(per_split * nsplit)),
                       dim3(THREADS), lds, st, x, gamma, beta, eps, w, bias, res, out, M, N, nc_per_split, (int)per_split);
    return soc_check_launch();
}

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
        if constexpr (K > 192) {          // no LayerNorm beyond K = 192 (K = 256: registers; K > 256: step-by-step form)
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

// K13b entry used by soc_ws_linear_f32 when the split arithmetic is switched on and the width is covered here
// (K = 96 / 128 / 192 with or without a LayerNorm in front: stages 0-1; K = 256 / 384 / 512 without one: the encoder's value_proj, the stage-0 fc2 layers and
// the stage-2 qkv / proj / fc1 layers behind K5's LayerNorm); SOC_EUNSUPPORTED sends the caller back to K13.
int soc_ws_linear_split_dispatch(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* w,
                                 const float* bias, const float* residual, float* out, long M, int N, int K, int act,
                                 hipStream_t st) {
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
