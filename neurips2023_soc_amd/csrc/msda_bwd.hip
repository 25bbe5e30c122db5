// K2 backward: gradients of multi-scale deformable attention for gfx950 (SURVEY 8f rank 4 -- the
// training-side half of the reference's native op, models/ops/src/vision.cpp:13-16,
// src/cuda/ms_deform_attn_cuda.cu:83-153, kernels src/cuda/ms_deform_im2col_cuda.cuh:301-921).
//
// With h = y*H - 0.5, w = x*W - 0.5, lh = h - floor(h), lw = w - floor(w), taps v1..v4 at
// (h0,w0) (h0,w0+1) (h0+1,w0) (h0+1,w0+1), each zero outside the map, and a point contributing only
// when -1 < h < H and -1 < w < W (the forward's rules):
//   val            = (1-lh)(1-lw) v1 + (1-lh) lw v2 + lh (1-lw) v3 + lh lw v4
//   grad_attn      = sum_d go[d] val[d]
//   grad_loc.x     = W * attn * sum_d go[d] (-(1-lh) v1 + (1-lh) v2 - lh v3 + lh v4)
//   grad_loc.y     = H * attn * sum_d go[d] (-(1-lw) v1 - lw v2 + (1-lw) v3 + lw v4)
//   grad_value[tap] += tap_weight * attn * go            (atomic: many queries hit one pixel)
//
// Mapping: LPI lanes (32 when D <= 32, else 64) own one (n, query, head); the lanes stride over the D
// channels, so the 4 tap loads and the 4 atomic adds of a sample are contiguous runs of D floats;
// the three per-sample sums are reduced across the LPI lanes with DPP shuffles and written once.
// The reference maps one thread per channel and reduces through shared memory, with a separate
// kernel per channel-count class; one kernel covers every D here.  grad_value is zeroed first.
#include "soc_common.h"
#include <math.h>

namespace {

template <typename T>
__device__ __forceinline__ void atomic_add(T* p, T v) { unsafeAtomicAdd(p, v); }

template <typename T, int LPI>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const T* __restrict__ loc, const T* __restrict__ attw, const T* __restrict__ gout,
    T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gattw, long items, int S, int M, int D,
    int L, int Lq, int P) {
    const long item = ((long)blockIdx.x * 256 + threadIdx.x) / LPI;     // (n, q, m) flat
    const int sub = threadIdx.x % LPI;
    if (item >= items) return;                                           // whole lane group leaves together
    const int m = (int)(item % M);
    const long n = item / ((long)M * Lq);
    const T* go = gout + item * D;                                       // grad_out [N, Lq, M*D]
    const long vrow = (long)M * D;                                       // elements between spatial positions
    const T* vbase = value + n * S * vrow + (long)m * D;
    T* gvbase = gvalue + n * S * vrow + (long)m * D;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long start = lsi[l];
        for (int p = 0; p < P; ++p) {
            const long sidx = (item * L + l) * P + p;
            const T x = loc[2 * sidx], y = loc[2 * sidx + 1], a = attw[sidx];
            const T h_im = y * (T)H - (T)0.5, w_im = x * (T)W - (T)0.5;
            T s_a = 0, s_w = 0, s_h = 0;
            if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
                const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
                const T lh = h_im - (T)h0, lw = w_im - (T)w0, hh = (T)1 - lh, hw = (T)1 - lw;
                const bool top = h0 >= 0, bot = h0 + 1 <= H - 1, lef = w0 >= 0, rig = w0 + 1 <= W - 1;
                const long o1 = (start + (long)h0 * W + w0) * vrow;
                const long o2 = o1 + vrow, o3 = o1 + (long)W * vrow, o4 = o3 + vrow;
                const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                for (int d = sub; d < D; d += LPI) {
                    const T g = go[d];
                    const T v1 = (top && lef) ? vbase[o1 + d] : (T)0;
                    const T v2 = (top && rig) ? vbase[o2 + d] : (T)0;
                    const T v3 = (bot && lef) ? vbase[o3 + d] : (T)0;
                    const T v4 = (bot && rig) ? vbase[o4 + d] : (T)0;
                    s_a += g * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    s_w += g * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
                    s_h += g * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
                    const T ga = g * a;
                    if (top && lef) atomic_add(gvbase + o1 + d, w1 * ga);
                    if (top && rig) atomic_add(gvbase + o2 + d, w2 * ga);
                    if (bot && lef) atomic_add(gvbase + o3 + d, w3 * ga);
                    if (bot && rig) atomic_add(gvbase + o4 + d, w4 * ga);
                }
            }
#pragma unroll
            for (int o = LPI / 2; o > 0; o >>= 1) {
                s_a += __shfl_xor(s_a, o);
                s_w += __shfl_xor(s_w, o);
                s_h += __shfl_xor(s_h, o);
            }
            if (sub == 0) {
                gattw[sidx] = s_a;
                gloc[2 * sidx] = (T)W * a * s_w;
                gloc[2 * sidx + 1] = (T)H * a * s_h;
            }
        }
    }
}

template <typename T>
int launch_bwd(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* attw,
               const T* gout, T* gvalue, T* gloc, T* gattw, int N, int S, int M, int D, int L, int Lq, int P,
               hipStream_t st) {
    if (N < 0 || Lq < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || P <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!value || !shapes || !lsi || !gvalue) return SOC_EINVAL;
    if (hipMemsetAsync(gvalue, 0, (size_t)N * S * M * D * sizeof(T), st) != hipSuccess) return SOC_ELAUNCH;
    if (Lq == 0) return SOC_OK;
    if (!loc || !attw || !gout || !gloc || !gattw) return SOC_EINVAL;
    const long items = (long)N * Lq * M;
    if (D <= 32) {
        const long threads = items * 32;
        hipLaunchKernelGGL((msda_bwd_kernel<T, 32>), dim3(soc_ceil_div(threads, 256)), dim3(256), 0, st, value, shapes,
                           lsi, loc, attw, gout, gvalue, gloc, gattw, items, S, M, D, L, Lq, P);
    } else {
        const long threads = items * 64;
        hipLaunchKernelGGL((msda_bwd_kernel<T, 64>), dim3(soc_ceil_div(threads, 256)), dim3(256), 0, st, value, shapes,
                           lsi, loc, attw, gout, gvalue, gloc, gattw, items, S, M, D, L, Lq, P);
    }
    return soc_check_launch();
}

}  // namespace

extern "C" int soc_msda_bwd_f32(const float* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const float* sampling_loc,
                                const float* attn_weight, const float* grad_out, float* grad_value,
                                float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M, int D,
                                int L, int Lq, int P, void* stream) {
    return launch_bwd<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out,
                             grad_value, grad_sampling_loc, grad_attn_weight, N, S, M, D, L, Lq, P,
                             (hipStream_t)stream);
}

extern "C" int soc_msda_bwd_f64(const double* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const double* sampling_loc,
                                const double* attn_weight, const double* grad_out, double* grad_value,
                                double* grad_sampling_loc, double* grad_attn_weight, int N, int S, int M, int D,
                                int L, int Lq, int P, void* stream) {
    return launch_bwd<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out,
                              grad_value, grad_sampling_loc, grad_attn_weight, N, S, M, D, L, Lq, P,
                              (hipStream_t)stream);
}
