// K2: multi-scale deformable attention forward for gfx950.
//
// out[n,q,m,:] = sum_{l,p} w[n,q,m,l,p] * bilinear(value_l[n,:,m,:], loc[n,q,m,l,p])
// Sampling rules restate the reference CUDA kernel (models/ops/src/cuda/ms_deform_im2col_cuda.cuh
// :33-84 bilinear taps with per-tap zero padding, :237-299 per-output loop): pixel coordinates
// h = y*H - 0.5, w = x*W - 0.5; a point contributes only when -1 < h < H and -1 < w < W.
//
// MI355X mapping (HBM/L2-bound gather):
//   * fast path D == 32, P == 4, f32: 8 lanes x float4 cover one (query, head) row of 32
//     channels, so a 64-lane wave covers 8 heads of one query and every tap is one fully used
//     128-B line; all 16 taps of a level are in flight together;
//   * the block -> frame mapping is round-robin (frame = block % N), which is how the dispatcher
//     deals blocks to the 8 XCDs: with N = 8 frames each XCD's private 4 MiB L2 only ever sees
//     one frame's 4.9 MB value map (speed only, never correctness);
//   * the generic path (any D, f64) is one thread per output scalar.
#include "soc_common.h"

namespace {

template <typename T>
__device__ __forceinline__ T ldg(const T* p) { return *p; }

// ---------------------------------------------------------------------------------------------
// fast path: D = 32, P = 4, float.  One lane = 4 channels; per level the lane fetches the 4
// points' locations + weights with three 16-B loads, derives all 16 tap addresses (clamped into
// the map so every load is unconditional) and issues the 16 float4 tap loads back to back --
// 16 independent 128-B-line requests in flight per 8-lane group instead of a dependent
// load -> branch -> load chain per point.  Out-of-range taps keep their (clamped) load but get
// weight 0, which is the reference's zero padding.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
    a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Block -> (frame, chunk of the frame).  Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8) and each XCD
// has its own 4 MiB L2.  `b % N` (rounds 2-5) gives an XCD ONE frame when N = 8 (a clip), but with a launch group of ten clips
// (N = 80) it gave every XCD ten frames AT ONCE -- the blocks it has in flight then sample ten 4.9-MB value maps, and the PMC
// traffic per clip went from 415 MB (= algorithmic) to 711 MB.  When N is a multiple of 8 an XCD now walks its
// frames (x, x + 8, x + 16, ...) ONE AFTER THE OTHER: block b is chunk (b / 8) % C of frame (b % 8) + 8 ((b / 8) / C).
// Identical to b % N for N = 8; any other N keeps the old map.  Speed only: every (frame, chunk) is visited exactly once.
__device__ __forceinline__ void frame_chunk(const int b, const int N, const int C, int& n, int& chunk) {
    if ((N & 7) == 0) {
        const int i = b >> 3, k = i / C;
        n = (b & 7) + 8 * k;
        chunk = i - k * C;
    } else {
        n = b % N;
        chunk = b / N;
    }
}

__global__ __launch_bounds__(256, 4) void msda_fwd_d32p4_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, float* __restrict__ out, int N, int S, int M, int L, int Lq,
    int groups_per_frame /* = Lq*M */) {
    int n, chunk;
    frame_chunk(blockIdx.x, N, (int)(gridDim.x / (unsigned)N), n, chunk);      // XCD-friendly: see frame_chunk
    const int sub = threadIdx.x >> 3;    // 32 (query, head) groups per block
    const int c4 = threadIdx.x & 7;      // which float4 of the 32 channels
    const int g = chunk * 32 + sub;      // (q, m) flat index inside frame n
    if (g >= groups_per_frame) return;
    const int m = g % M;
    const long gi = (long)n * groups_per_frame + g;          // (n, q, m) flat
    const float4* lp = reinterpret_cast<const float4*>(loc + gi * (long)(L * 8));
    const float4* wp = reinterpret_cast<const float4*>(attw + gi * (long)(L * 4));
    // taps are buffer loads: <frame descriptor> + <scalar level offset> + <32-bit per-lane byte offset>, i.e. one
    // VALU add per tap instead of 64-bit pointer arithmetic (the host guarantees a frame's value map is < 2 GiB)
    const __amdgpu_buffer_rsrc_t vframe = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value + (long)n * S * M * 32), 0, S * M * 32 * 4, 0x00020000);
    const unsigned lane_off = (unsigned)(m * 32 + c4 * 4) * 4u;
    const unsigned rstride = (unsigned)M * 32u * 4u;  // bytes between consecutive spatial positions

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        const int lstart = (int)lsi[l];
        const unsigned vl = (unsigned)lstart * rstride;       // wave-uniform byte offset of the level
        const float4 la = lp[2 * l], lb = lp[2 * l + 1];
        const float4 wv = wp[l];
        float xs[4] = {la.x, la.z, lb.x, lb.z};
        float ys[4] = {la.y, la.w, lb.y, lb.w};
        const float ws[4] = {wv.x, wv.y, wv.z, wv.w};
        float tw[4][4];
        unsigned ptr[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float him = ys[p] * Hl - 0.5f;
            const float wim = xs[p] * Wl - 0.5f;
            const bool ok = him > -1.f && wim > -1.f && him < Hl && wim < Wl;
            const float hf = floorf(him), wf = floorf(wim);
            const float lh = him - hf, lw = wim - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            // clamp BEFORE the int conversion so wild locations cannot overflow
            const int h0 = (int)fminf(fmaxf(hf, -1.f), (float)Hl);
            const int w0 = (int)fminf(fmaxf(wf, -1.f), (float)Wl);
            const bool h0ok = ok && h0 >= 0, h1ok = ok && h0 + 1 <= Hl - 1;
            const bool w0ok = w0 >= 0, w1ok = w0 + 1 <= Wl - 1;
            const int h0c = min(max(h0, 0), Hl - 1), h1c = min(max(h0 + 1, 0), Hl - 1);
            const int w0c = min(max(w0, 0), Wl - 1), w1c = min(max(w0 + 1, 0), Wl - 1);
            const float wgt = ws[p];
            tw[p][0] = (h0ok && w0ok) ? hh * hw * wgt : 0.f;
            tw[p][1] = (h0ok && w1ok) ? hh * lw * wgt : 0.f;
            tw[p][2] = (h1ok && w0ok) ? lh * hw * wgt : 0.f;
            tw[p][3] = (h1ok && w1ok) ? lh * lw * wgt : 0.f;
            // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): positions and the row stride are < 2^24
            const unsigned r0 = __umul24(__umul24(h0c, Wl), rstride) + lane_off;
            const unsigned r1 = __umul24(__umul24(h1c, Wl), rstride) + lane_off;
            const unsigned c0 = __umul24(w0c, rstride), c1 = __umul24(w1c, rstride);
            ptr[p][0] = r0 + c0;
            ptr[p][1] = r0 + c1;
            ptr[p][2] = r1 + c0;
            ptr[p][3] = r1 + c1;
        }
        float4 tv[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(vframe, ptr[p][k], vl, 0);
                tv[p][k] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                                       __uint_as_float(raw.w));
            }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) fma4(acc, tw[p][k], tv[p][k]);
    }
    *reinterpret_cast<float4*>(out + gi * 32 + c4 * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// fused path with the sampling-point tiles staged in LDS (round 2).  In the kernel above the 8 lanes that share a
// (query, head) row each repeat the whole per-point arithmetic -- softmax, location, floor, four bilinear weights, four
// clamped addresses: ~40 vector instructions x 16 points per lane -- and the counters (profiles/r02_k2_counters.json)
// show the launch vector-issue-bound: 54.9 M vector instructions, 93 us of VALU time in a 131-us kernel.  Here every lane
// works out TWO of its group's 16 points once (phase 1: lane c of the group takes points 2c, 2c+1, the softmax runs over
// the 8 lanes with DPP shuffles) and writes (4 byte offsets, 4 weights) per point to LDS; phase 2 reads them back as two
// broadcast ds_read_b128 per point (the 8 groups of a wave sit 136 dwords apart: conflict-free) and is the plain gather:
// 4 buffer loads + 16 FMAs per point.  A group's producers and consumers are the same 8 lanes of one wave, so no
// workgroup barrier is needed.  Same per-point formulas as above; the softmax sums in another order.
// ---------------------------------------------------------------------------------------------
constexpr int TILE_GROUP_STRIDE = 16 * 8 + 8;     // dwords per (query, head) group: 16 points x (4 offsets + 4 weights) + pad

__global__ __launch_bounds__(256, 4) void msda_fused_tiles_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
    const float* __restrict__ offs, const float* __restrict__ logits, float* __restrict__ out, int N, int S, int M,
    int Lq, int groups_per_frame, const float* __restrict__ ref, int ref_dim, const uint8_t* __restrict__ pad,
    const int* __restrict__ any_pad) {
    __shared__ __attribute__((aligned(16))) unsigned tile[32 * TILE_GROUP_STRIDE];
    int n, chunk;
    frame_chunk(blockIdx.x, N, (int)(gridDim.x / (unsigned)N), n, chunk);      // XCD-friendly: see frame_chunk
    const int sub = threadIdx.x >> 3;    // 32 (query, head) groups per block
    const int c4 = threadIdx.x & 7;      // phase 1: points 2*c4, 2*c4+1; phase 2: which float4 of the 32 channels
    // a block = 32 consecutive queries of ONE head, so a wave = 8 raster-adjacent queries whose taps share L1 lines
    // (98.5 vs 101.6 us for the 8-heads-of-one-query order of the kernel above).  The head is the SLOW index inside a
    // frame: the blocks an XCD has in flight (its frame's, in chunk order) then sample one or two heads' 617-KB value
    // planes at a time, which its 4 MiB L2 holds -- with the head fastest they cycled through all 8 heads of the 4.9 MB
    // map (PMC traffic 198 MB per encoder call against 138 MB algorithmic, L2 hit rate 0.90 in round 2)
    const int nqb = (Lq + 31) >> 5;
    const int m = chunk / nqb;
    const int q_ = (chunk - m * nqb) * 32 + sub;
    const bool live = q_ < Lq;                                // queries past the end recompute the last one
    const int g = min(q_, Lq - 1) * M + m;
    const long gi = (long)n * groups_per_frame + g;          // (n, q, m) flat
    const unsigned rstride = (unsigned)M * 32u * 4u;         // bytes between consecutive spatial positions

    // ---- phase 1: two points of the group per lane
    {
        const float2 lg = reinterpret_cast<const float2*>(logits + gi * 16)[c4];
        float mx = fmaxf(lg.x, lg.y);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e0 = __expf(lg.x - mx), e1 = __expf(lg.y - mx);
        float sum = e0 + e1;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
        const float inv = __builtin_amdgcn_rcpf(sum);      // 1 ulp: the weights are a convex combination either way
        const float wsm[2] = {e0 * inv, e1 * inv};
        const float4 of = reinterpret_cast<const float4*>(offs + gi * 32)[c4];      // (x, y) of the two points
        const int l = c4 >> 1;
        const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
        const int lstart = (int)lsi[l];
        const float* rp = ref + (((long)n * Lq + g / M) * 4 + l) * ref_dim;
        float xs[2] = {of.x, of.z}, ys[2] = {of.y, of.w};
        if (ref_dim == 2) {
            const float rW = 1.0f / (float)Wl, rH = 1.0f / (float)Hl;
#pragma unroll
            for (int j = 0; j < 2; ++j) { xs[j] = rp[0] + xs[j] * rW; ys[j] = rp[1] + ys[j] * rH; }
        } else {
            const float rw = rp[2], rh = rp[3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {      // / 4 is exact (power of two)
                xs[j] = rp[0] + xs[j] * 0.25f * rw * 0.5f;
                ys[j] = rp[1] + ys[j] * 0.25f * rh * 0.5f;
            }
        }
        const bool use_pad = pad != nullptr && any_pad != nullptr && *any_pad != 0;
        const uint8_t* padn = pad + (long)n * S;
        unsigned* dst = tile + sub * TILE_GROUP_STRIDE + (2 * c4) * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float him = ys[j] * Hl - 0.5f;
            const float wim = xs[j] * Wl - 0.5f;
            const bool ok = him > -1.f && wim > -1.f && him < Hl && wim < Wl;
            const float hf = floorf(him), wf = floorf(wim);
            const float lh = him - hf, lw = wim - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const int h0 = (int)fminf(fmaxf(hf, -1.f), (float)Hl);
            const int w0 = (int)fminf(fmaxf(wf, -1.f), (float)Wl);
            const bool h0ok = ok && h0 >= 0, h1ok = ok && h0 + 1 <= Hl - 1;
            const bool w0ok = w0 >= 0, w1ok = w0 + 1 <= Wl - 1;
            const int h0c = min(max(h0, 0), Hl - 1), h1c = min(max(h0 + 1, 0), Hl - 1);
            const int w0c = min(max(w0, 0), Wl - 1), w1c = min(max(w0 + 1, 0), Wl - 1);
            const float wgt = wsm[j];
            float tw[4] = {(h0ok && w0ok) ? hh * hw * wgt : 0.f, (h0ok && w1ok) ? hh * lw * wgt : 0.f,
                           (h1ok && w0ok) ? lh * hw * wgt : 0.f, (h1ok && w1ok) ? lh * lw * wgt : 0.f};
            const int pos[4] = {lstart + h0c * Wl + w0c, lstart + h0c * Wl + w1c, lstart + h1c * Wl + w0c,
                                lstart + h1c * Wl + w1c};
            if (use_pad) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (padn[pos[k]]) tw[k] = 0.f;
            }
            const unsigned head_off = (unsigned)m * 128u;
            *reinterpret_cast<u32x4*>(dst + j * 8) = (u32x4){__umul24(pos[0], rstride) + head_off, __umul24(pos[1], rstride) + head_off,
                                                            __umul24(pos[2], rstride) + head_off, __umul24(pos[3], rstride) + head_off};
            *reinterpret_cast<float4*>(dst + j * 8 + 4) = make_float4(tw[0], tw[1], tw[2], tw[3]);
        }
    }
    // producers and consumers of a group's tile are the same 8 lanes of one wave: order the LDS traffic, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase 2: the gather.  Taps are buffer loads: <frame descriptor> + <32-bit per-lane byte offset>
    const __amdgpu_buffer_rsrc_t vframe = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value + (long)n * S * M * 32), 0, S * M * 32 * 4, 0x00020000);
    const unsigned lane_off = (unsigned)c4 * 16u;
    const unsigned* src = tile + sub * TILE_GROUP_STRIDE;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        u32x4 po[4];
        float4 tw[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            po[p] = *reinterpret_cast<const u32x4*>(src + (l * 4 + p) * 8);
            tw[p] = *reinterpret_cast<const float4*>(src + (l * 4 + p) * 8 + 4);
        }
        float4 tv[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(vframe, po[p][k] + lane_off, 0, 0);
                tv[p][k] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z),
                                       __uint_as_float(raw.w));
            }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            fma4(acc, tw[p].x, tv[p][0]);
            fma4(acc, tw[p].y, tv[p][1]);
            fma4(acc, tw[p].z, tv[p][2]);
            fma4(acc, tw[p].w, tv[p][3]);
        }
    }
    if (live) *reinterpret_cast<float4*>(out + gi * 32 + c4 * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// generic path: one thread per output scalar (any D, float or double)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const T* __restrict__ loc, const T* __restrict__ attw,
    T* __restrict__ out, long total, int S, int M, int D, int L, int Lq, int P) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long gi = idx / D;  // (n, q, m)
        const int m = (int)(gi % M);
        const long n = gi / ((long)M * Lq);
        const T* lp = loc + gi * (long)(L * P * 2);
        const T* wp = attw + gi * (long)(L * P);
        const long rstride = (long)M * D;
        const T* vb = value + n * S * rstride + (long)m * D + c;
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const T* vl = vb + (long)lsi[l] * rstride;
            for (int p = 0; p < P; ++p) {
                const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
                const T wgt = wp[l * P + p];
                const T him = y * Hl - (T)0.5, wim = x * Wl - (T)0.5;
                if (him > -1 && wim > -1 && him < Hl && wim < Wl) {
                    const int h0 = (int)floor(him), w0 = (int)floor(wim);
                    const T lh = him - h0, lw = wim - w0, hh = 1 - lh, hw = 1 - lw;
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    const T* r0 = vl + ((long)h0 * Wl + w0) * rstride;
                    if (h0 >= 0 && w0 >= 0) v1 = r0[0];
                    if (h0 >= 0 && w0 + 1 <= Wl - 1) v2 = r0[rstride];
                    if (h0 + 1 <= Hl - 1 && w0 >= 0) v3 = r0[(long)Wl * rstride];
                    if (h0 + 1 <= Hl - 1 && w0 + 1 <= Wl - 1) v4 = r0[(long)Wl * rstride + rstride];
                    acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * wgt;
                }
            }
        }
        out[idx] = acc;
    }
}

template <typename T>
int launch_generic(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,
                   const T* attw, T* out, int N, int S, int M, int D, int L, int Lq, int P,
                   hipStream_t st) {
    const long total = (long)N * Lq * M * D;
    if (total == 0) return SOC_OK;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(msda_fwd_generic_kernel<T>, dim3(blocks), dim3(256), 0, st, value, shapes,
                       lsi, loc, attw, out, total, S, M, D, L, Lq, P);
    return soc_check_launch();
}

bool bad_args(const void* a, const void* b, const void* c, const void* d, const void* e,
              const void* f, int N, int S, int M, int D, int L, int Lq, int P) {
    return !a || !b || !c || !d || !e || !f || N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 ||
           Lq < 0 || P <= 0;
}

}  // namespace

extern "C" int soc_msda_fwd_f32(const float* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const float* sampling_loc,
                                const float* attn_weight, float* out, int N, int S, int M, int D,
                                int L, int Lq, int P, void* stream) {
    if (N == 0 || Lq == 0) return (N < 0 || Lq < 0) ? SOC_EINVAL : SOC_OK;
    if (bad_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S,
                 M, D, L, Lq, P))
        return SOC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (D == 32 && P == 4 && (long)S * M * 128 < (1L << 31) && S < (1 << 24) && M * 128 < (1 << 24)) {   // 32-bit tap offsets
        const int gpf = Lq * M;
        const int bpf = soc_ceil_div(gpf, 32);
        hipLaunchKernelGGL(msda_fwd_d32p4_kernel, dim3(bpf * N), dim3(256), 0, st, value,
                           spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N,
                           S, M, L, Lq, gpf);
        return soc_check_launch();
    }
    return launch_generic<float>(value, spatial_shapes, level_start_index, sampling_loc,
                                 attn_weight, out, N, S, M, D, L, Lq, P, st);
}

extern "C" int soc_msda_fwd_f64(const double* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const double* sampling_loc,
                                const double* attn_weight, double* out, int N, int S, int M,
                                int D, int L, int Lq, int P, void* stream) {
    if (N == 0 || Lq == 0) return (N < 0 || Lq < 0) ? SOC_EINVAL : SOC_OK;
    if (bad_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S,
                 M, D, L, Lq, P))
        return SOC_EINVAL;
    return launch_generic<double>(value, spatial_shapes, level_start_index, sampling_loc,
                                  attn_weight, out, N, S, M, D, L, Lq, P, (hipStream_t)stream);
}

extern "C" int soc_msda_fused_fwd_f32(const float* value, const uint8_t* value_pad_mask,
                                      const int32_t* any_pad, const int64_t* spatial_shapes,
                                      const int64_t* level_start_index, const float* ref_points,
                                      int ref_dim, const float* offsets, const float* attn_logits,
                                      float* out, int N, int S, int M, int D, int L, int Lq, int P,
                                      void* stream) {
    if (N == 0 || Lq == 0) return (N < 0 || Lq < 0) ? SOC_EINVAL : SOC_OK;
    if (bad_args(value, spatial_shapes, level_start_index, offsets, attn_logits, out, N, S, M, D, L, Lq, P) ||
        !ref_points)
        return SOC_EINVAL;
    if (D != 32 || P != 4 || L != 4 || (ref_dim != 2 && ref_dim != 4)) return SOC_EUNSUPPORTED;
    if ((long)S * M * 128 >= (1L << 31) || S >= (1 << 24) || M * 128 >= (1 << 24))
        return SOC_EUNSUPPORTED;   // 32-bit tap offsets inside a frame, 24-bit multiplies
    if ((value_pad_mask == nullptr) != (any_pad == nullptr)) return SOC_EINVAL;
    const int gpf = Lq * M;
    const int bpf = soc_ceil_div(Lq, 32) * M;
    hipLaunchKernelGGL(msda_fused_tiles_kernel, dim3(bpf * N), dim3(256), 0, (hipStream_t)stream, value, spatial_shapes,
                       level_start_index, offsets, attn_logits, out, N, S, M, Lq, gpf, ref_points, ref_dim, value_pad_mask,
                       (const int*)any_pad);
    return soc_check_launch();
}
