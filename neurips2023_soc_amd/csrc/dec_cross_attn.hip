// K15: the cross-attention block of a deformable-decoder layer in ONE launch.
//
//   tgt' = LayerNorm(tgt + output_proj(MSDeformAttn(tgt + query_pos, ref, value_proj(memory))))
//                                        reference models/deformable_transformer.py:335-341,
//                                        models/ops/modules/ms_deform_attn.py:79-117
//
// The reference (and K2) project the WHOLE memory (38 560 x 256 x 256 per layer at the BASELINE config: a chip-filling
// GEMM) although a decoder layer samples only Lq * M * L * P = 2 560 points per frame.  value_proj is linear, so here the
// 256-wide memory rows are sampled first and the per-head slice of value_proj is applied to the 8 x 256 sampled sums:
//
//   sum_taps c_t * (mem[row_t] Wv_m^T + bv_m)  =  (sum_taps c_t * mem[row_t]) Wv_m^T + bv_m * sum_taps c_t
//
// (taps outside the map or on padded positions have c_t = 0 on both sides).  Same arithmetic up to the order of the f32
// sums.  One workgroup per (frame, query) row, 16 waves:
//   A  offsets / attention logits = (tgt + pos) [W_off; W_att]^T + b        384 dot products of length 256
//   B  softmax over the 16 logits of a head, sampling locations, 4 bilinear taps each -> 8 x 64 (row, coefficient) in LDS
//      (the location / tap arithmetic is K2's, csrc/msda_fwd.hip)
//   C  wave (m, half) sums its 32 taps of the 1-KB memory rows, one float4 per lane
//   D  v[m*32+d] = S_m . Wv[m*32+d] + bv * sum(c)                            256 dot products
//   E  o = v Wo^T + bo, LayerNorm(tgt + o)                                   256 dot products + one row reduction
// A dot-product phase gives each wave 16 output rows: 16 coalesced 1-KB weight-row loads in flight per wave, the 16 x 64
// partial products reduced by a transposing butterfly (17 cross-lane moves instead of 96).
#include "soc_common.h"
#include "row_ops.h"

namespace {

constexpr int DM = ROW_DM;    // d_model
constexpr int NH = 8;         // heads
constexpr int NLV = 4;        // levels
constexpr int NPT = 4;        // points
// 16 waves x 16 weight rows in flight.  A 4-wave / 8-row build with <= 80 registers per lane (workgroups that fit beside
// another clip's chip-filling kernels) measured the same inside the software pipeline (8.76 vs 8.78 ms per clip over
// 3 x 150 clips, tools/experiments/README.md) and is not built.

struct DecXArgs {
    const float* tgt;        // [N, Lq, 256]
    const float* qpos;       // [N or 1, Lq, 256]
    long qpos_nstride;       // Lq*256, or 0 when one [Lq,256] embedding serves every frame
    const float* ref;        // [N, Lq, 4, ref_dim]
    int ref_dim;
    const float* memory;     // [N, S, 256]
    const uint8_t* pad;      // [N, S] or null
    const int* any_pad;      // int32[1] or null
    const int64_t* shapes;   // [4, 2] (H, W)
    const int64_t* lsi;      // [4]
    const float *w_off, *b_off;   // [256, 256], [256]   sampling_offsets
    const float *w_att, *b_att;   // [128, 256], [128]   attention_weights
    const float *w_val, *b_val;   // [256, 256], [256]   value_proj
    const float *w_out, *b_out;   // [256, 256], [256]   output_proj
    const float *gamma, *beta;    // [256]               norm1
    float eps;
    float* out;              // [N, Lq, 256]
    int N, Lq, S;
};

template <int THREADS, int R>
__global__ __launch_bounds__(THREADS) void dec_cross_attn_kernel(const DecXArgs a) {
    constexpr int NWAVES = THREADS / 64;
    __shared__ __attribute__((aligned(16))) float qv[DM];            // tgt + pos
    __shared__ __attribute__((aligned(16))) float offlog[DM + 128];  // 256 raw offsets, then 128 raw logits
    __shared__ float tcoef[NH][64];
    __shared__ int trow[NH][64];
    __shared__ __attribute__((aligned(16))) float part[2][NH][DM];   // half-sums of the sampled rows
    __shared__ float csum[NH];
    __shared__ __attribute__((aligned(16))) float vproj[DM];
    __shared__ __attribute__((aligned(16))) float oproj[DM];
    __shared__ float red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x % a.N, q = blockIdx.x / a.N;        // frames of one query on neighbouring workgroups
    const long row = (long)n * a.Lq + q;
    const float* trow_in = a.tgt + row * DM;

    if (tid < DM) qv[tid] = trow_in[tid] + a.qpos[(long)n * a.qpos_nstride + (long)q * DM + tid];
    __syncthreads();

    // ---- A: 384 outputs = 24 groups of 16 rows over 16 waves
    {
        const float4 x4 = reinterpret_cast<const float4*>(qv)[lane];
        for (int g = wave; g < 384 / R; g += NWAVES) {
            if (g < 256 / R) matvec_rows<R>(a.w_off, a.b_off, 1.f, g * R, 256, x4, offlog, lane, 0);
            else matvec_rows<R>(a.w_att, a.b_att, 1.f, g * R - 256, 128, x4, offlog + DM, lane, 0);
        }
    }
    __syncthreads();

    // ---- B: thread t < 128 = (head m, level l, point p): softmax weight, location, four taps
    if (tid < NH * NLV * NPT) {
        const int m = tid >> 4, l = (tid >> 2) & 3, p = tid & 3;
        const float logit = offlog[DM + tid];
        float mx = logit;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = __expf(logit - mx);
        float sum = e;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float wgt = e * (1.f / sum);
        const int Hl = (int)a.shapes[2 * l], Wl = (int)a.shapes[2 * l + 1];
        const int lstart = (int)a.lsi[l];
        float xs = offlog[2 * tid], ys = offlog[2 * tid + 1];
        const float* rp = a.ref + (row * NLV + l) * a.ref_dim;
        if (a.ref_dim == 2) {
            const float rW = 1.0f / (float)Wl, rH = 1.0f / (float)Hl;
            xs = rp[0] + xs * rW;
            ys = rp[1] + ys * rH;
        } else {
            xs = rp[0] + xs * 0.25f * rp[2] * 0.5f;
            ys = rp[1] + ys * 0.25f * rp[3] * 0.5f;
        }
        const float him = ys * Hl - 0.5f;
        const float wim = xs * Wl - 0.5f;
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
        float tw[4] = {(h0ok && w0ok) ? hh * hw * wgt : 0.f, (h0ok && w1ok) ? hh * lw * wgt : 0.f,
                       (h1ok && w0ok) ? lh * hw * wgt : 0.f, (h1ok && w1ok) ? lh * lw * wgt : 0.f};
        const int rr[4] = {lstart + h0c * Wl + w0c, lstart + h0c * Wl + w1c, lstart + h1c * Wl + w0c,
                           lstart + h1c * Wl + w1c};
        if (a.pad != nullptr && a.any_pad != nullptr && *a.any_pad != 0) {
            const uint8_t* padn = a.pad + (long)n * a.S;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (padn[rr[k]]) tw[k] = 0.f;
        }
        const int slot = (l * NPT + p) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tcoef[m][slot + k] = tw[k];
            trow[m][slot + k] = rr[k];
        }
        // sum of the head's 64 coefficients (the weight of value_proj's bias)
        float cs = (tw[0] + tw[1]) + (tw[2] + tw[3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
        if ((tid & 15) == 0) csum[m] = cs;
    }
    __syncthreads();

    // ---- C: unit (m, half): 32 taps of 1-KB memory rows, one float4 per lane; 16 units over the waves
    for (int u = wave; u < 16; u += NWAVES) {
        const int m = u >> 1, half = u & 1;
        const float4* mem = reinterpret_cast<const float4*>(a.memory + (long)n * a.S * DM) + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll R
        for (int t = 0; t < 32; ++t) {
            const float c = tcoef[m][half * 32 + t];
            const float4 r = mem[(long)trow[m][half * 32 + t] * (DM / 4)];
            acc.x += c * r.x; acc.y += c * r.y; acc.z += c * r.z; acc.w += c * r.w;
        }
        reinterpret_cast<float4*>(part[half][m])[lane] = acc;
    }
    __syncthreads();

    // ---- D: value_proj rows j0 .. j0 + R - 1 belong to head j0 / 32
    for (int g = wave; g < 256 / R; g += NWAVES) {
        const int m = (g * R) >> 5;
        const float4 s0 = reinterpret_cast<const float4*>(part[0][m])[lane];
        const float4 s1 = reinterpret_cast<const float4*>(part[1][m])[lane];
        const float4 x4 = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
        matvec_rows<R>(a.w_val, a.b_val, csum[m], g * R, 256, x4, vproj, lane, 0);
    }
    __syncthreads();

    // ---- E: output_proj, residual, LayerNorm
    {
        const float4 x4 = reinterpret_cast<const float4*>(vproj)[lane];
        for (int g = wave; g < 256 / R; g += NWAVES) matvec_rows<R>(a.w_out, a.b_out, 1.f, g * R, 256, x4, oproj, lane, 0);
    }
    __syncthreads();
    float mean = 0.f, x = 0.f;
    if (tid < DM) {      // waves 0-3 hold the row
        x = trow_in[tid] + oproj[tid];
        float s = x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
    }
    __syncthreads();
    if (tid < DM) {
        mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.f / DM);
        const float dx = x - mean;
        float v = dx * dx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[4 + wave] = v;
    }
    __syncthreads();
    if (tid < DM) {
        const float var = ((red[4] + red[5]) + (red[6] + red[7])) * (1.f / DM);
        const float rstd = rsqrtf(var + a.eps);
        a.out[row * DM + tid] = (x - mean) * rstd * a.gamma[tid] + a.beta[tid];
    }
}

}  // namespace

extern "C" int soc_decoder_cross_attn_f32(const float* tgt, const float* query_pos, int query_pos_per_frame,
                                          const float* ref_points, int ref_dim, const float* memory,
                                          const uint8_t* memory_pad_mask, const int32_t* any_pad,
                                          const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* w_off, const float* b_off, const float* w_att, const float* b_att,
                                          const float* w_val, const float* b_val, const float* w_out, const float* b_out,
                                          const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int N,
                                          int Lq, int S, int d_model, int n_heads, int n_levels, int n_points,
                                          void* stream) {
    if (N < 0 || Lq < 0 || S <= 0) return SOC_EINVAL;
    if (N == 0 || Lq == 0) return SOC_OK;
    if (!tgt || !query_pos || !ref_points || !memory || !spatial_shapes || !level_start_index || !w_off || !b_off ||
        !w_att || !b_att || !w_val || !b_val || !w_out || !b_out || !ln_gamma || !ln_beta || !out)
        return SOC_EINVAL;
    if ((memory_pad_mask == nullptr) != (any_pad == nullptr)) return SOC_EINVAL;
    if (d_model != DM || n_heads != NH || n_levels != NLV || n_points != NPT || (ref_dim != 2 && ref_dim != 4))
        return SOC_EUNSUPPORTED;
    if ((long)S * DM >= (1L << 31)) return SOC_EUNSUPPORTED;
    DecXArgs a;
    a.tgt = tgt; a.qpos = query_pos; a.qpos_nstride = query_pos_per_frame ? (long)Lq * DM : 0;
    a.ref = ref_points; a.ref_dim = ref_dim; a.memory = memory; a.pad = memory_pad_mask; a.any_pad = any_pad;
    a.shapes = spatial_shapes; a.lsi = level_start_index;
    a.w_off = w_off; a.b_off = b_off; a.w_att = w_att; a.b_att = b_att; a.w_val = w_val; a.b_val = b_val;
    a.w_out = w_out; a.b_out = b_out; a.gamma = ln_gamma; a.beta = ln_beta; a.eps = ln_eps; a.out = out;
    a.N = N; a.Lq = Lq; a.S = S;
    hipLaunchKernelGGL((dec_cross_attn_kernel<1024, 16>), dim3(N * Lq), dim3(1024), 0, (hipStream_t)stream, a);
    return soc_check_launch();
}
