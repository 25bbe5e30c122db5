// K25: multi-head self-attention core for short sequences (L <= 64 tokens), one workgroup per (batch, head).
//
// Replaces torch.nn.functional.scaled_dot_product_attention as HuggingFace's RobertaSelfAttention calls it for the text encoder
// SOC.forward_text runs per clip (reference models/soc.py:167-181: RobertaModel on the tokenised expression; 12 layers x 12 heads
// x 64 dims, 10-32 tokens).  PyTorch dispatches that call to its AOTriton `attn_fwd` kernel -- the one compiled-by-Triton kernel
// that was left in the timed graph (VERDICT r3 "missing" #6).  The problem is tiny and latency-bound (12 launches per clip on the
// text branch, beside Video-Swin): a wave per (batch, head); K and V of the head sit in LDS ([L][D + 4] floats), a lane owns a
// query row (q in registers), scores go through an LDS row per lane (two-pass softmax, as the reference's), P.V accumulates in
// registers.  f32 throughout.  q / k / v / out are token-major [B, L, H * D] (what the projections produce: no transposes), the
// mask is additive ([B, 1, Lq | 1, Lk] with the given strides, -inf on padding) or NULL.
#include "soc_common.h"

namespace {

template <int D>
__global__ __launch_bounds__(64) void small_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ mask,
                                                        float* __restrict__ out, int L, int H, float scale, long mask_b,
                                                        long mask_q) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RS = D + 4;                       // row stride: 16-B aligned, rows of one matrix on different banks
    float* Ks = lds;                                // [L][RS]
    float* Vs = Ks + L * RS;                        // [L][RS]
    float* Ss = Vs + L * RS;                        // [64][L + 1] scores, a row per lane
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const long E = (long)H * D;
    const float* kb = k + (long)b * L * E + (long)h * D;
    const float* vb = v + (long)b * L * E + (long)h * D;
    for (int i = lane; i < L * (D / 4); i += 64) {
        const int row = i / (D / 4), c = i % (D / 4);
        *reinterpret_cast<float4*>(Ks + row * RS + 4 * c) = *reinterpret_cast<const float4*>(kb + (long)row * E + 4 * c);
        *reinterpret_cast<float4*>(Vs + row * RS + 4 * c) = *reinterpret_cast<const float4*>(vb + (long)row * E + 4 * c);
    }
    __syncthreads();
    if (lane >= L) return;
    float qr[D];
    const float* qp = q + ((long)b * L + lane) * E + (long)h * D;
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * c);
        qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
    }
    float* srow = Ss + lane * (L + 1);
    const float* mrow = mask ? mask + (long)b * mask_b + (long)lane * mask_q : nullptr;
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) {
        const float* kr = Ks + j * RS;              // the same address in every lane: broadcast reads
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 kk = *reinterpret_cast<const float4*>(kr + c);
            s0 = fmaf(qr[c], kk.x, s0); s1 = fmaf(qr[c + 1], kk.y, s1); s2 = fmaf(qr[c + 2], kk.z, s2); s3 = fmaf(qr[c + 3], kk.w, s3);
        }
        float s = ((s0 + s1) + (s2 + s3)) * scale;
        if (mrow) s += mrow[j];
        srow[j] = s;
        mx = fmaxf(mx, s);
    }
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    float sum = 0.f;
    for (int j = 0; j < L; ++j) {
        const float p = mx == -INFINITY ? 0.f : __expf(srow[j] - mx);      // a fully masked row gives zeros, not NaN
        sum += p;
        const float* vr = Vs + j * RS;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 vv = *reinterpret_cast<const float4*>(vr + c);
            acc[c] = fmaf(p, vv.x, acc[c]); acc[c + 1] = fmaf(p, vv.y, acc[c + 1]);
            acc[c + 2] = fmaf(p, vv.z, acc[c + 2]); acc[c + 3] = fmaf(p, vv.w, acc[c + 3]);
        }
    }
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    float* op = out + ((long)b * L + lane) * E + (long)h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4)
        *reinterpret_cast<float4*>(op + c) = make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv);
}

}  // namespace

extern "C" int soc_small_attn_f32(const float* q, const float* k, const float* v, const float* mask, float* out, int B, int L,
                                  int H, int D, float scale, long mask_batch_stride, long mask_query_stride, void* stream) {
    if (B < 0 || L < 0 || H <= 0) return SOC_EINVAL;
    if (B == 0 || L == 0) return SOC_OK;
    if (!q || !k || !v || !out) return SOC_EINVAL;
    if (L > 64 || (D != 32 && D != 64)) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) != 0) return SOC_EUNSUPPORTED;
    const size_t lds = ((size_t)2 * L * (D + 4) + (size_t)64 * (L + 1)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (D == 64)
        hipLaunchKernelGGL(small_attn_kernel<64>, dim3((unsigned)(B * H)), dim3(64), lds, st, q, k, v, mask, out, L, H, scale,
                           mask_batch_stride, mask_query_stride);
    else
        hipLaunchKernelGGL(small_attn_kernel<32>, dim3((unsigned)(B * H)), dim3(64), lds, st, q, k, v, mask, out, L, H, scale,
                           mask_batch_stride, mask_query_stride);
    return soc_check_launch();
}
