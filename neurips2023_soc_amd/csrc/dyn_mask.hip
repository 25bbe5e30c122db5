// K4: per-instance dynamic mask head for gfx950.
//
// For instance (t, q) and pixel (y, x) of the 1/4-scale map:
//   in  = [feats[t, 0..C-1, y, x], rx, ry],  (rx, ry) = ref*(img_w, img_h) - (stride*x + stride/2, ...)
//   out = W2 . relu(W1 . relu(W0 . in + b0) + b1) + b2
// The reference materialises a [1, T*Q*(C+2), h, w] tensor (92 MB at config) and runs three
// grouped convolutions (models/soc.py:439-443, 465-483); here each thread keeps one pixel's C
// features in registers and walks the Q instances of its frame, so HBM traffic is the
// algorithmic minimum: feats read once, [T*Q, h, w] written once (13 MB at config).
// The 169 parameters of an instance are wave-uniform: they are read through the scalar cache
// (s_load) and used as SGPR operands of the FMAs.
#include "soc_common.h"

namespace {

constexpr int CF = 8;    // feature channels (mask_kernels_dim)
constexpr int CH = 8;    // hidden channels (dynamic_mask_channels)
constexpr int NPARAM = (CF + 2) * CH + CH * CH + CH + CH + CH + 1;  // 169

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two horizontally adjacent pixels per lane: every FMA of the three layers is a v_pk_fma_f32 with the
// (wave-uniform, scalar) weight broadcast to both halves -- half the vector instructions per pixel of the
// one-pixel form (the kernel is VALU-issue-bound: 152 FMAs per (pixel, instance) against 36 bytes of traffic).
__global__ __launch_bounds__(256) void dyn_mask_kernel(
    const float* __restrict__ feats, const float* __restrict__ params,
    const float* __restrict__ refs, float* __restrict__ out, int Q, int hw, int w, float img_h,
    float img_w, int stride, int q_per_block) {
    const int t = blockIdx.y;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * 2;    // pixels pix, pix + 1 of the flattened map
    const bool live0 = pix < hw, live1 = pix + 1 < hw;
    const int p0 = live0 ? pix : hw - 1, p1 = live1 ? pix + 1 : hw - 1;
    const int y0 = p0 / w, x0 = p0 - y0 * w;
    const int y1 = p1 / w, x1 = p1 - y1 * w;
    const f32x2 px = {(float)(stride * x0 + stride / 2), (float)(stride * x1 + stride / 2)};
    const f32x2 py = {(float)(stride * y0 + stride / 2), (float)(stride * y1 + stride / 2)};

    f32x2 f[CF];
    const float* fp = feats + (long)t * CF * hw;
#pragma unroll
    for (int c = 0; c < CF; ++c) f[c] = (f32x2){fp[(long)c * hw + p0], fp[(long)c * hw + p1]};

    // blockIdx.z picks a slice of the frame's instances: with all Q per thread the launch has < 2 waves per
    // SIMD at the BASELINE config and the scalar parameter loads are fully exposed
    const int q_lo = blockIdx.z * q_per_block, q_hi = min(Q, q_lo + q_per_block);
    // pix is even: a float2 store is aligned whenever inst*hw is, i.e. always for an even map (wave-uniform test)
    const bool pair_store = (hw & 1) == 0;
    for (int q = q_lo; q < q_hi; ++q) {
        const int inst = t * Q + q;
        const float* __restrict__ P = params + (long)inst * NPARAM;  // wave-uniform
        const f32x2 rx = refs[inst * 2] * img_w - px;
        const f32x2 ry = refs[inst * 2 + 1] * img_h - py;
        f32x2 h0[CH], h1[CH];
        const float* W0 = P;
        const float* W1 = P + (CF + 2) * CH;
        const float* W2 = W1 + CH * CH;
        const float* B0 = W2 + CH;
        const float* B1 = B0 + CH;
        const float* B2 = B1 + CH;
#pragma unroll
        for (int o = 0; o < CH; ++o) {
            f32x2 a = B0[o];
#pragma unroll
            for (int c = 0; c < CF; ++c) a += W0[o * (CF + 2) + c] * f[c];
            a += W0[o * (CF + 2) + CF] * rx;
            a += W0[o * (CF + 2) + CF + 1] * ry;
            h0[o] = __builtin_elementwise_max(a, (f32x2){0.f, 0.f});
        }
#pragma unroll
        for (int o = 0; o < CH; ++o) {
            f32x2 a = B1[o];
#pragma unroll
            for (int c = 0; c < CH; ++c) a += W1[o * CH + c] * h0[c];
            h1[o] = __builtin_elementwise_max(a, (f32x2){0.f, 0.f});
        }
        f32x2 r = B2[0];
#pragma unroll
        for (int c = 0; c < CH; ++c) r += W2[c] * h1[c];
        float* op = out + (long)inst * hw + pix;
        if (pair_store) {
            if (live1) *reinterpret_cast<float2*>(op) = make_float2(r[0], r[1]);
        } else {
            if (live0) op[0] = r[0];
            if (live1) op[1] = r[1];
        }
    }
}

}  // namespace

extern "C" int soc_dyn_mask_f32(const float* feats, const float* params, const float* refs,
                                float* out, int T, int Q, int C, int h, int w, float img_h,
                                float img_w, int stride, void* stream) {
    if (!feats || !params || !refs || !out || T < 0 || Q < 0 || h <= 0 || w <= 0 || stride <= 0)
        return SOC_EINVAL;
    if (C != CF) return SOC_EUNSUPPORTED;
    if (T == 0 || Q == 0) return SOC_OK;
    const int hw = h * w;
    // enough workgroups for ~8 waves per SIMD (256 CUs x 4 SIMDs): split the Q instances over grid.z
    const long base_waves = (long)soc_ceil_div(hw, 512) * 4 * T;
    int groups = (int)((8192 + base_waves - 1) / base_waves);
    groups = groups < 1 ? 1 : (groups > Q ? Q : groups);
    const int q_per_block = soc_ceil_div(Q, groups);
    dim3 grid(soc_ceil_div(hw, 512), T, soc_ceil_div(Q, q_per_block));
    hipLaunchKernelGGL(dyn_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, params,
                       refs, out, Q, hw, w, img_h, img_w, stride, q_per_block);
    return soc_check_launch();
}
