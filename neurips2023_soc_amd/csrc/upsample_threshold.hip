// K6: fused bilinear up-sampling (align_corners = False) + threshold of the selected mask logits
// (SURVEY 8a row a21 / 8f rank 3): the output side of the reference's inference loop,
//     F.interpolate(pred_masks, size=(H0, W0), mode='bilinear', align_corners=False)
//     (pred_masks.sigmoid() > 0.5)                          infer_refytb.py:230-231,
//                                                           models/postprocessing.py:222-224
// in one pass: the [T, h, w] logits (57 KB per frame, cache resident) are read, the up-sampled
// logits are never written -- only one byte (0/1) per output pixel leaves the chip.  The source
// index / weight arithmetic restates torch's upsample_bilinear2d (scale = in / out,
// src = max(scale * (dst + 0.5) - 0.5, 0), second tap clamped to the last row / column);
// sigmoid(x) > 0.5 is evaluated as x > 0 (they differ only for 0 < x < 1.2e-7, where fp32
// sigmoid rounds to exactly 0.5).
#include "soc_common.h"

namespace {

constexpr int PX = 16;  // output pixels per thread -> one 16-B store

__global__ __launch_bounds__(256) void upsample_threshold_kernel(
    const float* __restrict__ logits, uint8_t* __restrict__ out, int h, int w, int H0, int W0,
    float sy, float sx, float thr) {
    const int t = blockIdx.z;
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    if (x0 >= W0) return;
    const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f);
    const int y1 = (int)fy;
    const int yp = (y1 < h - 1) ? 1 : 0;
    const float ly1 = fy - (float)y1, ly0 = 1.f - ly1;
    const float* r0 = logits + ((long)t * h + y1) * w;
    const float* r1 = r0 + (long)yp * w;
    union { uint8_t b[PX]; uint4 v; } res;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int x = x0 + i;
        const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
        int x1 = (int)fx;
        x1 = x1 < w - 1 ? x1 : w - 1;
        const int xp = (x1 < w - 1) ? 1 : 0;
        const float lx1 = fx - (float)x1, lx0 = 1.f - lx1;
        const float v = ly0 * (lx0 * r0[x1] + lx1 * r0[x1 + xp]) + ly1 * (lx0 * r1[x1] + lx1 * r1[x1 + xp]);
        res.b[i] = (x < W0 && v > thr) ? 1 : 0;
    }
    uint8_t* o = out + ((long)t * H0 + y) * W0 + x0;
    if (x0 + PX <= W0 && (((uintptr_t)o) & 15) == 0) {
        *reinterpret_cast<uint4*>(o) = res.v;
    } else {
        for (int i = 0; i < PX && x0 + i < W0; ++i) o[i] = res.b[i];
    }
}

// DAVIS form (infer_davis.py:248-272): O objects, each up-sampled and turned into a sigmoid score; scores
// below 0.5 are zeroed, a constant 0.1 background plane is put in front and the label is the argmax
// (first maximum wins, as torch.argmax).  One pass, one byte per output pixel.
__global__ __launch_bounds__(256) void upsample_merge_labels_kernel(
    const float* __restrict__ logits, uint8_t* __restrict__ out, int O, int T, int h, int w, int H0, int W0,
    float sy, float sx, float thr, float bg) {
    const int t = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W0) return;
    const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f);
    const int y1 = (int)fy;
    const int yp = (y1 < h - 1) ? 1 : 0;
    const float ly1 = fy - (float)y1, ly0 = 1.f - ly1;
    const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
    int x1 = (int)fx;
    x1 = x1 < w - 1 ? x1 : w - 1;
    const int xp = (x1 < w - 1) ? 1 : 0;
    const float lx1 = fx - (float)x1, lx0 = 1.f - lx1;
    float best = bg;
    int label = 0;
    for (int o = 0; o < O; ++o) {
        const float* r0 = logits + (((long)o * T + t) * h + y1) * w;
        const float* r1 = r0 + (long)yp * w;
        const float v = ly0 * (lx0 * r0[x1] + lx1 * r0[x1 + xp]) + ly1 * (lx0 * r1[x1] + lx1 * r1[x1 + xp]);
        float sc = 1.f / (1.f + expf(-v));
        if (sc < thr) sc = 0.f;
        if (sc > best) { best = sc; label = o + 1; }
    }
    out[((long)t * H0 + y) * W0 + x] = (uint8_t)label;
}

}  // namespace

extern "C" int soc_upsample_merge_labels_u8(const float* logits, uint8_t* out, int O, int T, int h, int w,
                                            int H0, int W0, float threshold, float background,
                                            void* stream) {
    if (O < 0 || T < 0 || h <= 0 || w <= 0 || H0 <= 0 || W0 <= 0) return SOC_EINVAL;
    if (T == 0) return SOC_OK;
    if ((O > 0 && !logits) || !out) return SOC_EINVAL;
    if (T > 65535 || H0 > 65535 || O > 255) return SOC_EUNSUPPORTED;
    dim3 grid(soc_ceil_div(W0, 256), H0, T);
    hipLaunchKernelGGL(upsample_merge_labels_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, out, O, T,
                       h, w, H0, W0, (float)h / (float)H0, (float)w / (float)W0, threshold, background);
    return soc_check_launch();
}

extern "C" int soc_upsample_threshold_u8(const float* logits, uint8_t* out, int T, int h, int w,
                                         int H0, int W0, float threshold_logit, void* stream) {
    if (T < 0 || h <= 0 || w <= 0 || H0 <= 0 || W0 <= 0) return SOC_EINVAL;
    if (T == 0) return SOC_OK;
    if (!logits || !out) return SOC_EINVAL;
    if (T > 65535 || H0 > 65535) return SOC_EUNSUPPORTED;
    dim3 grid(soc_ceil_div(soc_ceil_div(W0, PX), 256), H0, T);
    hipLaunchKernelGGL(upsample_threshold_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, out, h, w,
                       H0, W0, (float)h / (float)H0, (float)w / (float)W0, threshold_logit);
    return soc_check_launch();
}
