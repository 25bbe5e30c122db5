// K11: Video-Swin patch merging, gather + LayerNorm in one pass (HBM-bound).
//
//   x [B, D, H, W, C] token-major  ->  out [B, D, ceil(H/2), ceil(W/2), 4C]
//   out[.., h2, w2, :] = LayerNorm(concat(x[2h2, 2w2], x[2h2+1, 2w2], x[2h2, 2w2+1], x[2h2+1, 2w2+1]))
// (rows / columns past an odd edge read as zeros), i.e. PatchMerging.forward up to the `reduction`
// Linear (reference models/video_swin_transformer.py:279-313: pad, four strided slices, torch.cat, norm).
// The reference materialises the concatenation (4 strided copies + cat), then normalises it; here one
// wave owns an output row, gathers its four C-float source rows with 16-B loads, keeps the 4C values in
// registers for the two-pass mean / variance and writes the normalised row once.
#include "soc_common.h"
#include <type_traits>

namespace {

template <int VPL>   // float4 per lane: 4C <= 256 * VPL
__global__ __launch_bounds__(256) void patch_merge_ln_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out, long rows, int H, int W, int C, int H2, int W2, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int cv = C >> 2, nvec = C;          // float4 per source row / per output row (4C / 4)
    const float inv = 1.0f / (float)(4 * C);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long row = wave; row < rows; row += nwaves) {
        const int w2 = (int)(row % W2);
        const int h2 = (int)((row / W2) % H2);
        const long bd = row / ((long)W2 * H2);
        float4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int slot = lane + 64 * i;
            v[i] = z4;
            if (slot < nvec) {
                const int seg = slot / cv, c4 = slot - seg * cv;
                const int h = 2 * h2 + (seg & 1), w = 2 * w2 + (seg >> 1);
                if (h < H && w < W)
                    v[i] = reinterpret_cast<const float4*>(x + ((bd * H + h) * W + w) * C)[c4];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            if (lane + 64 * i < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * inv + eps);
        float4* orow = reinterpret_cast<float4*>(out + row * 4 * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int slot = lane + 64 * i;
            if (slot < nvec) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[slot];
                const float4 b = reinterpret_cast<const float4*>(beta)[slot];
                orow[slot] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                         (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
            }
        }
    }
}

}  // namespace

extern "C" int soc_patch_merge_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out,
                                             int BD, int H, int W, int C, float eps, void* stream) {
    if (BD < 0 || H <= 0 || W <= 0 || C <= 0) return SOC_EINVAL;
    if (BD == 0) return SOC_OK;
    if (!x || !gamma || !beta || !out) return SOC_EINVAL;
    if (C % 4 != 0 || C > 512) return SOC_EUNSUPPORTED;
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long rows = (long)BD * H2 * W2;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    auto launch = [&](auto vpl) {
        hipLaunchKernelGGL(patch_merge_ln_kernel<decltype(vpl)::value>, dim3(blocks), dim3(256), 0, st, x, gamma, beta,
                           out, rows, H, W, C, H2, W2, eps);
    };
    if (C <= 64) launch(std::integral_constant<int, 1>{});
    else if (C <= 128) launch(std::integral_constant<int, 2>{});
    else if (C <= 256) launch(std::integral_constant<int, 4>{});
    else launch(std::integral_constant<int, 8>{});
    return soc_check_launch();
}
