// K5: fused residual add + LayerNorm (row-wise over the last dimension), HBM-bound.
//
//   sum = x + y            (y optional)      -> out_sum  (optional)
//   out_norm = (sum - mean) * rsqrt(var + eps) * gamma + beta
//
// Replaces the `x = shortcut + part1(x)` / `norm2(x)` and `x = x + mlp(..)` / next `norm1(x)` pairs
// of the Swin blocks (reference models/video_swin_transformer.py:262-272, 219) and the
// `src = src + src2; src = norm(src)` pairs of the deformable encoder / decoder / VOC layers
// (models/deformable_transformer.py:247-263,324-347, models/voc.py:44-48,84-94,141-153).
// torch runs these as an add kernel plus a LayerNorm kernel whose C = 96 case reaches only
// 0.75 TB/s (117 us for 88 MB at stage 0); here a row is read once, kept in registers (two-pass
// mean / variance in registers, no Welford), and both outputs are written once.
// Mapping: G = 32 or 64 lanes per row, each lane holds VPL float4 (C <= 2048, C % 4 == 0);
// waves grid-stride over rows, 16-B accesses throughout.
#include "soc_common.h"

namespace {

template <int G, int VPL, int UNR>
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ out_sum, float* __restrict__ out_norm,
    long rows, int C, float eps) {
    constexpr int RPW = 64 / G;                       // row groups per wave
    const int lane = threadIdx.x & 63;
    const int gl = lane % G;                          // lane inside its row group
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    const int nvec = C >> 2;
    const float invC = 1.0f / (float)C;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 gw[VPL], bw[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int c4 = gl + v * G;
        const bool ok = c4 < nvec;
        gw[v] = ok ? reinterpret_cast<const float4*>(gamma)[c4] : z4;
        bw[v] = ok ? reinterpret_cast<const float4*>(beta)[c4] : z4;
    }
    // a row group handles UNR consecutive rows per iteration: all their loads are issued before
    // the first reduction, which is what keeps enough bytes in flight for narrow rows (C = 96)
    for (long base = (wave * RPW + lane / G) * UNR; base < rows; base += nwaves * RPW * UNR) {
        float4 s[UNR][VPL];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long row = base + u;
            const bool live = row < rows;
            const float4* xr = reinterpret_cast<const float4*>(x + row * C);
            const float4* yr = reinterpret_cast<const float4*>(y + row * C);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c4 = gl + v * G;
                float4 a = z4;
                if (live && c4 < nvec) {
                    a = xr[c4];
                    if (y) {
                        const float4 b = yr[c4];
                        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                    }
                }
                s[u][v] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long row = base + u;
            if (row >= rows) break;
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc += (s[u][v].x + s[u][v].y) + (s[u][v].z + s[u][v].w);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            const float mean = acc * invC;
            float var = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                if (gl + v * G < nvec) {
                    const float dx = s[u][v].x - mean, dy = s[u][v].y - mean, dz = s[u][v].z - mean,
                                dw = s[u][v].w - mean;
                    var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) var += __shfl_xor(var, o);
            const float rstd = rsqrtf(var * invC + eps);
            float4* sr = out_sum ? reinterpret_cast<float4*>(out_sum + row * C) : nullptr;
            float4* nr = reinterpret_cast<float4*>(out_norm + row * C);
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c4 = gl + v * G;
                if (c4 < nvec) {
                    if (sr) sr[c4] = s[u][v];
                    float4 o;
                    o.x = (s[u][v].x - mean) * rstd * gw[v].x + bw[v].x;
                    o.y = (s[u][v].y - mean) * rstd * gw[v].y + bw[v].y;
                    o.z = (s[u][v].z - mean) * rstd * gw[v].z + bw[v].z;
                    o.w = (s[u][v].w - mean) * rstd * gw[v].w + bw[v].w;
                    nr[c4] = o;
                }
            }
        }
    }
}

template <int G, int VPL, int UNR>
int launch(const float* x, const float* y, const float* gamma, const float* beta, float* out_sum,
           float* out_norm, long rows, int C, float eps, hipStream_t st) {
    constexpr int RPB = 4 * (64 / G) * UNR;  // rows per 256-thread workgroup per iteration
    long blocks = (rows + RPB - 1) / RPB;
    if (blocks > 256 * 8) blocks = 256 * 8;  // grid-stride beyond 8 workgroups per CU
    hipLaunchKernelGGL((add_layernorm_kernel<G, VPL, UNR>), dim3((unsigned)blocks), dim3(256), 0, st, x, y,
                       gamma, beta, out_sum, out_norm, rows, C, eps);
    return soc_check_launch();
}

}  // namespace

extern "C" int soc_add_layernorm_f32(const float* x, const float* y, const float* gamma,
                                     const float* beta, float* out_sum, float* out_norm, long rows,
                                     int C, float eps, void* stream) {
    if (rows < 0 || C <= 0) return SOC_EINVAL;
    if (C % 4 != 0 || C > 2048) return SOC_EUNSUPPORTED;
    if (rows == 0) return SOC_OK;  // empty tensors carry null pointers
    if (!x || !gamma || !beta || !out_norm) return SOC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nvec = C / 4;
    if (nvec <= 32) return launch<32, 1, 4>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    if (nvec <= 64) return launch<64, 1, 4>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    if (nvec <= 128) return launch<64, 2, 2>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    if (nvec <= 192) return launch<64, 3, 2>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    if (nvec <= 256) return launch<64, 4, 1>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    if (nvec <= 384) return launch<64, 6, 1>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
    return launch<64, 8, 1>(x, y, gamma, beta, out_sum, out_norm, rows, C, eps, st);
}
