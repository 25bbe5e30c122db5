// K17 / K18: the elementwise passes of the FPN spatial decoder (reference models/segmentation.py:41-74) as two fused
// kernels on NCHW maps.  Through torch each conv3x3 is followed by bias add, three GroupNorm launches and a ReLU, and each
// lateral connection by a bias add, a nearest up-sampling and an add: ~35 chip-wide launches of ~5 us per clip that run
// beside the next clip's head in the software pipeline and are charged to it ("what the pipeline leaves on the
// table").  Here:
//   K17  y = relu(GroupNorm(x + bias_c))          one workgroup per (sample, group), the group's values in registers
//   K18  y = lateral + bias_c + nearest_up(prev)  F.interpolate(mode="nearest") index rule
// The convolutions themselves stay MIOpen calls (without bias).
#include "soc_common.h"

namespace {

constexpr int GN_THREADS = 1024;
constexpr int GN_MAXV = 8;       // float4 per thread: groups of up to 32 768 values

__device__ __forceinline__ float block_sum(float v, float* red, const int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                       // red may still be read from the previous call
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GN_THREADS / 64; ++i) s += red[i];
    return s;
}

__global__ __launch_bounds__(GN_THREADS) void groupnorm_nchw_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, int C, int G, int HW, float eps, int relu) {
    __shared__ float red[GN_THREADS / 64];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const int nvec = Cg * HW / 4;                       // HW % 4 == 0 (host check): a float4 never straddles channels
    const long base = ((long)n * C + (long)g * Cg) * HW;
    const float4* xp = reinterpret_cast<const float4*>(x + base);
    float4 v[GN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < GN_MAXV; ++k) {
        const int i = tid + k * GN_THREADS;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nvec) {
            v[k] = xp[i];
            if (bias) {
                const float b = bias[g * Cg + (4 * i) / HW];
                v[k].x += b; v[k].y += b; v[k].z += b; v[k].w += b;
            }
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    const float inv = 1.f / (float)(Cg * HW);
    const float mean = block_sum(s, red, tid) * inv;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < GN_MAXV; ++k) {
        if (tid + k * GN_THREADS < nvec) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(block_sum(q, red, tid) * inv + eps);
    float4* yp = reinterpret_cast<float4*>(y + base);
#pragma unroll
    for (int k = 0; k < GN_MAXV; ++k) {
        const int i = tid + k * GN_THREADS;
        if (i < nvec) {
            const int c = g * Cg + (4 * i) / HW;
            const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
            float4 o = make_float4(v[k].x * sc + sh, v[k].y * sc + sh, v[k].z * sc + sh, v[k].w * sc + sh);
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            yp[i] = o;
        }
    }
}

__global__ __launch_bounds__(256) void upsample_add_nchw_kernel(
    const float* __restrict__ lat, const float* __restrict__ bias, const float* __restrict__ prev,
    float* __restrict__ y, int C, int H, int W, int Hp, int Wp, float sh, float sw, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % W);
    const long r = i / W;
    const int yo = (int)(r % H);
    const long nc = r / H;
    // at::native::nearest_neighbor_compute_source_index: min(floor(dst * scale), in - 1), scale = (float)in / out
    const int ys = min((int)floorf(yo * sh), Hp - 1), xs = min((int)floorf(xo * sw), Wp - 1);
    float v = lat[i] + prev[(nc * Hp + ys) * Wp + xs];
    if (bias) v += bias[nc % C];
    y[i] = v;
}

__global__ __launch_bounds__(256) void upsample_add_tokens_kernel(
    const float* __restrict__ lat, const float* __restrict__ bias, const float* __restrict__ prev,
    float* __restrict__ y, int C4, int H, int W, int Hp, int Wp, float sh, float sw, long total4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // float4 index into [N, H, W, C]
    if (i >= total4) return;
    const int c4 = (int)(i % C4);
    const long pix = i / C4;
    const int xo = (int)(pix % W);
    const long r = pix / W;
    const int yo = (int)(r % H);
    const long n = r / H;
    const int ys = min((int)floorf(yo * sh), Hp - 1), xs = min((int)floorf(xo * sw), Wp - 1);
    const float4 a = reinterpret_cast<const float4*>(lat)[i];
    const float4 p = reinterpret_cast<const float4*>(prev)[((n * Hp + ys) * Wp + xs) * C4 + c4];
    float4 o = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    if (bias) {
        const float4 b = reinterpret_cast<const float4*>(bias)[c4];
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    reinterpret_cast<float4*>(y)[i] = o;
}

}  // namespace

extern "C" int soc_upsample_add_tokens_f32(const float* lateral, const float* bias, const float* prev, float* y, int N,
                                           int C, int H, int W, int Hp, int Wp, void* stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || Hp <= 0 || Wp <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!lateral || !prev || !y) return SOC_EINVAL;
    if (C % 4 || (((uintptr_t)lateral | (uintptr_t)prev | (uintptr_t)y | (uintptr_t)bias) & 15)) return SOC_EUNSUPPORTED;
    const long total4 = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample_add_tokens_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, lateral, bias, prev, y, C / 4, H, W, Hp, Wp, (float)Hp / (float)H,
                       (float)Wp / (float)W, total4);
    return soc_check_launch();
}

extern "C" int soc_groupnorm_nchw_f32(const float* x, const float* bias, const float* gamma, const float* beta, float* y,
                                      int N, int C, int HW, int groups, float eps, int relu, void* stream) {
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!x || !gamma || !beta || !y) return SOC_EINVAL;
    const long per_group = (long)(C / groups) * HW;
    if (HW % 4 || per_group / 4 > (long)GN_MAXV * GN_THREADS || (((uintptr_t)x | (uintptr_t)y) & 15))
        return SOC_EUNSUPPORTED;
    hipLaunchKernelGGL(groupnorm_nchw_kernel, dim3(N * groups), dim3(GN_THREADS), 0, (hipStream_t)stream, x, bias, gamma,
                       beta, y, C, groups, HW, eps, relu);
    return soc_check_launch();
}

extern "C" int soc_upsample_add_nchw_f32(const float* lateral, const float* bias, const float* prev, float* y, int N,
                                         int C, int H, int W, int Hp, int Wp, void* stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || Hp <= 0 || Wp <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!lateral || !prev || !y) return SOC_EINVAL;
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(upsample_add_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       lateral, bias, prev, y, C, H, W, Hp, Wp, (float)Hp / (float)H, (float)Wp / (float)W, total);
    return soc_check_launch();
}
