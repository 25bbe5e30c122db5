// K4 experiments (not shipped): P pixel pairs per lane, instance slice per block.  Built and timed by tools/k4_probe.py.
#include <hip/hip_runtime.h>
namespace {
constexpr int CF = 8, CH = 8;
constexpr int NPARAM = (CF + 2) * CH + CH * CH + CH + CH + CH + 1;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int P>
__global__ __launch_bounds__(256) void k4v(const float* __restrict__ feats, const float* __restrict__ params,
                                           const float* __restrict__ refs, float* __restrict__ out, int Q, int hw, int w,
                                           float img_h, float img_w, int stride, int q_per_block) {
    const int t = blockIdx.y;
    int pix[P];
    f32x2 px[P], py[P], f[P][CF];
    const float* fp = feats + (long)t * CF * hw;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        pix[j] = ((blockIdx.x * P + j) * 256 + threadIdx.x) * 2;
        const int p0 = min(pix[j], hw - 1), p1 = min(pix[j] + 1, hw - 1);
        const int y0 = p0 / w, x0 = p0 - y0 * w, y1 = p1 / w, x1 = p1 - y1 * w;
        px[j] = (f32x2){(float)(stride * x0 + stride / 2), (float)(stride * x1 + stride / 2)};
        py[j] = (f32x2){(float)(stride * y0 + stride / 2), (float)(stride * y1 + stride / 2)};
#pragma unroll
        for (int c = 0; c < CF; ++c) f[j][c] = (f32x2){fp[(long)c * hw + p0], fp[(long)c * hw + p1]};
    }
    const int q_lo = blockIdx.z * q_per_block, q_hi = min(Q, q_lo + q_per_block);
    for (int q = q_lo; q < q_hi; ++q) {
        const int inst = t * Q + q;
        const float* __restrict__ Pm = params + (long)inst * NPARAM;
        const float* W0 = Pm;
        const float* W1 = Pm + (CF + 2) * CH;
        const float* W2 = W1 + CH * CH;
        const float* B0 = W2 + CH;
        const float* B1 = B0 + CH;
        const float* B2 = B1 + CH;
        const float RX = refs[inst * 2] * img_w, RY = refs[inst * 2 + 1] * img_h;
        f32x2 h0[P][CH], h1[P][CH];
#pragma unroll
        for (int o = 0; o < CH; ++o) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                f32x2 a = B0[o];
#pragma unroll
                for (int c = 0; c < CF; ++c) a += W0[o * (CF + 2) + c] * f[j][c];
                a += W0[o * (CF + 2) + CF] * (RX - px[j]);
                a += W0[o * (CF + 2) + CF + 1] * (RY - py[j]);
                h0[j][o] = __builtin_elementwise_max(a, (f32x2){0.f, 0.f});
            }
        }
#pragma unroll
        for (int o = 0; o < CH; ++o) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                f32x2 a = B1[o];
#pragma unroll
                for (int c = 0; c < CH; ++c) a += W1[o * CH + c] * h0[j][c];
                h1[j][o] = __builtin_elementwise_max(a, (f32x2){0.f, 0.f});
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            f32x2 r = B2[0];
#pragma unroll
            for (int c = 0; c < CH; ++c) r += W2[c] * h1[j][c];
            if (pix[j] + 1 < hw) *reinterpret_cast<float2*>(out + (long)inst * hw + pix[j]) = make_float2(r[0], r[1]);
        }
    }
}
template <int P>
void go(const float* feats, const float* params, const float* refs, float* out, int T, int Q, int h, int w, float img_h,
        float img_w, int stride, int qpb, hipStream_t s) {
    const int hw = h * w;
    dim3 grid((hw + 512 * P - 1) / (512 * P), T, (Q + qpb - 1) / qpb);
    hipLaunchKernelGGL(k4v<P>, grid, dim3(256), 0, s, feats, params, refs, out, Q, hw, w, img_h, img_w, stride, qpb);
}
}  // namespace
extern "C" int k4_variant(int P, int qpb, const float* feats, const float* params, const float* refs, float* out, int T,
                          int Q, int h, int w, float img_h, float img_w, int stride, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (P == 1) go<1>(feats, params, refs, out, T, Q, h, w, img_h, img_w, stride, qpb, s);
    else if (P == 2) go<2>(feats, params, refs, out, T, Q, h, w, img_h, img_w, stride, qpb, s);
    else if (P == 4) go<4>(feats, params, refs, out, T, Q, h, w, img_h, img_w, stride, qpb, s);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
