// K8: iterative box refinement of the deformable decoder for gfx950 -- one launch for what the
// reference writes as ~13 elementwise ops per layer:
//   new_ref = sigmoid(delta + inverse_sigmoid(ref))            (ref_dim 4)
//   new_ref = sigmoid([delta.xy + inverse_sigmoid(ref), delta.wh])   (ref_dim 2, first layer / heads)
//   ref_in  = new_ref[:, :, None, :] * [vr, vr][:, None, :, :]  (next layer's sampling reference)
// inverse_sigmoid(x) = log(max(clamp(x,0,1), eps) / max(1 - clamp(x,0,1), eps)), eps = 1e-5.
// One lane per (frame, query) row; 160 rows -- the point is the launch count on a latency-bound chain.
#include "soc_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float inv_sigmoid(float x) {
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float a = fmaxf(x, 1e-5f), b = fmaxf(1.f - x, 1e-5f);
    return logf(a / b);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(64) void box_refine_kernel(
    const float* __restrict__ delta, const float* __restrict__ ref, int ref_dim,
    const float* __restrict__ vr, float* __restrict__ new_ref, float* __restrict__ ref_in, int rows,
    int Q, int L) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float4 d = *reinterpret_cast<const float4*>(delta + (long)row * 4);
    float4 o;
    if (ref_dim == 4) {
        const float4 r = *reinterpret_cast<const float4*>(ref + (long)row * 4);
        o = make_float4(sigmoidf(d.x + inv_sigmoid(r.x)), sigmoidf(d.y + inv_sigmoid(r.y)),
                        sigmoidf(d.z + inv_sigmoid(r.z)), sigmoidf(d.w + inv_sigmoid(r.w)));
    } else {
        const float2 r = *reinterpret_cast<const float2*>(ref + (long)row * 2);
        o = make_float4(sigmoidf(d.x + inv_sigmoid(r.x)), sigmoidf(d.y + inv_sigmoid(r.y)), sigmoidf(d.z),
                        sigmoidf(d.w));
    }
    *reinterpret_cast<float4*>(new_ref + (long)row * 4) = o;
    if (ref_in) {
        const int n = row / Q;
        for (int l = 0; l < L; ++l) {
            const float2 v = *reinterpret_cast<const float2*>(vr + ((long)n * L + l) * 2);
            *reinterpret_cast<float4*>(ref_in + ((long)row * L + l) * 4) =
                make_float4(o.x * v.x, o.y * v.y, o.z * v.x, o.w * v.y);
        }
    }
}

}  // namespace

extern "C" int soc_box_refine_f32(const float* delta, const float* ref, int ref_dim,
                                  const float* valid_ratios, float* new_ref, float* ref_in, int N, int Q,
                                  int L, void* stream) {
    if (N < 0 || Q < 0 || (ref_dim != 2 && ref_dim != 4)) return SOC_EINVAL;
    const long rows = (long)N * Q;
    if (rows == 0) return SOC_OK;
    if (!delta || !ref || !new_ref) return SOC_EINVAL;
    if (ref_in && (!valid_ratios || L <= 0)) return SOC_EINVAL;
    hipLaunchKernelGGL(box_refine_kernel, dim3(soc_ceil_div(rows, 64)), dim3(64), 0, (hipStream_t)stream, delta,
                       ref, ref_dim, valid_ratios, new_ref, ref_in, (int)rows, Q, L);
    return soc_check_launch();
}
