// K16: a chain of up to three 256-wide linear layers on few rows, optionally closed by a residual add + LayerNorm, in
// ONE launch ("out_proj + residual + LN as one launch; the 3-layer bbox / controller MLPs as one
// launch each").
//
//   h0 = x (+ x_add);  h_{i+1} = relu(h_i W_i^T + b_i) for i < n_layers - 1;  y = h W_last^T + b_last      [n_out <= 256]
//   out = y                                   (no LayerNorm)
//   out = LayerNorm(residual + y)             (LayerNorm given: n_out == 256)
//
// Replaces, on the decoder -> VOC -> heads chain (M = 160 frame queries / 20 video queries):
//   * MLP.forward of bbox_embed (256-256-256-4) and controller (256-256-256-169)      reference models/soc.py:552-564
//   * out_proj of nn.MultiheadAttention + dropout-free residual + LayerNorm             models/deformable_transformer.py
//     :330-334 (self-attention of the decoder layer), models/voc.py:44-48,84-94,141-153 (VOC layers)
// Each of these was 2-3 launches of ~5 us with a dependent-launch gap in between.  One workgroup (16 waves) per row; a
// layer = 16 groups of 16 dot products of length 256 (csrc/row_ops.h), activations stay in LDS.
// For a chain that owns the GPU (one clip per replay 9.40 -> 9.17 ms together with K15).  Inside the software pipeline
// the chain runs beside another clip's chip-filling kernels and what it costs them is its CU time, not its launch
// count: there K7's many small workgroups are cheaper (8.70 vs 8.76 ms per clip) and hot_ops keeps them.
#include "soc_common.h"
#include "row_ops.h"

namespace {


struct RowMlpArgs {
    const float* x;          // [M, 256]
    const float* xadd;       // rows broadcast as (m / add_div) % add_mod, or null
    int add_div, add_mod;
    int n_layers;            // 1..3
    const float* w[3];       // [256, 256] except the last: [n_out, 256]
    const float* b[3];       // may be null
    int n_out;
    const float* residual;   // [M, n_out] or null
    const float *gamma, *beta;   // LayerNorm over n_out == 256, or null
    float eps;
    float* out;              // [M, n_out]
};

template <int THREADS, int R>
__global__ __launch_bounds__(THREADS) void row_mlp_kernel(const RowMlpArgs a) {
    constexpr int NWAVES = THREADS / 64;
    __shared__ __attribute__((aligned(16))) float buf[2][ROW_DM];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    if (tid < ROW_DM) {
        float v = a.x[row * ROW_DM + tid];
        if (a.xadd) v += a.xadd[(long)((row / a.add_div) % a.add_mod) * ROW_DM + tid];
        buf[0][tid] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const bool last = l + 1 == a.n_layers;
        const int n = last ? a.n_out : ROW_DM;
        const float4 x4 = reinterpret_cast<const float4*>(buf[cur])[lane];
        for (int g = wave; g * R < n; g += NWAVES)
            matvec_rows<R>(a.w[l], a.b[l], 1.f, g * R, n, x4, buf[cur ^ 1], lane, !last);
        cur ^= 1;
        __syncthreads();
    }
    const float* y = buf[cur];
    if (a.gamma == nullptr) {
        if (tid < a.n_out) {
            float v = y[tid];
            if (a.residual) v += a.residual[row * a.n_out + tid];
            a.out[row * a.n_out + tid] = v;
        }
        return;
    }
    // LayerNorm(residual + y) over 256 columns: waves 0-3 hold the row
    float v = 0.f, mean = 0.f;
    if (tid < ROW_DM) {
        v = y[tid] + (a.residual ? a.residual[row * ROW_DM + tid] : 0.f);
        float s = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
    }
    __syncthreads();
    if (tid < ROW_DM) {
        mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.f / ROW_DM);
        const float d = v - mean;
        float q = d * d;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if (lane == 0) red[4 + wave] = q;
    }
    __syncthreads();
    if (tid < ROW_DM) {
        const float var = ((red[4] + red[5]) + (red[6] + red[7])) * (1.f / ROW_DM);
        a.out[row * ROW_DM + tid] = (v - mean) * rsqrtf(var + a.eps) * a.gamma[tid] + a.beta[tid];
    }
}

}  // namespace

extern "C" int soc_row_mlp_f32(const float* x, const float* x_add, int add_div, int add_mod, int n_layers,
                               const float* const* w, const float* const* bias, int n_out, const float* residual,
                               const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int M, int K,
                               void* stream) {
    if (M < 0 || n_layers < 1 || !w || !out || n_out <= 0) return SOC_EINVAL;
    if (n_layers > 3 || K != ROW_DM || n_out > ROW_DM) return SOC_EUNSUPPORTED;
    if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return SOC_EINVAL;
    if (ln_gamma && n_out != ROW_DM) return SOC_EUNSUPPORTED;
    if (x_add && (add_div <= 0 || add_mod <= 0)) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (!x) return SOC_EINVAL;
    RowMlpArgs a;
    a.x = x; a.xadd = x_add; a.add_div = x_add ? add_div : 1; a.add_mod = x_add ? add_mod : 1;
    a.n_layers = n_layers;
    for (int i = 0; i < 3; ++i) {
        const int j = i < n_layers ? i : 0;
        if (!w[j]) return SOC_EINVAL;
        if (((uintptr_t)w[j] & 15) != 0) return SOC_EUNSUPPORTED;
        a.w[i] = w[j];
        a.b[i] = bias ? bias[j] : nullptr;
    }
    a.n_out = n_out; a.residual = residual; a.gamma = ln_gamma; a.beta = ln_beta; a.eps = ln_eps; a.out = out;
    hipLaunchKernelGGL((row_mlp_kernel<1024, 16>), dim3(M), dim3(1024), 0, (hipStream_t)stream, a);
    return soc_check_launch();
}
