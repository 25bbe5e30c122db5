// K3: multi-head attention core for gfx950 -- softmax(q k^T / sqrt(d)) v on projected tensors.
//
// Shapes on SOC's path: tens of thousands of pixel queries against
// <= ~20 text keys (vlf), a handful of word queries against thousands of pixel keys (lvf), and
// the 20..160-token attention of the decoder / VOC.  None of it is MFMA-shaped work (Lk or Lq
// is tiny, d = 32); the kernel is HBM-bound on q/out rows, so the mapping is:
//   * one lane = one query row of one head (q and the 32-wide accumulator live in registers);
//   * one wave  = 64 queries x one head, so K/V addresses are wave-uniform: the key loop reads
//     K and V through the scalar cache and the FMAs take them as SGPR operands -- no LDS at all;
//   * exact two-pass softmax per key chunk (max first, then exp/accumulate), which mirrors
//     torch.softmax as used by torch.nn.functional.multi_head_attention_forward;
//   * few queries x many keys (lvf) are split over keys across blocks; partial (max, sum, acc)
//     triples go to a workspace and a second kernel merges them.
// q is pre-scaled by 1/sqrt(d) and padded keys get weight 0 (-inf logit), as in PyTorch.
#include "soc_common.h"
#include <math.h>

namespace {

constexpr int HD = 32;

struct Split {
    int nsplit;
    int keys_per_split;
};

__host__ __device__ inline Split choose_split(int Lq, int Lk, int B, int n_heads) {
    const long waves = (long)((Lq + 63) / 64) * B * n_heads;
    Split s{1, Lk};
    if (Lk >= 256 && waves < 512) {
        int want = (int)((1024 + waves - 1) / waves);
        int maxs = (Lk + 63) / 64;
        int ns = want < maxs ? want : maxs;
        if (ns < 1) ns = 1;
        s.keys_per_split = ((Lk + ns - 1) / ns + 3) & ~3;
        s.nsplit = (Lk + s.keys_per_split - 1) / s.keys_per_split;
    }
    return s;
}

// PARTIAL = false: writes normalised output.  PARTIAL = true: writes (m, l, acc[32]) per
// (split, query, b, head) into ws.
template <bool PARTIAL>
__global__ __launch_bounds__(256) void xattn_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ kpm, float* __restrict__ out, float* __restrict__ ws, int Lq,
    int Lk, int B, int H, float scale, int keys_per_split) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.y % H;
    const int b = blockIdx.y / H;
    const int split = blockIdx.z;
    const int qi = (blockIdx.x * 4 + wave) * 64 + lane;
    const bool live = qi < Lq;
    const int qc = live ? qi : Lq - 1;
    const int E = H * HD;
    const int k0 = split * keys_per_split;
    const int k1 = min(Lk, k0 + keys_per_split);

    float qr[HD];
    {
        const float4* qp = reinterpret_cast<const float4*>(q + ((long)qc * B + b) * E + h * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 t = qp[i];
            qr[4 * i] = t.x * scale; qr[4 * i + 1] = t.y * scale;
            qr[4 * i + 2] = t.z * scale; qr[4 * i + 3] = t.w * scale;
        }
    }
    const float* kb = k + (long)b * E + h * HD;  // + j*B*E, wave-uniform
    const float* vb = v + (long)b * E + h * HD;
    const long kstride = (long)B * E;
    const uint8_t* mp = kpm ? kpm + (long)b * Lk : nullptr;

    // pass 1: row maximum over this block's keys
    float mx = -INFINITY;
    for (int j = k0; j < k1; ++j) {
        if (mp && mp[j]) continue;  // wave-uniform
        const float* kr = kb + j * kstride;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += qr[d] * kr[d];
        mx = fmaxf(mx, s);
    }
    // pass 2: exp / accumulate
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    float l = 0.f;
    for (int j = k0; j < k1; ++j) {
        if (mp && mp[j]) continue;
        const float* kr = kb + j * kstride;
        const float* vr = vb + j * kstride;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += qr[d] * kr[d];
        const float p = __expf(s - mx);
        l += p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += p * vr[d];
    }
    if (!live) return;
    if (PARTIAL) {
        float* w = ws + ((((long)split * Lq + qi) * B + b) * H + h) * (HD + 2);
        w[0] = mx;
        w[1] = l;
#pragma unroll
        for (int d = 0; d < HD; ++d) w[2 + d] = acc[d];
    } else {
        const float inv = 1.f / l;  // l == 0 (all keys padded) -> NaN, as torch.softmax gives
        float4* op = reinterpret_cast<float4*>(out + ((long)qi * B + b) * E + h * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i)
            op[i] = make_float4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv,
                                acc[4 * i + 3] * inv);
    }
}

// merge the per-split partials: one thread per (query, b, head, dim)
__global__ __launch_bounds__(256) void xattn_merge_kernel(const float* __restrict__ ws,
                                                          float* __restrict__ out, long rows,
                                                          int nsplit) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * HD) return;
    const long row = idx / HD;  // (query, b, head) flat == out row of 32
    const int d = (int)(idx % HD);
    float mx = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, ws[((long)s * rows + row) * (HD + 2)]);
    float l = 0.f, a = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* w = ws + ((long)s * rows + row) * (HD + 2);
        const float f = (w[0] == -INFINITY) ? 0.f : __expf(w[0] - mx);
        l += w[1] * f;
        a += w[2 + d] * f;
    }
    out[row * HD + d] = a / l;
}

}  // namespace

extern "C" size_t soc_xattn_workspace_bytes(int Lq, int Lk, int B, int n_heads, int head_dim) {
    if (Lq <= 0 || Lk <= 0 || B <= 0 || n_heads <= 0 || head_dim != HD) return 0;
    const Split s = choose_split(Lq, Lk, B, n_heads);
    if (s.nsplit <= 1) return 0;
    return (size_t)s.nsplit * Lq * B * n_heads * (HD + 2) * sizeof(float);
}

extern "C" int soc_xattn_f32(const float* q, const float* k, const float* v,
                             const uint8_t* key_pad_mask, float* out, int Lq, int Lk, int B,
                             int n_heads, int head_dim, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!q || !k || !v || !out || Lq < 0 || Lk <= 0 || B <= 0 || n_heads <= 0) return SOC_EINVAL;
    if (head_dim != HD) return SOC_EUNSUPPORTED;
    if (Lq == 0) return SOC_OK;
    hipStream_t st = (hipStream_t)stream;
    const Split s = choose_split(Lq, Lk, B, n_heads);
    const float scale = (float)sqrt(1.0 / (double)head_dim);
    dim3 grid(soc_ceil_div(Lq, 256), B * n_heads, s.nsplit);
    if (s.nsplit <= 1) {
        hipLaunchKernelGGL(xattn_kernel<false>, grid, dim3(256), 0, st, q, k, v, key_pad_mask, out,
                           (float*)nullptr, Lq, Lk, B, n_heads, scale, s.keys_per_split);
        return soc_check_launch();
    }
    const size_t need = soc_xattn_workspace_bytes(Lq, Lk, B, n_heads, head_dim);
    if (!workspace || workspace_bytes < need) return SOC_EWORKSPACE;
    hipLaunchKernelGGL(xattn_kernel<true>, grid, dim3(256), 0, st, q, k, v, key_pad_mask, out,
                       (float*)workspace, Lq, Lk, B, n_heads, scale, s.keys_per_split);
    const long rows = (long)Lq * B * n_heads;
    hipLaunchKernelGGL(xattn_merge_kernel, dim3(soc_ceil_div(rows * HD, 256)), dim3(256), 0, st,
                       (const float*)workspace, out, rows, s.nsplit);
    return soc_check_launch();
}
