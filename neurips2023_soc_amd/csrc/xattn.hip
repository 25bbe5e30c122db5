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
//   * few queries (decoder self-attention, VOC, lvf: L words x thousands of pixel keys) use a
//     second mapping: one workgroup per (query, head) row, keys spread over the 256 threads.
// q is pre-scaled by 1/sqrt(d) and padded keys get weight 0 (-inf logit), as in PyTorch.
#include "soc_common.h"
#include <math.h>

namespace {

constexpr int HD = 32;

// (query, b, head) rows below this count use the block-per-row mapping (few queries, any Lk)
constexpr long ROWS_BLOCK_PATH = 8192;

// many queries: lane = query row, wave = 64 queries x one head, K/V wave-uniform.
// KV_LDS (Lk <= KV_LDS_MAX, e.g. the <= 32 words of the vlf blocks): the head's K and V rows are staged in LDS once
// per workgroup with coalesced loads and the key loops read them as broadcast ds_reads -- with K/V fetched through
// the scalar cache inside the key loops every wave sits through ~2 Lk dependent memory round trips, and since
// all waves of the launch are resident at once that chain IS the kernel time (42 us for the 59 MB of the 28 800-row
// level).  Otherwise (long key lists) K/V come through the scalar cache as SGPR operands.
constexpr int KV_LDS_MAX = 64;

template <bool KV_LDS>
__global__ __launch_bounds__(256) void xattn_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ kpm, const float* __restrict__ amask, int mask_heads,
    float* __restrict__ out, int Lq, int Lk, int B, int H, float scale, int batch_first) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.y % H;
    const int b = blockIdx.y / H;
    const int qc = min((int)(blockIdx.x * 4 + wave) * 64 + lane, Lq - 1);   // this lane's query row (clamped)
    const int E = H * HD;
    const int k0 = 0, k1 = Lk;
    // element offset of row (l, b): sequence-first (l*B + b)*E, batch-first (b*L + l)*E
    const long q_ls = batch_first ? E : (long)B * E, q_bs = batch_first ? (long)Lq * E : E;
    const long kstride = batch_first ? E : (long)B * E, k_bs = batch_first ? (long)Lk * E : E;

    // q rows (and later the output rows) go through a per-wave LDS tile: global accesses are 8 rows x 128 B per
    // instruction (one fully used line per row) instead of 64 lanes each touching its own row
    __shared__ __attribute__((aligned(16))) float tile_s[4][64 * 36];
    float* tile = tile_s[wave];
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    const int sub_r = lane >> 3, sub_c = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + sub_r;
        const int qq = min(q0 + row, Lq - 1);
        const float4 t = *reinterpret_cast<const float4*>(q + qq * q_ls + b * q_bs + h * HD + sub_c * 4);
        *reinterpret_cast<float4*>(tile + row * 36 + sub_c * 4) = t;
    }
    __builtin_amdgcn_wave_barrier();
    float qr[HD];
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(tile + lane * 36 + 4 * i);   // same wave: no barrier needed
        qr[4 * i] = t.x * scale; qr[4 * i + 1] = t.y * scale;
        qr[4 * i + 2] = t.z * scale; qr[4 * i + 3] = t.w * scale;
    }
    const float* kb = k + b * k_bs + h * HD;  // + j*kstride, wave-uniform
    const float* vb = v + b * k_bs + h * HD;
    long kvs = kstride;                        // row stride of the K/V image the key loops read
    if (KV_LDS) {
        extern __shared__ __attribute__((aligned(16))) float kv_s[];   // [2][Lk][HD]
        float* Ksh = kv_s;
        float* Vsh = kv_s + Lk * HD;
        for (int idx = threadIdx.x; idx < Lk * 8; idx += 256) {
            const int j = idx >> 3, c = idx & 7;
            *reinterpret_cast<float4*>(Ksh + j * HD + c * 4) = *reinterpret_cast<const float4*>(kb + j * kstride + c * 4);
            *reinterpret_cast<float4*>(Vsh + j * HD + c * 4) = *reinterpret_cast<const float4*>(vb + j * kstride + c * 4);
        }
        __syncthreads();
        kb = Ksh; vb = Vsh; kvs = HD;
    }
    const uint8_t* mp = kpm ? kpm + (long)b * Lk : nullptr;
    // additive float mask row of this query: [B or B*H, Lq, Lk] (torch attn_mask semantics)
    const float* am = amask ? amask + (((long)b * mask_heads + (mask_heads > 1 ? h : 0)) * Lq + qc) * Lk : nullptr;

    // pass 1: row maximum over this block's keys
    float mx = -INFINITY;
    for (int j = k0; j < k1; ++j) {
        if (mp && mp[j]) continue;  // wave-uniform
        const float* kr = kb + j * kvs;
        float s = am ? am[j] : 0.f;
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
        const float* kr = kb + j * kvs;
        const float* vr = vb + j * kvs;
        float s = am ? am[j] : 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += qr[d] * kr[d];
        const float p = __expf(s - mx);
        l += p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] += p * vr[d];
    }
    const float inv = 1.f / l;  // l == 0 (all keys padded) -> NaN, as torch.softmax gives
#pragma unroll
    for (int i = 0; i < HD / 4; ++i)
        *reinterpret_cast<float4*>(tile + lane * 36 + 4 * i) =
            make_float4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + sub_r;
        if (q0 + row < Lq)
            *reinterpret_cast<float4*>(out + (q0 + row) * q_ls + b * q_bs + h * HD + sub_c * 4) =
                *reinterpret_cast<const float4*>(tile + row * 36 + sub_c * 4);
    }
}

// few queries (decoder / VOC / lvf): one 256-thread workgroup per (query, b, head) row.
//   phase A  thread j scores keys j, j+256, ... (q broadcast, K rows read as 8 x float4) -> LDS
//   phase B  block max / sum, p = exp(s - max) in place
//   phase C  32 key slots x 8 lanes(float4 of V): acc4 += p[j] * V[j]; slots reduced through LDS
__global__ __launch_bounds__(256) void xattn_row_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ kpm, const float* __restrict__ amask, int mask_heads,
    float* __restrict__ out, int Lq, int Lk, int B, int H, float scale, int batch_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sc = lds;                       // [Lk] scores -> probabilities
    float* red = lds + ((Lk + 3) & ~3);    // [32 slots][32 dims] partial outputs / 8 reduce words
    const int tid = threadIdx.x;
    const int row = blockIdx.x;            // (qi, b, h) flat, h fastest
    const int h = row % H;
    const int b = (row / H) % B;
    const int qi = row / (H * B);
    const int E = H * HD;
    const long q_ls = batch_first ? E : (long)B * E, q_bs = batch_first ? (long)Lq * E : E;
    const long kstride = batch_first ? E : (long)B * E, k_bs = batch_first ? (long)Lk * E : E;
    const float* kb = k + b * k_bs + h * HD;
    const float* vb = v + b * k_bs + h * HD;
    const uint8_t* mp = kpm ? kpm + (long)b * Lk : nullptr;
    const float* am = amask ? amask + (((long)b * mask_heads + (mask_heads > 1 ? h : 0)) * Lq + qi) * Lk : nullptr;

    float qr[HD];
    {
        const float4* qp = reinterpret_cast<const float4*>(q + qi * q_ls + b * q_bs + h * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            const float4 t = qp[i];
            qr[4 * i] = t.x * scale; qr[4 * i + 1] = t.y * scale;
            qr[4 * i + 2] = t.z * scale; qr[4 * i + 3] = t.w * scale;
        }
    }
    float mx = -INFINITY;
    for (int j = tid; j < Lk; j += 256) {
        float s = -INFINITY;
        if (!(mp && mp[j])) {
            const float4* kr = reinterpret_cast<const float4*>(kb + j * kstride);
            s = am ? am[j] : 0.f;
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                const float4 t = kr[i];
                s += qr[4 * i] * t.x + qr[4 * i + 1] * t.y + qr[4 * i + 2] * t.z + qr[4 * i + 3] * t.w;
            }
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < Lk; j += 256) {
        const float s = sc[j];
        const float e = (s == -INFINITY) ? 0.f : __expf(s - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    __syncthreads();  // red is reused below

    const int d4 = tid & 7, slot = tid >> 3;  // 32 slots
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = slot; j < Lk; j += 32) {
        const float p = sc[j];
        const float4 t = *reinterpret_cast<const float4*>(vb + j * kstride + d4 * 4);
        acc.x += p * t.x; acc.y += p * t.y; acc.z += p * t.z; acc.w += p * t.w;
    }
    *reinterpret_cast<float4*>(red + slot * HD + d4 * 4) = acc;
    __syncthreads();
    if (tid < HD) {
        float a = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) a += red[s2 * HD + tid];
        out[qi * q_ls + b * q_bs + h * HD + tid] = a / sum;
    }
}

}  // namespace

extern "C" size_t soc_xattn_workspace_bytes(int Lq, int Lk, int B, int n_heads, int head_dim) {
    (void)Lq; (void)Lk; (void)B; (void)n_heads; (void)head_dim;
    return 0;  // ABI v1 keeps the query; no mapping of the current build needs scratch memory
}

extern "C" int soc_xattn_f32(const float* q, const float* k, const float* v,
                             const uint8_t* key_pad_mask, const float* attn_mask, int attn_mask_heads,
                             float* out, int Lq, int Lk, int B, int n_heads, int head_dim,
                             int batch_first, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (!q || !k || !v || !out || Lq < 0 || Lk <= 0 || B <= 0 || n_heads <= 0) return SOC_EINVAL;
    if (head_dim != HD) return SOC_EUNSUPPORTED;
    if (attn_mask && attn_mask_heads != 1 && attn_mask_heads != n_heads) return SOC_EINVAL;
    if (Lq == 0) return SOC_OK;
    hipStream_t st = (hipStream_t)stream;
    const float scale = (float)sqrt(1.0 / (double)head_dim);
    const long rows = (long)Lq * B * n_heads;
    const size_t row_lds = (size_t)(((Lk + 3) & ~3) + 32 * HD) * sizeof(float);
    if (rows <= ROWS_BLOCK_PATH && row_lds <= 64 * 1024) {
        hipLaunchKernelGGL(xattn_row_kernel, dim3((unsigned)rows), dim3(256), row_lds, st, q, k, v,
                           key_pad_mask, attn_mask, attn_mask_heads, out, Lq, Lk, B, n_heads, scale, batch_first);
        return soc_check_launch();
    }
    dim3 grid(soc_ceil_div(Lq, 256), B * n_heads);
    if (Lk <= KV_LDS_MAX)
        hipLaunchKernelGGL(xattn_kernel<true>, grid, dim3(256), (size_t)2 * Lk * HD * sizeof(float), st, q, k, v,
                           key_pad_mask, attn_mask, attn_mask_heads, out, Lq, Lk, B, n_heads, scale, batch_first);
    else
        hipLaunchKernelGGL(xattn_kernel<false>, grid, dim3(256), 0, st, q, k, v, key_pad_mask, attn_mask,
                           attn_mask_heads, out, Lq, Lk, B, n_heads, scale, batch_first);
    return soc_check_launch();
}
