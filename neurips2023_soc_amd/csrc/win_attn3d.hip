// K1: 3-D shifted-window attention for gfx950.
//
// One workgroup = one (window, head) pair (optionally one slice of its query tiles).
//   stage 0  every thread derives, for the window's N <= 400 tokens, where the token lives in
//            the un-padded token-major qkv tensor after pad + cyclic roll (or that it is a
//            zero-padded token, whose q/k/v are the qkv bias), its shift-mask region id and its
//            relative-position code e(i) -- no mask / index tensors are ever read from HBM;
//   stage 1  K and V of the head (N x 32 f32 each) and the head's bias-table column are staged
//            in LDS (row stride 36 floats: conflict-free V reads, 2-way K reads);
//   stage 2  each wave owns 16-query tiles.  S^T = K . Q^T is computed with
//            v_mfma_f32_16x16x4_f32 (exact f32, SURVEY 8d: fp32 only on this path) for ALL key
//            tiles at once -- 25 tiles x 4 accumulators = 100 VGPRs hold the whole 16 x 400
//            score block, so the softmax is the exact two-pass form (no online rescale) and is
//            done in registers: bias via LDS lookup table[e(i) - e(j) + E0], shift mask -100,
//            row max / sum by two cross-lane xor-shuffles (the C layout of S^T keeps one query
//            per lane column);
//   stage 3  O^T = V^T . P^T reuses the score accumulators directly as the MFMA B operand (the
//            key order inside each 4-wide k-step is permuted consistently on the V side), so P
//            never leaves registers; the result is scaled by 1/rowsum and scattered back to the
//            token-major output with the inverse roll, dropping padded tokens.
// MFMA work per (window, head): 2 * 25 * 25 * 8 = 10 000 instructions of 2048 FLOP.
#include "soc_common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int HD = 32;          // head dim (all Video-Swin variants)
constexpr int RS = 36;          // LDS row stride in floats for K and V
constexpr int NT_MAX = 25;      // key / query tiles of 16 -> up to 400 tokens
constexpr int NP_MAX = NT_MAX * 16;
constexpr int THREADS = 512;    // 8 waves, 2 per SIMD

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float lds_cfloat;  // LDS-space pointer (ds_read, 32-bit address)

struct WinParams {
    int B, D, H, W, C, nH;
    int wd, wh, ww;      // clamped window
    int sd, sh, sw;      // clamped shift
    int td, th, tw;      // nominal (table) window
    int Dp, Hp, Wp;      // padded volume
    int nwd, nwh, nww;   // windows per axis
    int N;               // tokens per window
    int NT;              // tiles of 16
    int qsplit;          // blocks per (window, head) for the pairs at index >= n_main
    int n_main;          // pairs [0, n_main) get one workgroup each; the tail pairs are split qsplit ways
    int table_len;
    int shifted;
#ifdef SOC_K1_STAMPS
    unsigned long long* dbg;  // diagnostic build only: [block][wave][32] s_memtime stamps
#endif
};

#ifdef SOC_K1_STAMPS
#define STAMP(slot)                                                                      \
    do {                                                                                 \
        if (lane == 0 && (slot) < 32)                                                    \
            p.dbg[((long)blockIdx.x * 8 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define STAMP_RT(slot)                                                                   \
    do {                                                                                 \
        if (lane == 0)                                                                   \
            p.dbg[((long)blockIdx.x * 8 + wave) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define STAMP_HWID(slot)                                                                 \
    do {                                                                                 \
        if (lane == 0)                                                                   \
            p.dbg[((long)blockIdx.x * 8 + wave) * 32 + (slot)] =                         \
                (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |          \
                ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);  \
    } while (0)
static unsigned long long* g_dbg = nullptr;
extern "C" void soc_debug_set_buffer(void* ptr) { g_dbg = (unsigned long long*)ptr; }
#else
#define STAMP(slot) do {} while (0)
#define STAMP_RT(slot) do {} while (0)
#define STAMP_HWID(slot) do {} while (0)
#endif

__device__ __forceinline__ int region1d(int c, int P, int w, int s) {
    // reference compute_mask slices: [0,P-w) -> 0, [P-w,P-s) -> 1, [P-s,P) -> 2; shift 0 -> uniform
    if (s == 0) return 0;
    return c < P - w ? 0 : (c < P - s ? 1 : 2);
}

template <int NT, int NT_PREV, bool SHIFTED>
__global__ __launch_bounds__(THREADS, 2) void win_attn3d_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
    const float* __restrict__ table, float* __restrict__ out, const WinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NP = NT * 16;
    float* Tb = reinterpret_cast<float*>(smem_raw);           // [table_len] bias column (x log2e)
    float* Ks = Tb + ((p.table_len + 3) & ~3);                 // [NP][RS]
    float* Vs = Ks + NP * RS;                                  // [NP][RS]
    int* src = reinterpret_cast<int*>(Vs + NP * RS);           // [NP] token offset or <0
    int* e4 = src + NP;                                        // [NP] 4 * e(i): byte offset into Tb
    int* rgn = e4 + NP;                                        // [NP] shift-mask region id
    int* wflag = rgn + NP;                                     // [8] per-wave "window has a mask"

    constexpr float LOG2E = 1.4426950408889634f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the last (pairs mod #CU) pairs would leave most CUs idle in the final round of workgroups, so
    // those tail pairs are split over several workgroups (each re-stages K/V for a slice of the tiles)
    int bid = blockIdx.x;
    int qpart = 0, qsplit = 1;
    if (bid >= p.n_main) {
        const int rem = bid - p.n_main;
        qsplit = p.qsplit;
        qpart = rem % qsplit;
        bid = p.n_main + rem / qsplit;
    }
    const int head = bid % p.nH; bid /= p.nH;
    const int wx = bid % p.nww; bid /= p.nww;
    const int wy = bid % p.nwh; bid /= p.nwh;
    const int wz = bid % p.nwd; bid /= p.nwd;
    const int b = bid;
    const int C3 = 3 * p.C;

    STAMP(0);
    // ---- stage 0: token metadata ------------------------------------------------------------
    int differs = 0;  // does any token of this window sit in another shift-mask region than token 0?
    const int reg0 = (region1d(wz * p.wd, p.Dp, p.wd, p.sd) * 3 + region1d(wy * p.wh, p.Hp, p.wh, p.sh)) * 3 +
                     region1d(wx * p.ww, p.Wp, p.ww, p.sw);
    for (int i = tid; i < NP; i += THREADS) {
        int s = -2, ecode = 0, reg = 0;
        if (i < p.N) {
            const int dz = i / (p.wh * p.ww), r = i - dz * (p.wh * p.ww);
            const int dy = r / p.ww, dx = r - dy * p.ww;
            const int zs = wz * p.wd + dz, ys = wy * p.wh + dy, xs = wx * p.ww + dx;  // shifted frame
            int z = zs + p.sd; if (z >= p.Dp) z -= p.Dp;
            int y = ys + p.sh; if (y >= p.Hp) y -= p.Hp;
            int x = xs + p.sw; if (x >= p.Wp) x -= p.Wp;
            s = (z < p.D && y < p.H && x < p.W) ? ((b * p.D + z) * p.H + y) * p.W + x : -1;
            // relative_position_index[:N,:N]: token i is decoded with the NOMINAL window dims
            const int tz = i / (p.th * p.tw), tr = i - tz * (p.th * p.tw);
            const int ty = tr / p.tw, tx = tr - ty * p.tw;
            ecode = 4 * ((tz * (2 * p.th - 1) + ty) * (2 * p.tw - 1) + tx);
            reg = (region1d(zs, p.Dp, p.wd, p.sd) * 3 + region1d(ys, p.Hp, p.wh, p.sh)) * 3 +
                  region1d(xs, p.Wp, p.ww, p.sw);
        }
        src[i] = s;
        e4[i] = ecode;
        rgn[i] = reg;
        differs |= (i < p.N && reg != reg0);
    }
    // scores are kept in log2 units (q and the bias column are pre-multiplied by log2 e) so the
    // softmax exponent is a bare v_exp_f32
    for (int i = tid; i < p.table_len; i += THREADS) Tb[i] = table[(long)i * p.nH + head] * LOG2E;
    // interior windows of a shifted block have a single region: their mask is all zero and the
    // per-element mask arithmetic is skipped (block-uniform branch)
    if (SHIFTED && lane == 0) wflag[wave] = 0;
    if (SHIFTED && __any(differs) && lane == 0) wflag[wave] = 1;
    __syncthreads();
    bool has_mask = false;
    if (SHIFTED) {
        int f = 0;
#pragma unroll
        for (int w8 = 0; w8 < THREADS / 64; ++w8) f |= wflag[w8];
        has_mask = f != 0;
    }
    STAMP(1);

    // ---- stage 1: K, V -> LDS (float4 per thread: 8 threads per token row) --------------------
    {
        const int part = tid & 7;
        const float4 kbias = *reinterpret_cast<const float4*>(qkv_bias + p.C + head * HD + part * 4);
        const float4 vbias = *reinterpret_cast<const float4*>(qkv_bias + 2 * p.C + head * HD + part * 4);
        constexpr int ROWS = THREADS / 8;                 // 64 token rows per pass
        constexpr int PASSES = (NP + ROWS - 1) / ROWS;
        float4 kv[PASSES], vv[PASSES];
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {             // issue every global load first
            const int i = it * ROWS + (tid >> 3);
            const int s = i < NP ? src[i] : -2;
            if (s >= 0) {
                const float* row = qkv + (long)s * C3 + head * HD + part * 4;
                kv[it] = *reinterpret_cast<const float4*>(row + p.C);
                vv[it] = *reinterpret_cast<const float4*>(row + 2 * p.C);
            } else if (s == -1) {
                kv[it] = kbias; vv[it] = vbias;
            } else {
                kv[it] = make_float4(0.f, 0.f, 0.f, 0.f); vv[it] = kv[it];
            }
        }
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int i = it * ROWS + (tid >> 3);
            if (i < NP) {
                *reinterpret_cast<float4*>(Ks + i * RS + part * 4) = kv[it];
                *reinterpret_cast<float4*>(Vs + i * RS + part * 4) = vv[it];
            }
        }
    }
    __syncthreads();
    STAMP(2);
    int stamp_slot = 3;
    (void)stamp_slot;

    // ---- stage 2/3: per 16-query tile ---------------------------------------------------------
    const int r = lane & 15;   // MFMA column: query inside the tile (also A-operand row)
    const int g = lane >> 4;   // MFMA k index / C row group
    const float scale = 0.17677669529663687f * LOG2E;  // 32^-0.5, in log2 units
    const int E0 = ((p.td - 1) * (2 * p.th - 1) + (p.th - 1)) * (2 * p.tw - 1) + (p.tw - 1);
    const int nwaves_total = (THREADS / 64) * qsplit;
    const float* kbase = Ks + r * RS + g;          // + 16t*RS + 4kk
    const float* vbase = Vs + (4 * g) * RS + r;    // + (16t + s)*RS (+16)

    // Q^T fragment (B operand): lane (r,g) holds q[token r][dim 4*kk+g] * scale; the next tile's
    // fragment is fetched while the current tile is in its softmax / PV phases.
    auto load_q = [&](int qt, float (&qf)[8], int& qsrc) {
        qsrc = src[qt * 16 + r];
        if (qsrc >= 0) {
            const float* qrow = qkv + (long)qsrc * C3 + head * HD + g;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = qrow[4 * kk];
        } else if (qsrc == -1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = qkv_bias[head * HD + 4 * kk + g];
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = 0.f;
        }
    };
    float qn[8];
    int qsrc_n = -2;
    int qt = qpart + qsplit * wave;      // part p of q takes tiles p, p+q, p+2q, ...; wave w the w-th, (w+8)-th, ... of them
    if (qt < p.NT) load_q(qt, qn, qsrc_n);
    for (; qt < p.NT; qt += nwaves_total) {
        const int qtok = qt * 16 + r;
        const int qsrc = qsrc_n;
        float qf[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = qn[kk] * scale;
        const int qe = e4[qtok] + 4 * E0;
        const int qreg = rgn[qtok];

        // S^T tiles: acc[t][i] = S[query r][key 16t + 4g + i].  k-step outermost: the NT MFMAs of
        // one k-step hit NT independent accumulators (no dependent-issue stalls) and the LDS reads
        // of the next k-step overlap them.
        // accumulators start from the relative-position bias: the LDS gather lands directly in the
        // MFMA C operand, so the bias costs no VALU add (f32 MFMA and VALU do not overlap here --
        // every VALU instruction saved in this loop is matrix-pipe time gained)
        f32x4 acc[NT];
        {
            const unsigned qaddr = (unsigned)(uintptr_t)(lds_cfloat*)Tb + (unsigned)qe;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int4 ke = *reinterpret_cast<const int4*>(e4 + 16 * t + 4 * g);
                acc[t][0] = *(lds_cfloat*)(uintptr_t)(qaddr - (unsigned)ke.x);
                acc[t][1] = *(lds_cfloat*)(uintptr_t)(qaddr - (unsigned)ke.y);
                acc[t][2] = *(lds_cfloat*)(uintptr_t)(qaddr - (unsigned)ke.z);
                acc[t][3] = *(lds_cfloat*)(uintptr_t)(qaddr - (unsigned)ke.w);
            }
        }
        {
            float a0[NT], a1[NT];  // K fragments, double-buffered across k-steps
#pragma unroll
            for (int t = 0; t < NT; ++t) a0[t] = kbase[16 * t * RS];
#pragma unroll
            for (int kk = 0; kk < 8; kk += 2) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a1[t] = kbase[16 * t * RS + 4 * (kk + 1)];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], qf[kk], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < 8) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) a0[t] = kbase[16 * t * RS + 4 * (kk + 2)];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], qf[kk + 1], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(stamp_slot); ++stamp_slot;
        // the K fragments are done with qf: fetch the next tile's Q now (hidden by softmax + PV)
        if (qt + nwaves_total < p.NT) load_q(qt + nwaves_total, qn, qsrc_n);

        // mask + row max + exp + row sum, all in registers (log2 units)
        if (has_mask) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int4 kr = *reinterpret_cast<const int4*>(rgn + 16 * t + 4 * g);
                const int krs[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[t][i] = (krs[i] != qreg) ? acc[t][i] - 100.0f * LOG2E : acc[t][i];
            }
        }
#pragma unroll
        for (int t = NT_PREV; t < NT; ++t)  // surplus keys of the last tile(s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * t + 4 * g + i >= p.N) acc[t][i] = -INFINITY;
        float mx = fmaxf(acc[0][0], acc[0][1]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t > 0) mx = __builtin_fmaxf(mx, fmaxf(acc[t][0], acc[t][1]));  // v_max3_f32
            mx = __builtin_fmaxf(mx, fmaxf(acc[t][2], acc[t][3]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const f32x4 mx4 = (f32x4){mx, mx, mx, mx};
        f32x2 sum2 = (f32x2){0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 d = acc[t] - mx4;  // 2 x v_pk_add_f32
            f32x4 e;
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(d[i]);
            acc[t] = e;
            sum2 += (f32x2){e[0], e[1]};
            sum2 += (f32x2){e[2], e[3]};
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        __builtin_amdgcn_s_setprio(0);
        STAMP(stamp_slot); ++stamp_slot;

        // O^T = V^T . P^T : A = V[key 16t+4g+s][dim 16*dt + r], B = acc[t][s]
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
        {
            float va[8], vb[8];  // V fragments of tile t / t+1: [s] dims 0-15, [4+s] dims 16-31
#pragma unroll
            for (int s = 0; s < 4; ++s) { va[s] = vbase[s * RS]; va[4 + s] = vbase[s * RS + 16]; }
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                if (t + 1 < NT) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        vb[s] = vbase[(16 * (t + 1) + s) * RS];
                        vb[4 + s] = vbase[(16 * (t + 1) + s) * RS + 16];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], acc[t][s], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[4 + s], acc[t][s], o1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NT) {
                    if (t + 2 < NT) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            va[s] = vbase[(16 * (t + 2) + s) * RS];
                            va[4 + s] = vbase[(16 * (t + 2) + s) * RS + 16];
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[s], acc[t + 1][s], o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[4 + s], acc[t + 1][s], o1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // C layout of O^T: column = query r, rows = dims 4g..4g+3 (o0) and 16+4g.. (o1)
        if (qsrc >= 0) {
            const float inv = 1.f / sum;
            float* orow = out + (long)qsrc * p.C + head * HD + 4 * g;
            *reinterpret_cast<float4*>(orow) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
            *reinterpret_cast<float4*>(orow + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
        }
        STAMP(stamp_slot); ++stamp_slot;
    }
}

// =============================================================================================
// Fast path for FULL windows (8 x 7 x 7 = 392 tokens, the only geometry of the BASELINE configs).
// Same algorithm as win_attn3d_kernel<25,..>, but the 392 tokens are enumerated with the TEMPORAL
// index fastest (slot = (dy*7 + dx)*8 + dz), which makes every group of 4 consecutive keys a run
// of 4 consecutive frames of one spatial position.  Consequences:
//   * the bias table is staged as T'[(dy_rel, dx_rel)][dz_rel]; the 4 biases of a key group are
//     4 consecutive LDS words -> ONE address subtraction + two ds_read2_b32 per group instead of
//     4 subtractions + 4 gathers + a code fetch; the per-(tile, lane) group codes are
//     query-independent and stay in 25 registers for the whole workgroup;
//   * shift-mask regions are constant inside a group (region boundaries are multiples of 4);
//   * the MFMA k-step kk uses head dims {8g + kk}: a lane's 8 K (and Q) operands of a key tile are
//     8 consecutive floats -> two ds_read_b128 (two global dwordx4 for Q) instead of 8 scalar reads;
//   * V is staged transposed ([dim][slot]) so a lane's 4 k-steps of a key tile are one ds_read_b128.
// LDS instructions per 16-query tile drop from ~525 to ~150, VALU address arithmetic from 100 to 25.
// =============================================================================================
// f32 MFMA executes on the FP32 vector lanes: while a wave streams v_mfma_f32_16x16x4_f32 back to
// back, a VALU-only wave on the same SIMD makes NO progress (tools/microbench/mfma_valu_overlap.hip:
// 32.0 cycles per MFMA alone and together; the VALU wave finishes exactly its stand-alone time
// later).  Every VALU instruction in the tile loop is therefore matrix time lost, so the softmax is
// written with the fewest possible vector instructions: 3-input max and packed subtract via
// inline asm (the compiler emits a canonicalising v_max per MFMA result and scalar subtracts).
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

constexpr int FN = 392, FNT = 25, FNP = 400;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int RSV = 404;        // row stride (floats) of the transposed V image [32][RSV]
constexpr int TBL = 169 * 15;   // 2535 entries
constexpr int PRS = 36;         // row stride of a wave's partial-output scratch (shared last tile)
constexpr int PWS = 16 * PRS + 32;  // floats per wave: O partial [16][PRS], row max [16], row sum [16]
constexpr size_t FULL_LDS_BYTES =
    (size_t)(2536 + FNP * RS + HD * RSV + (THREADS / 64) * PWS) * sizeof(float) + (2 * FNP + 8) * sizeof(int);

// One 16-query tile against ALL 25 key tiles: S^T = K.Q^T on top of the gathered bias, softmax in registers,
// O^T = V^T.P^T, normalise, scatter.  `qn` holds the (unscaled) Q fragment of this tile on entry and the next
// tile's on exit (prefetched behind the QK^T phase when next_qt >= 0).
template <bool SHIFTED, typename LoadQ>
__device__ __forceinline__ void full_tile(const int qt, const int next_qt, float (&qn)[8], int& qsrc_n,
                                          const float* Ks, const float* Vt, const int* qcd, const unsigned tbase,
                                          const int (&gcode)[FNT], const bool has_mask, const int r, const int g,
                                          const int head, const int C, float* __restrict__ out, LoadQ&& load_q) {
    const float scale = 0.17677669529663687f * LOG2E;
    // byte offset of T'[(0,0) rel][dz_rel = 0] in the reversed-temporal layout, minus the +8 bias of both codes
    const int C0 = ((6 * 13 + 6) * 15 + 7) * 4;
    const float* kb = Ks + r * RS + 8 * g;           // + 16t*RS : 8 consecutive dims of key 16t + r
    const float* vb = Vt + r * RSV + 4 * g;          // + 16t (+16*RSV): 4 consecutive keys of dim r
    const int qtok = qt * 16 + r;
    const int qsrc = qsrc_n;
    float qf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = qn[kk] * scale;
    const int qc = qcd[qtok];
    const unsigned qaddr = tbase + (unsigned)((qc & 0xFFFF) + C0);
    const int qreg = qc >> 16;

    // accumulators start from the bias: keys 16t+4g+i (i = 0..3) are frames 4(g&1)+i of one column = 4
    // ASCENDING words of the reversed table row, landing in the accumulator registers in order
    f32x4 acc[FNT];
#pragma unroll
    for (int t = 0; t < FNT; ++t) {
        lds_cfloat* bp = (lds_cfloat*)(uintptr_t)(qaddr - (unsigned)(gcode[t] & 0xFFFF));
        acc[t][0] = bp[0]; acc[t][1] = bp[1]; acc[t][2] = bp[2]; acc[t][3] = bp[3];
    }
    // S^T = K . Q^T, two key tiles interleaved so consecutive MFMAs hit different accumulators;
    // K fragments (2 x b128 per tile) double-buffered across tile pairs
    {
        float ka[2][8], kc[2][8];
        auto kload = [&](int t, float (&f)[8]) {
            const float4 a = *reinterpret_cast<const float4*>(kb + 16 * t * RS);
            const float4 c = *reinterpret_cast<const float4*>(kb + 16 * t * RS + 4);
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
        };
        kload(0, ka[0]); kload(1, ka[1]);
#pragma unroll
        for (int tp = 0; tp < FNT; tp += 4) {
            if (tp + 2 < FNT) kload(tp + 2, kc[0]);
            if (tp + 3 < FNT) kload(tp + 3, kc[1]);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[0][kk], qf[kk], acc[tp], 0, 0, 0);
                if (tp + 1 < FNT)
                    acc[tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[1][kk], qf[kk], acc[tp + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tp + 2 < FNT) {
                if (tp + 4 < FNT) kload(tp + 4, ka[0]);
                if (tp + 5 < FNT) kload(tp + 5, ka[1]);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    acc[tp + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[0][kk], qf[kk], acc[tp + 2], 0, 0, 0);
                    if (tp + 3 < FNT)
                        acc[tp + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[1][kk], qf[kk], acc[tp + 3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (next_qt >= 0) load_q(next_qt, qn, qsrc_n);  // next tile's Q, hidden behind softmax + PV

    if (has_mask) {
#pragma unroll
        for (int t = 0; t < FNT; ++t) {
            const float pen = ((gcode[t] >> 16) != qreg) ? -100.0f * LOG2E : 0.f;
            acc[t] += (f32x4){pen, pen, pen, pen};
        }
    }
    if (g >= 2) acc[FNT - 1] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // slots 392..399
    float mx = vmax3(acc[0][0], acc[0][1], acc[0][2]);
    mx = fmaxf(mx, acc[0][3]);
#pragma unroll
    for (int t = 1; t < FNT; ++t) {
        mx = vmax3(mx, acc[t][0], acc[t][1]);
        mx = vmax3(mx, acc[t][2], acc[t][3]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // softmax is shift-invariant: p = 2^(s - c) / sum 2^(s - c) for ANY c.  The reference's c = row max only
    // guards the exponent range, so when every row max of the wave is within +-96 (log2 units; f32 holds 2^+-126,
    // a row sums at most 400 terms and terms 2^24 below the row max do not matter)
    // c = 0 is used and the 100 subtractions are not executed -- VALU time is matrix-pipe time on this part.
    // Scores outside that range (possible with arbitrary weights) take the subtracting path.
    if (!__all(fabsf(mx) < 96.f)) {
        const f32x2 mx2 = (f32x2){mx, mx};
#pragma unroll
        for (int t = 0; t < FNT; ++t) {
            const f32x2 d0 = pk_sub((f32x2){acc[t][0], acc[t][1]}, mx2);
            const f32x2 d1 = pk_sub((f32x2){acc[t][2], acc[t][3]}, mx2);
            acc[t] = (f32x4){d0[0], d0[1], d1[0], d1[1]};
        }
    }
    f32x2 sum2 = (f32x2){0.f, 0.f};
#pragma unroll
    for (int t = 0; t < FNT; ++t) {
        f32x4 e;
        e[0] = __builtin_amdgcn_exp2f(acc[t][0]); e[1] = __builtin_amdgcn_exp2f(acc[t][1]);
        e[2] = __builtin_amdgcn_exp2f(acc[t][2]); e[3] = __builtin_amdgcn_exp2f(acc[t][3]);
        acc[t] = e;
        sum2 += (f32x2){e[0], e[1]};
        sum2 += (f32x2){e[2], e[3]};
    }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);

    // O^T = V^T . P^T: A = V^T[dim r (+16)][keys 16t+4g .. +3] (one b128 each), B = acc[t][s]
    f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
    {
        float4 va0, va1, vc0, vc1;
        va0 = *reinterpret_cast<const float4*>(vb);
        va1 = *reinterpret_cast<const float4*>(vb + 16 * RSV);
#pragma unroll
        for (int t = 0; t < FNT; t += 2) {
            if (t + 1 < FNT) {
                vc0 = *reinterpret_cast<const float4*>(vb + 16 * (t + 1));
                vc1 = *reinterpret_cast<const float4*>(vb + 16 * (t + 1) + 16 * RSV);
            }
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va0.x, acc[t][0], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va1.x, acc[t][0], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va0.y, acc[t][1], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va1.y, acc[t][1], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va0.z, acc[t][2], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va1.z, acc[t][2], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va0.w, acc[t][3], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va1.w, acc[t][3], o1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < FNT) {
                if (t + 2 < FNT) {
                    va0 = *reinterpret_cast<const float4*>(vb + 16 * (t + 2));
                    va1 = *reinterpret_cast<const float4*>(vb + 16 * (t + 2) + 16 * RSV);
                }
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc0.x, acc[t + 1][0], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc1.x, acc[t + 1][0], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc0.y, acc[t + 1][1], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc1.y, acc[t + 1][1], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc0.z, acc[t + 1][2], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc1.z, acc[t + 1][2], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc0.w, acc[t + 1][3], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vc1.w, acc[t + 1][3], o1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (qsrc >= 0) {
        const float inv = 1.f / sum;
        float* orow = out + (long)qsrc * C + head * HD + 4 * g;
        *reinterpret_cast<float4*>(orow) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
        *reinterpret_cast<float4*>(orow + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
    }
}

template <bool SHIFTED>
__global__ __launch_bounds__(THREADS, 2) void win_attn3d_full_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
    const float* __restrict__ table, float* __restrict__ out, const WinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Tb = reinterpret_cast<float*>(smem_raw);            // [2535 (+1 pad)] T'[(yy,xx)][14 - zz] * log2e
    float* Ks = Tb + 2536;                                      // [400][RS]   K, slot-major
    float* Vt = Ks + FNP * RS;                                  // [32][RSV]   V transposed
    float* Pw = Vt + HD * RSV;                                  // [8][PWS]    per-wave partials of the shared last tile
    int* src = reinterpret_cast<int*>(Pw + (THREADS / 64) * PWS);  // [400] token row offset or <0
    int* qcd = src + FNP;                                       // [400] 4*(c(q)) | region << 16
    int* wflag = qcd + FNP;                                     // [8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // pairs [0, n_main) get one workgroup each; the pairs behind them are split qsplit ways over query tiles
    // (plan_schedule(): the last round of workgroups would otherwise leave most CUs idle)
    int bid = blockIdx.x;
    int qpart = 0, qsplit = 1;
    if (bid >= p.n_main) {
        const int rem = bid - p.n_main;
        qsplit = p.qsplit;
        qpart = rem % qsplit;
        bid = p.n_main + rem / qsplit;
    }
    const int head = bid % p.nH; bid /= p.nH;
    const int wx = bid % p.nww; bid /= p.nww;
    const int wy = bid % p.nwh; bid /= p.nwh;
    const int wz = bid % p.nwd; bid /= p.nwd;
    const int b = bid;
    const int C3 = 3 * p.C;
    const int r = lane & 15, g = lane >> 4;
    STAMP_RT(30);
    STAMP_HWID(29);
    STAMP(0);

    // slot = col*8 + dz, col = dy*7 + dx (temporal index fastest)
    auto slot_info = [&](int i, int& reg, int& ccode) -> int {
        const int col = i >> 3, dz = i & 7;
        const int dy = col / 7, dx = col - dy * 7;
        const int zs = wz * 8 + dz, ys = wy * 7 + dy, xs = wx * 7 + dx;  // shifted frame
        int z = zs + p.sd; if (z >= p.Dp) z -= p.Dp;
        int y = ys + p.sh; if (y >= p.Hp) y -= p.Hp;
        int x = xs + p.sw; if (x >= p.Wp) x -= p.Wp;
        reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
        ccode = ((dy * 13 + dx) * 15 - dz + 8) * 4;  // byte code of the slot in the reversed-temporal table
        return (z < p.D && y < p.H && x < p.W) ? ((b * p.D + z) * p.H + y) * p.W + x : -1;
    };

    // ---- stage 0+1: every thread derives the metadata of the K/V rows it stages (8 threads x float4 per row)
    // and issues their global loads at once -- all K rows first, then all V rows, then the bias column.
    // K -> [slot][36], V -> transposed [dim][404], bias column and metadata go to LDS behind ONE barrier.
    // (Holding V back until the first tile's Q.K^T has run -- loads issued early or late -- shortens this stage by
    // 4k cycles and lengthens the tile phase by the same 4k: measured null, tools/experiments/README.md.)
    const int spart = tid & 7;
    constexpr int SROWS = THREADS / 8, SPASSES = (FNP + SROWS - 1) / SROWS;
    float4 vv[SPASSES];
    {
        const float4 kbias = *reinterpret_cast<const float4*>(qkv_bias + p.C + head * HD + spart * 4);
        const float4 vbias = *reinterpret_cast<const float4*>(qkv_bias + 2 * p.C + head * HD + spart * 4);
        float4 kv[SPASSES];
        int sl[SPASSES];
        int differs = 0;
        int reg0, c0;
        (void)slot_info(0, reg0, c0);
#pragma unroll
        for (int it = 0; it < SPASSES; ++it) {
            const int i = it * SROWS + (tid >> 3);
            int s = -2, reg = 0, cc = 0;
            if (i < FN) {
                s = slot_info(i, reg, cc);
                differs |= (reg != reg0);
            }
            if (i < FNP && spart == 0) {
                src[i] = s;
                qcd[i] = cc | (reg << 16);
            }
            sl[it] = s;
            if (s >= 0) kv[it] = *reinterpret_cast<const float4*>(qkv + (long)s * C3 + head * HD + spart * 4 + p.C);
            else kv[it] = s == -1 ? kbias : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < SPASSES; ++it) {
            const int s = sl[it];
            if (s >= 0) vv[it] = *reinterpret_cast<const float4*>(qkv + (long)s * C3 + head * HD + spart * 4 + 2 * p.C);
            else vv[it] = s == -1 ? vbias : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // bias column with the temporal offset fastest and REVERSED: word yx*15 + (14 - zz) holds
        // table[(zz*169 + yx)][head], so the 4 frames kz0..kz0+3 of a key group are 4 ascending words
        constexpr int TPASS = (TBL + THREADS - 1) / THREADS;
        float tv[TPASS];
#pragma unroll
        for (int it = 0; it < TPASS; ++it) {
            const int i = it * THREADS + tid;
            const int yx = i / 15, zz = i - yx * 15;
            tv[it] = i < TBL ? table[(long)(zz * 169 + yx) * p.nH + head] : 0.f;
        }
        const int wave_differs = __any(differs);        // all 64 lanes vote (not inside the lane-0 branch)
        if (SHIFTED && lane == 0) wflag[wave] = wave_differs ? 1 : 0;
#pragma unroll
        for (int it = 0; it < SPASSES; ++it) {
            const int i = it * SROWS + (tid >> 3);
            if (i < FNP) *reinterpret_cast<float4*>(Ks + i * RS + spart * 4) = kv[it];
        }
#pragma unroll
        for (int it = 0; it < TPASS; ++it) {
            const int i = it * THREADS + tid;
            const int yx = i / 15, zz = i - yx * 15;
            if (i < TBL) Tb[yx * 15 + 14 - zz] = tv[it] * LOG2E;
        }
    }
#pragma unroll
    for (int it = 0; it < SPASSES; ++it) {
        const int i = it * SROWS + (tid >> 3);
        if (i < FNP) {
            Vt[(spart * 4 + 0) * RSV + i] = vv[it].x;
            Vt[(spart * 4 + 1) * RSV + i] = vv[it].y;
            Vt[(spart * 4 + 2) * RSV + i] = vv[it].z;
            Vt[(spart * 4 + 3) * RSV + i] = vv[it].w;
        }
    }
    STAMP(1);
    // per-(tile, lane) key-group codes: group = slots 16t+4g..+3 = column 2t + (g>>1), frames 4(g&1)..+3
    auto group_code = [&](int t) -> int {
        const int col = 2 * t + (g >> 1);
        const int dy = col / 7, dx = col - dy * 7;
        int reg = 0;
        if (SHIFTED) {
            const int zs = wz * 8 + 4 * (g & 1), ys = wy * 7 + dy, xs = wx * 7 + dx;
            reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
        }
        return (((dy * 13 + dx) * 15 - 4 * (g & 1) + 8) * 4) | (reg << 16);
    };
    int gcode[FNT];
#pragma unroll
    for (int t = 0; t < FNT; ++t) gcode[t] = group_code(t);
    __syncthreads();
    bool has_mask = false;
    if (SHIFTED) {
        int f = 0;
#pragma unroll
        for (int w8 = 0; w8 < THREADS / 64; ++w8) f |= wflag[w8];
        has_mask = f != 0;
    }
    STAMP(2);

    // ---- stage 2/3: 16-query tiles ---------------------------------------------------------------
    const unsigned tbase = (unsigned)(uintptr_t)(lds_cfloat*)Tb;
    auto load_q = [&](int qt, float (&qf)[8], int& qsrc) {
        qsrc = src[qt * 16 + r];
        if (qsrc >= 0) {
            const float4* qrow = reinterpret_cast<const float4*>(qkv + (long)qsrc * C3 + head * HD + 8 * g);
            const float4 a = qrow[0], c = qrow[1];
            qf[0] = a.x; qf[1] = a.y; qf[2] = a.z; qf[3] = a.w; qf[4] = c.x; qf[5] = c.y; qf[6] = c.z; qf[7] = c.w;
        } else if (qsrc == -1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = qkv_bias[head * HD + 8 * g + kk];
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = 0.f;
        }
    };
    // An unsplit workgroup deals tiles 0..23 three to a wave and SHARES the 25th (8 real queries): every wave
    // takes 3-4 of its 25 key tiles and the partial results are merged through LDS -- 6.25 tile-times per SIMD
    // instead of 7/6/6/6.  Split workgroups (a part of the tiles each) keep whole tiles.
    const bool share_last = qsplit == 1;
    const int ntile = share_last ? FNT - 1 : FNT;
    const int stride = (THREADS / 64) * qsplit;
    float qn[8];
    int qsrc_n = -2;
    int qt = qpart + qsplit * wave;   // part p of q owns tiles p, p+q, ...; wave w the w-th, (w+8)-th, ... of them
    if (qt < ntile) load_q(qt, qn, qsrc_n);

    for (; qt < ntile; qt += stride)
        full_tile<SHIFTED>(qt, qt + stride < ntile ? qt + stride : -1, qn, qsrc_n, Ks, Vt, qcd, tbase, gcode, has_mask,
                           r, g, head, p.C, out, load_q);
    STAMP(3);

    if (share_last) {
        constexpr int QT = FNT - 1;
        const float scale = 0.17677669529663687f * LOG2E;
        const int C0 = ((6 * 13 + 6) * 15 + 7) * 4;
        float qf[8];
        int qsrc;
        load_q(QT, qf, qsrc);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] *= scale;
        const int qc = qcd[QT * 16 + r];
        const unsigned qaddr = tbase + (unsigned)((qc & 0xFFFF) + C0);
        const int qreg = qc >> 16;
        const float* kb = Ks + r * RS + 8 * g;
        const float* vb = Vt + r * RSV + 4 * g;
        constexpr int NJ = (FNT + THREADS / 64 - 1) / (THREADS / 64);   // key tiles per wave: kt = wave + 8j
        f32x4 a4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kt = wave + (THREADS / 64) * j;          // wave-uniform
            a4[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt < FNT) {
                const int gc = group_code(kt);
                lds_cfloat* bp = (lds_cfloat*)(uintptr_t)(qaddr - (unsigned)(gc & 0xFFFF));
                f32x4 a = (f32x4){bp[0], bp[1], bp[2], bp[3]};
                const float4 k0 = *reinterpret_cast<const float4*>(kb + 16 * kt * RS);
                const float4 k1 = *reinterpret_cast<const float4*>(kb + 16 * kt * RS + 4);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[3], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[4], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[5], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[6], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[7], a, 0, 0, 0);
                if (has_mask) {
                    const float pen = ((gc >> 16) != qreg) ? -100.0f * LOG2E : 0.f;
                    a += (f32x4){pen, pen, pen, pen};
                }
                if (kt == FNT - 1 && g >= 2) a = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // slots 392..399
                a4[j] = a;
            }
        }
        // this wave's keys: local max / exp / sum (merged below with the other waves' by the usual rescaling)
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) mx = fmaxf(fmaxf(mx, fmaxf(a4[j][0], a4[j][1])), fmaxf(a4[j][2], a4[j][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a4[j][i] = __builtin_amdgcn_exp2f(a4[j][i] - mx);
                sum += a4[j][i];
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kt = wave + (THREADS / 64) * j;
            if (kt < FNT) {
                const float4 v0 = *reinterpret_cast<const float4*>(vb + 16 * kt);
                const float4 v1 = *reinterpret_cast<const float4*>(vb + 16 * kt + 16 * RSV);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.x, a4[j][0], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.x, a4[j][0], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.y, a4[j][1], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.y, a4[j][1], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.z, a4[j][2], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.z, a4[j][2], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.w, a4[j][3], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.w, a4[j][3], o1, 0, 0, 0);
            }
        }
        // partial O^T: column = query r, rows = dims 4g..4g+3 (o0) and 16+4g.. (o1)
        float* pw = Pw + wave * PWS;
        *reinterpret_cast<float4*>(pw + r * PRS + 4 * g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(pw + r * PRS + 16 + 4 * g) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        if (g == 0) {
            pw[16 * PRS + r] = mx;
            pw[16 * PRS + 16 + r] = sum;
        }
        __syncthreads();
        // merge: thread = (query, dim); only queries 0..7 of the tile exist (slots 384..391)
        const int q = tid >> 5, d = tid & 31;
        const int osrc = q < FN - QT * 16 ? src[QT * 16 + q] : -1;
        if (osrc >= 0) {
            float M = -INFINITY;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) M = fmaxf(M, Pw[w8 * PWS + 16 * PRS + q]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) {
                const float f = __builtin_amdgcn_exp2f(Pw[w8 * PWS + 16 * PRS + q] - M);
                num += f * Pw[w8 * PWS + q * PRS + d];
                den += f * Pw[w8 * PWS + 16 * PRS + 16 + q];
            }
            out[(long)osrc * p.C + head * HD + d] = num / den;
        }
    }
    STAMP(4);
    STAMP_RT(31);
}

// =============================================================================================
// K1 on the bf16 matrix cores (round 3).  The f32-input MFMA above runs at the f32 vector rate (1/16 of the bf16 matrix
// rate) and blocks the vector ALU while it runs; this form keeps f32 semantics -- every f32 operand is split EXACTLY
// into three bf16 numbers (8 + 8 + 8 significant bits; csrc/linear_split.hip has the arithmetic), six of the nine
// bf16 x bf16 products (each exact in f32) are accumulated in the f32 accumulators, the dropped three are <= 2^-23 |a b| --
// at 6/16 of the matrix time and with the softmax's vector work running BESIDE the matrix cores instead of in front of
// them.  Same algorithm and window enumeration as win_attn3d_full_kernel; what changes:
//   * K and V of the head are split while they are staged: three planes each of [slot][32 dims] bf16 (64-B rows), 16-B
//     chunk c of row r at chunk c ^ pi[(r >> 2) & 3], pi = (0, 3, 2, 1): the per-lane ds_read_b128 of the K operand
//     (lane = key, 8 consecutive dims) and the ds_read_b64_tr_b16 of the V operand are both bank-conflict free, and V needs
//     no transposed image -- the transposing read hands a lane 4 keys of one dim;
//   * S^T = K . Q^T is ONE v_mfma_f32_16x16x32_bf16 per key tile and product (head dim 32 = one k-step), Q split per tile;
//   * P is split in registers after the softmax; O^T = V^T . P^T takes two key tiles per MFMA (k = 32 keys: a lane's 8
//     P values are its 4 + 4 accumulator registers of the two tiles, the V side reads the matching 4 + 4 keys);
//   * nothing per-slot is kept in LDS (token offsets and bias codes are recomputed where needed): 160.7 KB of planes +
//     bias column leave room for nothing else.
// =============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KROWB = 64;                          // bytes per plane row: 32 bf16
constexpr int PLANEB = FN * KROWB;                 // 25 088 B per plane (392 slots; tile 24 reads 8 rows past the end)
constexpr int SP_TB = 0;                           // [2536] f32 bias column
constexpr int SP_K = 2536 * 4;                     // 3 K planes
constexpr int SP_V = SP_K + 3 * PLANEB;            // 3 V planes
constexpr int SP_MISC = SP_V + 3 * PLANEB;         // wflag[8], gtab[25][4]
constexpr size_t SPLIT_LDS_BYTES = SP_MISC + 8 * 4 + FNT * 4 * 4 + 512;   // + slack: tile 24 of V plane 2 reads 512 B past the planes
static_assert(SPLIT_LDS_BYTES <= 160 * 1024, "LDS");
static_assert((THREADS / 64) * PWS * 4 <= 3 * PLANEB, "shared-tile scratch aliases the K planes");

__device__ __forceinline__ void split8v(const float (&v)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        h0[i] = a0; h1[i] = a1; h2[i] = (__bf16)r2;
    }
}
__device__ __forceinline__ void split4v(const f32x4& v, bf16x4& h0, bf16x4& h1, bf16x4& h2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        h0[i] = a0; h1[i] = a1; h2[i] = (__bf16)r2;
    }
}
__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }     // pi[(row >> 2) & 3]

// six products of the three-way split, smallest first
#define MFMA6_32(acc, a, b)                                                                   \
    do {                                                                                      \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);              \
    } while (0)
#define MFMA6_16(acc, a, b)                                                                   \
    do {                                                                                      \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[2], b[0], acc, 0, 0, 0);            \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[1], acc, 0, 0, 0);            \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[2], acc, 0, 0, 0);            \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[0], acc, 0, 0, 0);            \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[1], acc, 0, 0, 0);            \
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);            \
    } while (0)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// The transposing read: a group of 16 lanes (one k-chunk g of the MFMA) reads a block of 4 keys x 16 dims of a V plane; lane
// li = 4 q + pp of the group supplies the address of key row q, dims 4 pp .. 4 pp + 3, and receives dim li of the 4 keys.
// For key rows 16 t + 4 g + q the swizzle term pi[(row >> 2) & 3] is pi[g] whatever t is, so the address is a per-lane base
// (one per plane and 16-dim half) plus the compile-time offset 1024 t.
__device__ __forceinline__ bf16x4 tr_read(const unsigned addr) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)addr));
}

template <bool SHIFTED>
__global__ __launch_bounds__(THREADS, 2) void win_attn3d_split_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
    const float* __restrict__ table, float* __restrict__ out, const WinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // Like K20 the kernel owns its CUs (all 256 architectural VGPRs x 2 waves per SIMD = the whole register file): beside
    // its bf16 MFMA + LDS waves, v_pk_fma_f32 with an SGPR source in a neighbouring kernel's wave returned wrong values in
    // lanes 48..63 (the dynamic mask head in the pipelined replay).
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    float* Tb = reinterpret_cast<float*>(smem_raw + SP_TB);
    char* Kp = smem_raw + SP_K;
    int* wflag = reinterpret_cast<int*>(smem_raw + SP_MISC);
    int* gtab = wflag + 8;                                      // [25][4] key-group codes: 4 * c | region << 16
    float* Pw = reinterpret_cast<float*>(Kp);                   // shared-tile scratch, aliases the K planes (behind a barrier)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    int qpart = 0, qsplit = 1;
    if (bid >= p.n_main) {
        const int rem = bid - p.n_main;
        qsplit = p.qsplit;
        qpart = rem % qsplit;
        bid = p.n_main + rem / qsplit;
    }
    const int head = bid % p.nH; bid /= p.nH;
    const int wx = bid % p.nww; bid /= p.nww;
    const int wy = bid % p.nwh; bid /= p.nwh;
    const int wz = bid % p.nwd; bid /= p.nwd;
    const int b = bid;
    const int C3 = 3 * p.C;
    const int r = lane & 15, g = lane >> 4;
    STAMP_RT(30);
    STAMP_HWID(29);
    STAMP(0);

    // slot = col*8 + dz, col = dy*7 + dx (temporal index fastest)
    auto slot_info = [&](int i, int& reg, int& ccode) -> int {
        const int col = i >> 3, dz = i & 7;
        const int dy = (col * 37) >> 8, dx = col - dy * 7;     // col / 7 for col < 64
        const int zs = wz * 8 + dz, ys = wy * 7 + dy, xs = wx * 7 + dx;  // shifted frame
        int z = zs + p.sd; if (z >= p.Dp) z -= p.Dp;
        int y = ys + p.sh; if (y >= p.Hp) y -= p.Hp;
        int x = xs + p.sw; if (x >= p.Wp) x -= p.Wp;
        reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
        ccode = ((dy * 13 + dx) * 15 - dz + 8) * 4;
        return (z < p.D && y < p.H && x < p.W) ? ((b * p.D + z) * p.H + y) * p.W + x : -1;
    };

    // ---- staging: K and V rows (8 dims per thread) split into three bf16 planes each, bias column, group table
    {
        int differs = 0, reg0, c0;
        (void)slot_info(0, reg0, c0);
        constexpr int ITEMS = FN * 4;                       // (slot, 8-dim chunk)
        constexpr int PASSES = (ITEMS + THREADS - 1) / THREADS;
        float4 kv[PASSES][2], vv[PASSES][2];
        int ss[PASSES];
        const int c = tid & 3;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int i = (tid + THREADS * it) >> 2;
            int s = -2, reg = 0, cc = 0;
            if (i < FN) {
                s = slot_info(i, reg, cc);
                differs |= (reg != reg0);
            }
            ss[it] = s;
            if (s >= 0) {
                const float4* row = reinterpret_cast<const float4*>(qkv + (long)s * C3 + head * HD + 8 * c);
                kv[it][0] = row[p.C / 4]; kv[it][1] = row[p.C / 4 + 1];
                vv[it][0] = row[p.C / 2]; vv[it][1] = row[p.C / 2 + 1];
            } else {
                const float4* kb = reinterpret_cast<const float4*>(qkv_bias + p.C + head * HD + 8 * c);
                const float4* vb = reinterpret_cast<const float4*>(qkv_bias + 2 * p.C + head * HD + 8 * c);
                kv[it][0] = kb[0]; kv[it][1] = kb[1];
                vv[it][0] = vb[0]; vv[it][1] = vb[1];
            }
        }
        constexpr int TPASS = (TBL + THREADS - 1) / THREADS;
        float tv[TPASS];
#pragma unroll
        for (int it = 0; it < TPASS; ++it) {
            const int i = it * THREADS + tid;
            const int yx = i / 15, zz = i - yx * 15;
            tv[it] = i < TBL ? table[(long)(zz * 169 + yx) * p.nH + head] : 0.f;
        }
        const int wave_differs = __any(differs);
        if (SHIFTED && lane == 0) wflag[wave] = wave_differs ? 1 : 0;
        STAMP(1);
        if (tid < FNT * 4) {                                // key-group codes: group = slots 16t + 4gg .. +3
            const int t = tid >> 2, gg = tid & 3;
            const int col = 2 * t + (gg >> 1);
            const int dy = (col * 37) >> 8, dx = col - dy * 7;
            int reg = 0;
            if (SHIFTED) {
                const int zs = wz * 8 + 4 * (gg & 1), ys = wy * 7 + dy, xs = wx * 7 + dx;
                reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
            }
            gtab[tid] = (((dy * 13 + dx) * 15 - 4 * (gg & 1) + 8) * 4) | (reg << 16);
        }
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int i = (tid + THREADS * it) >> 2;
            if (i < FN) {
                const float kf[8] = {kv[it][0].x, kv[it][0].y, kv[it][0].z, kv[it][0].w, kv[it][1].x, kv[it][1].y, kv[it][1].z, kv[it][1].w};
                const float vf[8] = {vv[it][0].x, vv[it][0].y, vv[it][0].z, vv[it][0].w, vv[it][1].x, vv[it][1].y, vv[it][1].z, vv[it][1].w};
                bf16x8 h0, h1, h2;
                char* dst = Kp + i * KROWB + ((c ^ swz(i)) << 4);
                split8v(kf, h0, h1, h2);
                *reinterpret_cast<bf16x8*>(dst) = h0;
                *reinterpret_cast<bf16x8*>(dst + PLANEB) = h1;
                *reinterpret_cast<bf16x8*>(dst + 2 * PLANEB) = h2;
                split8v(vf, h0, h1, h2);
                *reinterpret_cast<bf16x8*>(dst + 3 * PLANEB) = h0;
                *reinterpret_cast<bf16x8*>(dst + 4 * PLANEB) = h1;
                *reinterpret_cast<bf16x8*>(dst + 5 * PLANEB) = h2;
            }
        }
#pragma unroll
        for (int it = 0; it < TPASS; ++it) {
            const int i = it * THREADS + tid;
            const int yx = i / 15, zz = i - yx * 15;
            if (i < TBL) Tb[yx * 15 + 14 - zz] = tv[it] * LOG2E;
        }
    }
    STAMP(2);
    __syncthreads();
    bool has_mask = false;
    if (SHIFTED) {
        int f = 0;
#pragma unroll
        for (int w8 = 0; w8 < THREADS / 64; ++w8) f |= wflag[w8];
        has_mask = f != 0;
    }
    STAMP(3);
    int stamp_slot = 4;
    (void)stamp_slot;

    const unsigned tbase = (unsigned)(uintptr_t)(lds_cfloat*)Tb;
    const unsigned kaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Kp;
    const unsigned vaddr = kaddr + 3 * PLANEB;
    // The scale lives in a vector register: as an SGPR operand of v_pk_fma_f32 it would be the form that returns wrong
    // lanes beside waves mixing bf16 MFMAs with LDS traffic (tools/experiments/pk_mfma_probe.hip) -- this kernel's own.
    float scale;
    asm volatile("v_mov_b32 %0, %1" : "=v"(scale) : "v"(0.17677669529663687f * LOG2E));
    const int C0 = ((6 * 13 + 6) * 15 + 7) * 4;
    // K operand of key tile t, plane pl: lane (key r, dims 8g .. 8g+7); + 1024 t
    unsigned kfrag[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) kfrag[pl] = kaddr + (unsigned)(pl * PLANEB + r * KROWB + ((g ^ swz(r)) << 4));
    // V operand bases: lane li = r of group g supplies key row 4 g + (r >> 2), dims 16 dt + 4 (r & 3) ..; + 1024 t
    unsigned vfrag[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int row = 4 * g + (r >> 2), pp = r & 3;
            vfrag[pl][dt] = vaddr + (unsigned)(pl * PLANEB + row * KROWB + (((2 * dt + (pp >> 1)) ^ swz(row)) << 4) + 8 * (pp & 1));
        }
    // slots 392 .. 399 of the last key tile do not exist (k-chunks g >= 2): their P is 0, but 0 x (whatever bytes follow
    // the plane) could be NaN -- those lanes re-read tile 23 instead ("pad, don't mask": the transposing read needs every lane)
    const unsigned last_off = (unsigned)((g >= 2 ? FNT - 2 : FNT - 1) * 16 * KROWB);

    // Q fragment of a tile: lane (query r, dims 8g .. 8g+7), scaled, split; + the query's token / bias code / region
    auto load_q = [&](int qt, float (&qf)[8], int& qsrc, int& qcode) {
        int reg, cc;
        const int slot = qt * 16 + r;
        qsrc = -2; qcode = 0;
        if (slot < FN) {
            qsrc = slot_info(slot, reg, cc);
            qcode = cc | (reg << 16);
        }
        if (qsrc >= 0) {
            const float4* qrow = reinterpret_cast<const float4*>(qkv + (long)qsrc * C3 + head * HD + 8 * g);
            const float4 a = qrow[0], c = qrow[1];
            qf[0] = a.x; qf[1] = a.y; qf[2] = a.z; qf[3] = a.w; qf[4] = c.x; qf[5] = c.y; qf[6] = c.z; qf[7] = c.w;
        } else if (qsrc == -1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = qkv_bias[head * HD + 8 * g + kk];
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = 0.f;
        }
    };

    // scores of key tile kt for the query fragment qs: bias gather + 6 MFMAs (+ mask)
    auto score_tile = [&](int kt, const bf16x8 (&qs)[3], unsigned qaddr, int qreg) -> f32x4 {
        const int gc = gtab[kt * 4 + g];
        lds_cfloat* bp = (lds_cfloat*)(uintptr_t)(qaddr - (unsigned)(gc & 0xFFFF));
        f32x4 a = (f32x4){bp[0], bp[1], bp[2], bp[3]};
        bf16x8 kf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            kf[pl] = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(kfrag[pl] + (unsigned)(kt * 16 * KROWB));
        MFMA6_32(a, kf, qs);
        if (has_mask) {
            const float pen = ((gc >> 16) != qreg) ? -100.0f * LOG2E : 0.f;
            a += (f32x4){pen, pen, pen, pen};
        }
        return a;
    };

    const bool share_last = qsplit == 1;
    const int ntile = share_last ? FNT - 1 : FNT;
    const int stride = (THREADS / 64) * qsplit;
    float qn[8];
    int qsrc_n = -2, qcode_n = 0;
    int qt = qpart + qsplit * wave;
    if (qt < ntile) load_q(qt, qn, qsrc_n, qcode_n);

    for (; qt < ntile; qt += stride) {
        const int qsrc = qsrc_n;
        const unsigned qaddr = tbase + (unsigned)((qcode_n & 0xFFFF) + C0);
        const int qreg = qcode_n >> 16;
        bf16x8 qs[3];
        {
            float qf[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) qf[kk] = qn[kk] * scale;
            split8v(qf, qs[0], qs[1], qs[2]);
        }
        f32x4 acc[FNT];
#pragma unroll
        for (int t = 0; t < FNT; ++t) {
            acc[t] = score_tile(t, qs, qaddr, qreg);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);     // two key tiles per scheduling window: the LDS reads of one
        }                                                     // overlap the MFMAs of the other, no more fragments live
        STAMP(stamp_slot); ++stamp_slot;
        if (qt + stride < ntile) load_q(qt + stride, qn, qsrc_n, qcode_n);   // next tile's Q, hidden behind softmax + PV

        if (g >= 2) acc[FNT - 1] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // slots 392..399
        float mx = vmax3(acc[0][0], acc[0][1], acc[0][2]);
        mx = fmaxf(mx, acc[0][3]);
#pragma unroll
        for (int t = 1; t < FNT; ++t) {
            mx = vmax3(mx, acc[t][0], acc[t][1]);
            mx = vmax3(mx, acc[t][2], acc[t][3]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // softmax is shift-invariant; the subtraction is only needed when 2^score could leave the f32 range
        if (!__all(fabsf(mx) < 96.f)) {
#pragma unroll
            for (int t = 0; t < FNT; ++t) acc[t] -= (f32x4){mx, mx, mx, mx};
        }
        STAMP(stamp_slot); ++stamp_slot;
        float sum = 0.f;
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
        // exp, row sums, split of P and O^T = V^T . P^T, two key tiles (32 keys) per MFMA k-step
#pragma unroll
        for (int t = 0; t < FNT - 1; t += 2) {
            float pv[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pv[i] = __builtin_amdgcn_exp2f(acc[t][i]);
                pv[4 + i] = __builtin_amdgcn_exp2f(acc[t + 1][i]);
            }
            sum += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            bf16x8 ps[3];
            split8v(pv, ps[0], ps[1], ps[2]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x8 vf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x4 lo = tr_read(vfrag[pl][dt] + (unsigned)(t * 16 * KROWB));
                    const bf16x4 hi = tr_read(vfrag[pl][dt] + (unsigned)((t + 1) * 16 * KROWB));
                    vf[pl] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                if (dt == 0) MFMA6_32(o0, vf, ps); else MFMA6_32(o1, vf, ps);
            }
            __builtin_amdgcn_sched_barrier(0);                // one key-tile pair per scheduling window
        }
        {   // the 25th key tile alone: k = 16
            constexpr int t = FNT - 1;
            f32x4 pe;
#pragma unroll
            for (int i = 0; i < 4; ++i) pe[i] = __builtin_amdgcn_exp2f(acc[t][i]);
            sum += (pe[0] + pe[1]) + (pe[2] + pe[3]);
            bf16x4 ps[3];
            split4v(pe, ps[0], ps[1], ps[2]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x4 vf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) vf[pl] = tr_read(vfrag[pl][dt] + last_off);
                if (dt == 0) MFMA6_16(o0, vf, ps); else MFMA6_16(o1, vf, ps);
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (qsrc >= 0) {
            const float inv = 1.f / sum;
            float* orow = out + (long)qsrc * p.C + head * HD + 4 * g;
            *reinterpret_cast<float4*>(orow) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
            *reinterpret_cast<float4*>(orow + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
        }
        STAMP(stamp_slot); ++stamp_slot;
    }

    if (share_last) {
        // the 25th query tile (8 real queries) is shared: wave w takes key tiles w, w + 8, ...; partial (O, max, sum) are
        // merged through LDS (the scratch aliases the K planes: every wave must be done with them first)
        constexpr int QT = FNT - 1;
        float qf[8];
        int qsrc, qcode;
        load_q(QT, qf, qsrc, qcode);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] *= scale;
        bf16x8 qs[3];
        split8v(qf, qs[0], qs[1], qs[2]);
        const unsigned qaddr = tbase + (unsigned)((qcode & 0xFFFF) + C0);
        const int qreg = qcode >> 16;
        constexpr int NJ = (FNT + THREADS / 64 - 1) / (THREADS / 64);
        f32x4 a4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kt = wave + (THREADS / 64) * j;          // wave-uniform
            a4[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt < FNT) {
                f32x4 a = score_tile(kt, qs, qaddr, qreg);
                if (kt == FNT - 1 && g >= 2) a = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                a4[j] = a;
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) mx = fmaxf(fmaxf(mx, fmaxf(a4[j][0], a4[j][1])), fmaxf(a4[j][2], a4[j][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int kt = wave + (THREADS / 64) * j;
            if (kt < FNT) {
                f32x4 pe;
#pragma unroll
                for (int i = 0; i < 4; ++i) pe[i] = __builtin_amdgcn_exp2f(a4[j][i] - mx);
                sum += (pe[0] + pe[1]) + (pe[2] + pe[3]);
                bf16x4 ps[3];
                split4v(pe, ps[0], ps[1], ps[2]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x4 vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        vf[pl] = tr_read(vfrag[pl][dt] + (kt == FNT - 1 ? last_off : (unsigned)(kt * 16 * KROWB)));
                    if (dt == 0) MFMA6_16(o0, vf, ps); else MFMA6_16(o1, vf, ps);
                }
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        __syncthreads();                                   // every wave has finished reading the K planes
        float* pw = Pw + wave * PWS;
        *reinterpret_cast<float4*>(pw + r * PRS + 4 * g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(pw + r * PRS + 16 + 4 * g) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        if (g == 0) {
            pw[16 * PRS + r] = mx;
            pw[16 * PRS + 16 + r] = sum;
        }
        __syncthreads();
        const int q = tid >> 5, d = tid & 31;
        int osrc = -1;
        if (q < FN - QT * 16) {
            int reg, cc;
            osrc = slot_info(QT * 16 + q, reg, cc);
        }
        if (osrc >= 0) {
            float M = -INFINITY;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) M = fmaxf(M, Pw[w8 * PWS + 16 * PRS + q]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) {
                const float f = __builtin_amdgcn_exp2f(Pw[w8 * PWS + 16 * PRS + q] - M);
                num += f * Pw[w8 * PWS + q * PRS + d];
                den += f * Pw[w8 * PWS + 16 * PRS + 16 + q];
            }
            out[(long)osrc * p.C + head * HD + d] = num / den;
        }
    }
    // The waves retire together: a wave that ended early would free its half of the SIMD's registers for another kernel's
    // wave while its neighbour is still issuing MFMAs -- the co-residence the register claim at the top rules out.
    __syncthreads();
    STAMP(stamp_slot);
    STAMP_RT(31);
}

// =============================================================================================
// K1, round 6: the STREAMING form of the split kernel (what split_arith = 1 launches now).
// Round 3's win_attn3d_split_kernel above keeps a 16-query tile's whole 16 x 400 score block in registers and does the softmax
// between two MFMA phases; its counters say the vector ALU (40 M instructions per stage-0 launch against 7 M MFMAs) and not
// the matrix pipe sets its time, and tools/microbench/mfma_valu_roles.hip says why: beside a stream of v_mfma_f32_16x16x32_bf16
// a SIMD issues NO vector instruction.  tools/microbench/mfma32_valu.hip (round 6): beside v_mfma_f32_32x32x16_bf16 it does --
// about four softmax-mix instructions per MFMA ride for free, two waves per SIMD -- if the two kinds are INTERLEAVED in the
// instruction stream.  So:
//   * MFMA shape 32x32x16: a wave owns a 32-query tile and walks the keys in chunks of 32.  S^T = K.Q^T of a chunk is
//     2 k-steps x 6 products on ONE 16-register accumulator tile (lane = query l & 31, half h = l >> 5 holds keys
//     8 j + 4 h + e of the chunk in register 4 j + e), O^T = V^T.P^T likewise 2 k-steps x 6 on one accumulator tile
//     (dims x queries); a lane's 16 scores are one query's: row sums stay in the lane (+ one xor-32 shuffle per tile).
//   * No score block: the chunk's scores are exponentiated, summed, split and fed to P.V while the next chunk's Q.K^T
//     MFMAs run -- the vector work sits BETWEEN MFMAs instead of in a phase of its own.  Software pipeline per chunk c:
//     phase 1 = [Q.K^T(c+1) k-step 0 | P.V(c-1) k-step 1] beside exp / sum / split of scores 0..7 of chunk c,
//     phase 2 = [Q.K^T(c+1) k-step 1 | P.V(c) k-step 0] beside scores 8..15.
//   * Softmax without a max: 2^score is taken as it comes (scale and log2 e are folded into Q and the bias column) and the
//     row SUM says afterwards whether that was legal (sum in [2^-60, 2^60] => the largest term is a normal number and
//     nothing overflowed); a tile that fails the test (scores beyond +-60 in log2 units: never with trained or the synthetic
//     weights, test_window_attention_large_scores) is redone by the two-pass form with the row max subtracted.
//   * The bias gather needs no per-group code fetch: with the chunk's column a compile-time constant the address is a
//     per-lane base minus a constant, i.e. the offset field of the ds_read.
//   * 392 queries = 12 tiles of 32 + 8: waves 0..3 take two tiles, waves 4..7 (their SIMD partners) one, and the 8-query
//     tile is shared between waves 4..7 by key chunks, merged through LDS (as the 25th tile was).
// K / V planes, their swizzle, the bias column and the staging code are the split kernel's: the K operand of a 32-key chunk
// (lane = key l & 31, 16-B piece 2 kk + h) and the transposing V reads (a 16-lane group = 4 keys x 16 dims) are conflict-free
// on the same image.
// =============================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SNC = 13;                          // key chunks of 32 slots: 12 full + one with 8
constexpr int SNQ = 13;                          // query tiles of 32: 12 full + one with 8
constexpr int SCHB = 32 * KROWB;                 // bytes of a chunk in a plane
constexpr int SQ_PRS = 36;                       // shared-tile scratch: row stride of a wave's O partial [8 queries][32 dims]
constexpr int SQ_PWS = 8 * SQ_PRS + 16;          // + row max [8] + row sum [8]
static_assert(4 * SQ_PWS * 4 <= 3 * PLANEB, "shared-tile scratch aliases the K planes");
__host__ __device__ constexpr int kc_of(int col) { return 60 * (13 * (col / 7) + col % 7); }   // bias-address term of a key column

#define MFMA6_BIG(acc, a, b)                                                                  \
    do {                                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);              \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);              \
    } while (0)

// The vector work of the streaming kernel is written instruction by instruction: left to itself hipcc packs the residual
// subtractions into v_pk_add_f32 (twice the issue cost of two v_sub_f32 beside MFMAs, MI355X_MICROARCH.md) and converts some
// elements one at a time.  A pair of f32 -> three packed bf16 pairs, exact (8 + 8 + 8 significand bits, round-to-nearest at
// each level): 3 converts + 4 unpacks + 4 subtractions.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// Compiler-visible (v_cvt_pk_bf16_f32): what reads a v_exp_f32 result must be an instruction hipcc knows, or its hazard
// recogniser does not keep the one wait state a VALU read needs behind a transcendental -- with the convert and the row-sum
// add as inline assembly, queries in lanes with bit 2 clear read stale exponentials (round 6, found with P == 1).
__device__ __forceinline__ unsigned cvt_pk_bf16(float x, float y) {
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// s += x where it is written (hipcc sinks a chain of plain adds to the end of the tile and spills the 196 terms until then).  `after`
// is a value computed FROM x by a compiler-visible instruction: the dependence keeps this add behind it, i.e. at least one
// instruction behind the v_exp_f32 that made x (see cvt_pk_bf16).
__device__ __forceinline__ void acc_f32(float& s, float x, unsigned after) {
    asm("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(x), "v"(after));
}
__device__ __forceinline__ void split_rest(float x, float y, unsigned c0, unsigned& c1, unsigned& c2) {
    float rx = sub_f32(x, __builtin_bit_cast(float, c0 << 16));
    float ry = sub_f32(y, __builtin_bit_cast(float, c0 & 0xffff0000u));
    c1 = cvt_pk_bf16(rx, ry);
    rx = sub_f32(rx, __builtin_bit_cast(float, c1 << 16));
    ry = sub_f32(ry, __builtin_bit_cast(float, c1 & 0xffff0000u));
    c2 = cvt_pk_bf16(rx, ry);
}
__device__ __forceinline__ void split_pair(float x, float y, unsigned& c0, unsigned& c1, unsigned& c2) {
    c0 = cvt_pk_bf16(x, y);
    split_rest(x, y, c0, c1, c2);
}
__device__ __forceinline__ void split8p(const float (&v)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
    u32x4 a, b, c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned c0, c1, c2;
        split_pair(v[2 * i], v[2 * i + 1], c0, c1, c2);
        a[i] = c0; b[i] = c1; c[i] = c2;
    }
    h0 = __builtin_bit_cast(bf16x8, a); h1 = __builtin_bit_cast(bf16x8, b); h2 = __builtin_bit_cast(bf16x8, c);
}

// Compile-time lists of key chunks (the unrolled tile walks one) and a compile-time loop.
template <int... C>
struct ChunkList {
    static constexpr int n = sizeof...(C);
    static constexpr int at(int i) {
        constexpr int a[] = {C...};
        return i >= 0 && i < n ? a[i] : -1;
    }
};
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS reads at (per-lane base register) + (compile-time byte offset): written as pointer arithmetic on an LDS pointer, so that the
// constant lands in the instruction's 16-bit offset field (plain unsigned address sums made hipcc add them in a VALU instruction).
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ bf16x4 tr_read_at(const unsigned base, const int off) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((lds_char*)(uintptr_t)base + off)));
}
__device__ __forceinline__ bf16x8 b128_read_at(const unsigned base, const int off) {
    return *(const __attribute__((address_space(3))) bf16x8*)((lds_char*)(uintptr_t)base + off);
}

#ifndef SOC_K1_PF_DIST
#define SOC_K1_PF_DIST 256                                    // workgroups ahead whose K / V rows are prefetched (same XCD, one round later)
#endif
template <bool SHIFTED>
__global__ __launch_bounds__(THREADS, 2) void win_attn3d_stream_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
    const float* __restrict__ table, float* __restrict__ out, const WinParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // owns its CUs like every bf16-MFMA kernel of the package (see win_attn3d_split_kernel)
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    float* Tb = reinterpret_cast<float*>(smem_raw + SP_TB);
    char* Kp = smem_raw + SP_K;
    int* wflag = reinterpret_cast<int*>(smem_raw + SP_MISC);
    int* gtab = wflag + 8;                                      // [2 col + half]: 4 * c | region << 16 (the region is what is read here)
    float* Pw = reinterpret_cast<float*>(Kp);                   // shared-tile scratch, aliases the K planes (behind a barrier)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    int qpart = 0, qsplit = 1;
    if (bid >= p.n_main) {
        const int rem = bid - p.n_main;
        qsplit = p.qsplit;
        qpart = rem % qsplit;
        bid = p.n_main + rem / qsplit;
    }
    const int head = bid % p.nH; bid /= p.nH;
    const int wx = bid % p.nww; bid /= p.nww;
    const int wy = bid % p.nwh; bid /= p.nwh;
    const int wz = bid % p.nwd; bid /= p.nwd;
    const int b = bid;
    const int C3 = 3 * p.C;
#ifdef SOC_K1_STAGGER                                         // diagnostic: the first round of workgroups starts spread over
    if (blockIdx.x < 256) {                                   // 8 steps of SOC_K1_STAGGER cycles (are the K / V bursts of a round in lockstep?)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * SOC_K1_STAGGER;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    STAMP_RT(30);
    STAMP_HWID(29);
    STAMP(0);

    // slot = col*8 + dz, col = dy*7 + dx (temporal index fastest)
    auto slot_info = [&](int i, int& reg, int& ccode) -> int {
        const int col = i >> 3, dz = i & 7;
        const int dy = (col * 37) >> 8, dx = col - dy * 7;     // col / 7 for col < 64
        const int zs = wz * 8 + dz, ys = wy * 7 + dy, xs = wx * 7 + dx;  // shifted frame
        int z = zs + p.sd; if (z >= p.Dp) z -= p.Dp;
        int y = ys + p.sh; if (y >= p.Hp) y -= p.Hp;
        int x = xs + p.sw; if (x >= p.Wp) x -= p.Wp;
        reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
        ccode = ((dy * 13 + dx) * 15 - dz + 8) * 4;
        return (z < p.D && y < p.H && x < p.W) ? ((b * p.D + z) * p.H + y) * p.W + x : -1;
    };

    const int n32 = lane & 31, hh = lane >> 5;
    // Q of a tile: lane (query n32, dims 8 hh .. + 7 and 16 + 8 hh ..), raw; + the query's token / bias code / region
    auto load_q = [&](int qt, float (&qf)[16], int& qsrc, int& qcode) {
        int reg, cc;
        const int slot = qt * 32 + n32;
        qsrc = -2; qcode = 0;
        if (slot < FN) {
            qsrc = slot_info(slot, reg, cc);
            qcode = cc | (reg << 16);
        }
        if (qsrc >= 0) {
            const float4* qrow = reinterpret_cast<const float4*>(qkv + (long)qsrc * C3 + head * HD + 8 * hh);
            const float4 a = qrow[0], c = qrow[1], d = qrow[4], e = qrow[5];
            qf[0] = a.x; qf[1] = a.y; qf[2] = a.z; qf[3] = a.w; qf[4] = c.x; qf[5] = c.y; qf[6] = c.z; qf[7] = c.w;
            qf[8] = d.x; qf[9] = d.y; qf[10] = d.z; qf[11] = d.w; qf[12] = e.x; qf[13] = e.y; qf[14] = e.z; qf[15] = e.w;
        } else if (qsrc == -1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                qf[kk] = qkv_bias[head * HD + 8 * hh + kk];
                qf[8 + kk] = qkv_bias[head * HD + 16 + 8 * hh + kk];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) qf[kk] = 0.f;
        }
    };
    // ---- tiles of this wave.  Unsplit workgroup: waves 0..3 own tiles w and w + 4, waves 4..7 tile w + 4 (8..11), tile 12
    //      (8 queries) is shared below.  A part of a split pair: tiles qpart + qsplit k, dealt k = wave, wave + 8, ...
    const bool share_last = qsplit == 1;
    int qt, qstep, qend;
    if (share_last) {
        qt = wave < 4 ? wave : wave + 4;
        qstep = 4;
        qend = wave < 4 ? 8 : SNQ - 1;
    } else {
        qt = qpart + qsplit * wave;
        qstep = (THREADS / 64) * qsplit;
        qend = SNQ;
    }
    // The first tile's Q (and, in waves 4..7 of an unsplit workgroup, the shared tile's) is requested BEFORE the K / V rows: its
    // latency then hides under the staging instead of in front of the first tile (the first tile of a wave took ~3 k cycles more
    // than its second, whose Q had been prefetched).
    const bool light = share_last && wave >= 4;
    float q_first[16], q_shared[16];
    int qsrc_first = -2, qcode_first = 0, qsrc_shared = -2, qcode_shared = 0;
    if (qt < qend) load_q(qt, q_first, qsrc_first, qcode_first);
    if (light) load_q(SNQ - 1, q_shared, qsrc_shared, qcode_shared);

    // ---- staging (the split kernel's): K and V rows (8 dims per thread) split into three bf16 planes each, bias column, group table
    {
        int differs = 0, reg0, c0;
        (void)slot_info(0, reg0, c0);
        constexpr int ITEMS = FN * 4;                       // (slot, 8-dim chunk)
        constexpr int PASSES = (ITEMS + THREADS - 1) / THREADS;
        float4 kv[PASSES][2], vv[PASSES][2];
        const int c = tid & 3;
        // The four lanes of a quad stage the same slot in every pass (8 dims each): lane c derives the token of the quad's slot of
        // pass c, once, and the quad reads it from that lane in pass c (one DPP move) -- every lane deriving all four was a third
        // of this phase's vector instructions.
        static_assert(PASSES == 4, "one pass per lane of a quad");
        int s_mine = -2;
        {
            const int i = (tid >> 2) + (THREADS / 4) * c;
            int reg = 0, cc = 0;
            if (i < FN) {
                s_mine = slot_info(i, reg, cc);
                differs |= (reg != reg0);
            }
        }
        const int s_pass[PASSES] = {__builtin_amdgcn_update_dpp(0, s_mine, 0x00, 0xf, 0xf, false),      // quad_perm: lane 0 of the quad
                                    __builtin_amdgcn_update_dpp(0, s_mine, 0x55, 0xf, 0xf, false),
                                    __builtin_amdgcn_update_dpp(0, s_mine, 0xaa, 0xf, 0xf, false),
                                    __builtin_amdgcn_update_dpp(0, s_mine, 0xff, 0xf, 0xf, false)};
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int s = s_pass[it];
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 16)         // diagnostic: no K / V row loads
            if (s >= 0) {
                kv[it][0] = kv[it][1] = vv[it][0] = vv[it][1] = make_float4(0.01f * c, 0.02f, 0.03f * s, 0.04f);
            } else {
#else
            if (s >= 0) {
                const float4* row = reinterpret_cast<const float4*>(qkv + (long)s * C3 + head * HD + 8 * c);
                kv[it][0] = row[p.C / 4]; kv[it][1] = row[p.C / 4 + 1];
                vv[it][0] = row[p.C / 2]; vv[it][1] = row[p.C / 2 + 1];
            } else {
#endif
                const float4* kb = reinterpret_cast<const float4*>(qkv_bias + p.C + head * HD + 8 * c);
                const float4* vb = reinterpret_cast<const float4*>(qkv_bias + 2 * p.C + head * HD + 8 * c);
                kv[it][0] = kb[0]; kv[it][1] = kb[1];
                vv[it][0] = vb[0]; vv[it][1] = vb[1];
            }
        }
        constexpr int TPASS = (TBL + THREADS - 1) / THREADS;
        float tv[TPASS];
#pragma unroll
        // The bias column of this head, walked in the TABLE's order (entry r = zz * 169 + yx at table[r * nH + head]: the 64 lanes of
        // a load touch 64 * nH floats, 6 cache lines at nH = 3, instead of one line each) and transposed by the LDS store below
        // (stride 15 words: no bank conflicts).  Measured against the split kernel's walk in the LDS layout's order, same box:
        // 1 151 -> 1 148 / 1 202 -> 1 198 us at ten clips per launch -- the gather's cost is its five load instructions in the
        // queue behind the K / V rows (a build without it: -5 %), not the lines it touches.
        for (int it = 0; it < TPASS; ++it) {
            const int r = it * THREADS + tid;
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 8)          // diagnostic: no bias-table gather
            tv[it] = 0.f;
#else
            tv[it] = r < TBL ? table[(long)r * p.nH + head] : 0.f;
#endif
        }
        const int wave_differs = __any(differs);
        if (SHIFTED && lane == 0) wflag[wave] = wave_differs ? 1 : 0;
        STAMP(1);
        if (tid < FNT * 4) {                                // key-group codes: group = slots 8 col + 4 half .. +3, index 2 col + half
            const int t = tid >> 2, gg = tid & 3;
            const int col = 2 * t + (gg >> 1);
            const int dy = (col * 37) >> 8, dx = col - dy * 7;
            int reg = 0;
            if (SHIFTED) {
                const int zs = wz * 8 + 4 * (gg & 1), ys = wy * 7 + dy, xs = wx * 7 + dx;
                reg = (region1d(zs, p.Dp, 8, p.sd) * 3 + region1d(ys, p.Hp, 7, p.sh)) * 3 + region1d(xs, p.Wp, 7, p.sw);
            }
            gtab[tid] = (((dy * 13 + dx) * 15 - 4 * (gg & 1) + 8) * 4) | (reg << 16);
        }
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int i = (tid + THREADS * it) >> 2;
            if (i < FN) {
                const float kf[8] = {kv[it][0].x, kv[it][0].y, kv[it][0].z, kv[it][0].w, kv[it][1].x, kv[it][1].y, kv[it][1].z, kv[it][1].w};
                const float vf[8] = {vv[it][0].x, vv[it][0].y, vv[it][0].z, vv[it][0].w, vv[it][1].x, vv[it][1].y, vv[it][1].z, vv[it][1].w};
                bf16x8 h0, h1, h2;
                char* dst = Kp + i * KROWB + ((c ^ swz(i)) << 4);
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 32)         // diagnostic: no operand split, one LDS store per item
                *reinterpret_cast<float4*>(dst) = make_float4(kf[0] + vf[1], kf[2], vf[3], kf[4]);
                continue;
#endif
                split8v(kf, h0, h1, h2);
                *reinterpret_cast<bf16x8*>(dst) = h0;
                *reinterpret_cast<bf16x8*>(dst + PLANEB) = h1;
                *reinterpret_cast<bf16x8*>(dst + 2 * PLANEB) = h2;
                split8v(vf, h0, h1, h2);
                *reinterpret_cast<bf16x8*>(dst + 3 * PLANEB) = h0;
                *reinterpret_cast<bf16x8*>(dst + 4 * PLANEB) = h1;
                *reinterpret_cast<bf16x8*>(dst + 5 * PLANEB) = h2;
            }
        }
#pragma unroll
        for (int it = 0; it < TPASS; ++it) {
            const int r = it * THREADS + tid;
            const int zz = r / 169, yx = r - zz * 169;
            if (r < TBL) Tb[yx * 15 + 14 - zz] = tv[it] * LOG2E;
        }
    }
    STAMP(2);
    __syncthreads();
    bool has_mask = false;
    if (SHIFTED) {
        int f = 0;
#pragma unroll
        for (int w8 = 0; w8 < THREADS / 64; ++w8) f |= wflag[w8];
        has_mask = f != 0;
    }
    STAMP(3);
    int stamp_slot = 4;
    (void)stamp_slot;

#ifdef SOC_K1_PREFETCH                                          // diagnostic, measured a LOSS (1 148 -> 1 186 us at ten clips, stage 0)
    // ---- the K / V rows of the workgroup that follows this one on this XCD (workgroup b runs on XCD b % 8: b + 256 is the one a
    //      round later) are pulled into the XCD's L2 while this workgroup computes: staging is bound by the latency of its 784 row
    //      reads (12.5 k cycles of a 68 k-cycle workgroup at launch-group sizes, where qkv no longer sits in the MALL), not by any
    //      bandwidth.  One dword per 128-B row through the LDS-DMA path (no destination register to keep alive), into a dummy
    //      LDS word per lane that nothing reads.
    {
        __shared__ int pf_sink[64];
        int nb = (int)blockIdx.x + SOC_K1_PF_DIST;
        if (nb < (int)gridDim.x && tid < FN) {
            if (nb >= p.n_main) nb = p.n_main + (nb - p.n_main) / p.qsplit;
            const int head2 = nb % p.nH; nb /= p.nH;
            const int wx2 = nb % p.nww; nb /= p.nww;
            const int wy2 = nb % p.nwh; nb /= p.nwh;
            const int wz2 = nb % p.nwd; nb /= p.nwd;
            const int col = tid >> 3, dz = tid & 7;
            const int dy = (col * 37) >> 8, dx = col - dy * 7;
            int z = wz2 * 8 + dz + p.sd; if (z >= p.Dp) z -= p.Dp;
            int y = wy2 * 7 + dy + p.sh; if (y >= p.Hp) y -= p.Hp;
            int x = wx2 * 7 + dx + p.sw; if (x >= p.Wp) x -= p.Wp;
            if (z < p.D && y < p.H && x < p.W) {
                const float* row = qkv + (long)(((nb * p.D + z) * p.H + y) * p.W + x) * C3 + head2 * HD;
                // inline asm, not __builtin_amdgcn_global_load_lds: hipcc would put s_waitcnt vmcnt(0) in front of the first
                // ds_read_b64_tr_b16 that follows (an LDS access it cannot tell from the sink) -- a stall of one HBM latency per
                // workgroup.  Untracked, the two loads only make later vmcnt waits conservative (returns are in order).
                const unsigned sink = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)pf_sink;
                unsigned m0_saved;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                             "global_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_saved) : "v"(row + p.C), "v"(row + 2 * p.C), "s"(sink));
            }
        }
    }
#endif

    const unsigned tbase = (unsigned)(uintptr_t)(lds_cfloat*)Tb;
    const unsigned kaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)Kp;
    const unsigned vaddr = kaddr + 3 * PLANEB;
    const unsigned gaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)gtab + 4u * (unsigned)hh;
    // The scale lives in a vector register (no SGPR operand in packed f32 code beside bf16 MFMA + LDS waves).
    float scale;
    asm volatile("v_mov_b32 %0, %1" : "=v"(scale) : "v"(0.17677669529663687f * LOG2E));
    const int C0 = ((6 * 13 + 6) * 15 + 7) * 4;
    // K operand of chunk c, plane pl, k-step kk: lane (key n32, dims 16 kk + 8 hh ..); + SCHB c.  Two base registers per k-step:
    // planes 0 / 1 ride on the instruction's 16-bit offset, plane 2 would not fit beside the chunk offset.
    unsigned kb01[2], kb2[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        kb01[kk] = kaddr + (unsigned)(n32 * KROWB + (((2 * kk + hh) ^ swz(n32)) << 4));
        kb2[kk] = kb01[kk] + 2u * PLANEB;
    }
    // V operand of chunk c, k-step s, plane pl, key quartet jj: a 16-lane group (dims 16 dhalf ..) reads keys 16 s + 8 jj + 4 hh + vq;
    // + SCHB c + 1024 s
    unsigned vb01[2], vb2[2];
    {
        const int li = lane & 15, vq = li >> 2, pp = li & 3, dhalf = (lane >> 4) & 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = 8 * jj + 4 * hh + vq;
            vb01[jj] = vaddr + (unsigned)(row * KROWB + (((2 * dhalf + (pp >> 1)) ^ swz(row)) << 4) + 8 * (pp & 1));
            vb2[jj] = vb01[jj] + 2u * PLANEB;
        }
    }
    auto k_read = [&](int pl, int kk, int off) -> bf16x8 {            // K fragment of plane pl, k-step kk at byte offset off
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 2)          // diagnostic: no LDS reads of K / V fragments
        return __builtin_bit_cast(bf16x8, (u32x4){kb01[kk] + (unsigned)off, kb2[kk], 0x3f803f80u, (unsigned)pl});
#endif
        return pl == 2 ? b128_read_at(kb2[kk], off) : b128_read_at(kb01[kk], pl * PLANEB + off);
    };
    auto v_read = [&](int pl, int jj, int off) -> bf16x4 {            // V key quartet jj of plane pl at byte offset off
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 2)
        return __builtin_bit_cast(bf16x4, (f32x2){__builtin_bit_cast(float, vb01[jj] + (unsigned)off), __builtin_bit_cast(float, 0x3f803f80u + (unsigned)pl)});
#endif
        return pl == 2 ? tr_read_at(vb2[jj], off) : tr_read_at(vb01[jj], pl * PLANEB + off);
    };

    auto split_q = [&](const float (&qf)[16], bf16x8 (&qs)[2][3]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float t8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t8[i] = qf[8 * kk + i] * scale;
            split8p(t8, qs[kk][0], qs[kk][1], qs[kk][2]);
        }
    };
    auto k_frag = [&](int pl, int kk, unsigned off) -> bf16x8 { return k_read(pl, kk, (int)off); };
    const float PEN = -100.0f * LOG2E;

    // ------------------------------------------------------------------------------------------------------------------
    // The pipelined tile: all 13 key chunks, no max subtraction.  Returns the lane's part of the row sum; O^T in `O`.
    // ------------------------------------------------------------------------------------------------------------------
    auto fast_tile = [&](auto masked_tag, auto list_tag, const bf16x8 (&qs)[2][3], const unsigned qb, const int qreg, f32x16& O) -> float {
        constexpr bool MASKED = decltype(masked_tag)::value;
        using L = decltype(list_tag);             // the key chunks this call walks: all 13 (a whole tile) or a wave's share of the shared tile
        // The instruction stream is laid out by hand, one MFMA per SLOT: [1 MFMA | <= 1-2 LDS reads | 5 vector instructions] and a
        // scheduling barrier behind every slot.  Left to the scheduler the unrolled tile came out as runs of 6-7 back-to-back MFMAs
        // between runs of ~35 vector instructions, and neither run overlaps with anything (tools/microbench/mfma32_valu.hip: beside
        // v_mfma_f32_32x32x16_bf16 about four vector instructions per MFMA ride for free only when they sit BETWEEN the MFMAs).
        // A phase = 12 slots = 6 P.V MFMAs (their V fragments and P planes were made a phase earlier) + 6 Q.K^T MFMAs of the next
        // chunk, beside the exp / row sum / three-way split of 8 scores (4 pairs x 3 segments of 5 instructions); every LDS read
        // is issued a phase ahead of its MFMA.
        f32x16 S[2];
        u32x4 Pa[3], Pb[3];                       // P planes of a k-step: packed bf16 pairs
        bf16x8 kf[3], kfn[3];                     // K fragments: this phase's Q.K^T k-step, the next phase's
        bf16x4 vlo[3], vhi[3], vlon[3], vhin[3];  // V fragments (two key quartets per plane): this phase's P.V k-step, the next phase's
        float sum = 0.f, sum2 = 0.f;
        float px = 0.f, py = 0.f, prx = 0.f, pry = 0.f;
        unsigned pc0 = 0u, pc1 = 0u;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};       // the six products, smallest first
        // The 15 vector instructions of a pair of scores, cut into three segments of five so that every instruction that needs a
        // wait state behind its producer (the convert behind v_exp_f32; the converts behind the inline-assembly subtractions,
        // which hipcc guards with an s_nop because it cannot see into them) finds the slot's MFMA in between instead of an s_nop:
        //   seg 0: [third plane + row sums of the PREVIOUS pair] exp x, exp y
        //   seg 1: first plane (convert), unpack, two subtractions
        //   seg 2: second plane (convert), unpack, two subtractions
        // The previous pair of a phase's first pair is the last pair of the phase before (its third plane is first read by the third
        // P.V MFMA of this phase).
        auto finish = [&](u32x4 (&P)[3], int i) {
            const unsigned c2 = cvt_pk_bf16(prx, pry);
            P[2][i] = c2;
            acc_f32(sum, px, c2);
            acc_f32(sum2, py, c2);
        };
        auto segA = [&](const f32x16& s, int e) {
            px = __builtin_amdgcn_exp2f(s[e]);
            py = __builtin_amdgcn_exp2f(s[e + 1]);
        };
        auto segB = [&](u32x4 (&P)[3], int i) {
            pc0 = cvt_pk_bf16(px, py);
            P[0][i] = pc0;
            prx = sub_f32(px, __builtin_bit_cast(float, pc0 << 16));
            pry = sub_f32(py, __builtin_bit_cast(float, pc0 & 0xffff0000u));
        };
        auto segC = [&](u32x4 (&P)[3], int i) {
            pc1 = cvt_pk_bf16(prx, pry);
            P[1][i] = pc1;
            prx = sub_f32(prx, __builtin_bit_cast(float, pc1 << 16));
            pry = sub_f32(pry, __builtin_bit_cast(float, pc1 & 0xffff0000u));
        };
        // slot k of a phase: segment k % 3 of pair k / 3 of the 8 scores s[8 half ..]; pairs >= npairs are keys that do not exist
        // (P = 0); nprev = pairs of the phase before (0: none), whose last one is finished in slot 0
        auto soft_slot = [&](const f32x16& s, int half, int k, int npairs, u32x4 (&P)[3], u32x4 (&Pprev)[3], int nprev) {
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 1)          // diagnostic (tools/k1_probe.py --flags): no vector work in the tile
            if (k == 0) { P[0] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; P[1] = P[0]; P[2] = P[0]; sum += s[half]; }
            return;
#endif
            const int i = k / 3, seg = k % 3;
            if (seg == 0) {
                if (i == 0 && nprev > 0) finish(Pprev, nprev - 1);
                if (i >= 1 && i - 1 < npairs) finish(P, i - 1);
                if (i < npairs) segA(s, 8 * half + 2 * i);
                else { P[0][i] = 0u; P[1][i] = 0u; P[2][i] = 0u; }
            } else if (i < npairs) {
                if (seg == 1) segB(P, i);
                else segC(P, i);
            }
        };
        auto mf_pv = [&](int k, const bf16x4 (&lo)[3], const bf16x4 (&hi)[3], const u32x4 (&P)[3]) {
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 4)          // diagnostic: no MFMAs (the operands are still consumed)
            if (k == 5) O[0] += (float)lo[0][0] + (float)hi[1][0] + (float)lo[2][1] + __builtin_bit_cast(float, P[0][0] ^ P[1][1] ^ P[2][2]);
            return;
#endif
            const bf16x4 l = lo[PA[k]], h = hi[PA[k]];
            const bf16x8 vf = {l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
            O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, P[PB[k]]), O, 0, 0, 0);
        };
        auto mf_qk = [&](int k, f32x16& s, const bf16x8 (&kfr)[3], int kk) {
#if defined(SOC_K1_VAR) && (SOC_K1_VAR & 4)
            if (k == 5) s[kk] += (float)kfr[0][0] + (float)kfr[1][1] + (float)kfr[2][2];
            return;
#endif
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[PA[k]], qs[kk][PB[k]], s, 0, 0, 0);
        };
        // bias of key column j of chunk c into scores 4 j .. 4 j + 3 (+ the shift-mask penalty)
        auto init_j = [&](auto c_tag, int j, f32x16& s) {
            constexpr int c = decltype(c_tag)::value;
            constexpr int NJ = c == SNC - 1 ? 1 : 4;
            constexpr int KCMAX = kc_of(4 * c + NJ - 1);
            lds_cfloat* bp = (lds_cfloat*)(uintptr_t)(qb - (unsigned)KCMAX);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * j + e] = j < NJ ? bp[(KCMAX - kc_of(4 * c + (j < NJ ? j : 0))) / 4 + e] : 0.f;
        };
        auto pen_j = [&](auto c_tag, int j, f32x16& s) {
            constexpr int c = decltype(c_tag)::value;
            constexpr int NJ = c == SNC - 1 ? 1 : 4;
            if (MASKED && j < NJ) {
                const __attribute__((address_space(3))) int* gp = (const __attribute__((address_space(3))) int*)(uintptr_t)gaddr;
                const float pen = ((gp[8 * c + 2 * j] >> 16) != qreg) ? PEN : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * j + e] += pen;
            }
        };
#define SOC_SLOT_END __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = 0.f;
        // ---- prologue: scores of the first chunk (nothing to run beside them), then what the first step expects to find:
        //      K(second chunk, k-step 0) and the second chunk's bias
        {
            using C0T = std::integral_constant<int, L::at(0)>;
            using C1T = std::integral_constant<int, (L::n > 1) ? L::at(1) : L::at(0)>;
#pragma unroll
            for (int j = 0; j < 4; ++j) init_j(C0T{}, j, S[0]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { kf[pl] = k_read(pl, 0, C0T::value * SCHB); kfn[pl] = k_read(pl, 1, C0T::value * SCHB); }
#pragma unroll
            for (int j = 0; j < 4; ++j) pen_j(C0T{}, j, S[0]);
#pragma unroll
            for (int k = 0; k < 6; ++k) mf_qk(k, S[0], kf, 0);
            if constexpr (L::n > 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) init_j(C1T{}, j, S[1]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = k_read(pl, 0, C1T::value * SCHB);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) mf_qk(k, S[0], kfn, 1);
            SOC_SLOT_END;
        }
        static_for<L::n>([&](auto i_tag) {
            constexpr int I = decltype(i_tag)::value;
            constexpr int c = L::at(I), cn = L::at(I + 1), cnn = L::at(I + 2);      // this chunk, the next two (-1: none)
            constexpr bool FIRST = I == 0, PARTIAL = c == SNC - 1, NEXT = cn >= 0, NEXT2 = cnn >= 0;
            using CNT = std::integral_constant<int, NEXT ? cn : c>;
            using CNNT = std::integral_constant<int, NEXT2 ? cnn : c>;
            f32x16& cur = S[I & 1];
            f32x16& nxt = S[(I + 1) & 1];
            // ---- phase 1: P.V(previous chunk, k-step 1) | Q.K^T(next chunk, k-step 0); scores 0..7 of this chunk -> Pa; reads for phase 2
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (k < 6) {
                    if constexpr (!FIRST) mf_pv(k, vlo, vhi, Pb);
                    // V(c, k-step 0); chunk 12 has no second key quartet (P is 0 there): its operand re-reads the first
                    if (k % 2 == 0) vlon[k / 2] = v_read(k / 2, 0, c * SCHB);
                    else vhin[k / 2] = v_read(k / 2, PARTIAL ? 0 : 1, c * SCHB);
                    if constexpr (NEXT) {
                        if (k < 4) pen_j(CNT{}, k, nxt);
                    }
                } else {
                    if constexpr (NEXT) {
                        mf_qk(k - 6, nxt, kf, 0);
                        if (k < 9) kfn[k - 6] = k_read(k - 6, 1, cn * SCHB);
                    }
                }
                soft_slot(cur, 0, k, PARTIAL ? 2 : 4, Pa, Pb, FIRST ? 0 : 4);
                SOC_SLOT_END;
            }
            // ---- phase 2: P.V(c, k-step 0) | Q.K^T(next chunk, k-step 1); scores 8..15 -> Pb; reads for the next phase 1
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (k < 6) {
                    mf_pv(k, vlon, vhin, Pa);
                    if constexpr (!PARTIAL) {
                        if (k % 2 == 0) vlo[k / 2] = v_read(k / 2, 0, c * SCHB + 1024);
                        else vhi[k / 2] = v_read(k / 2, 1, c * SCHB + 1024);
                    }
                } else {
                    if constexpr (NEXT) mf_qk(k - 6, nxt, kfn, 1);
                    if constexpr (NEXT2) {
                        if (k < 9) kf[k - 6] = k_read(k - 6, 0, cnn * SCHB);
                        // bias of the chunk after next into the registers this chunk's scores leave: 0..11 are free by now, 12..15
                        // after slot 9
                        if (k == 6) init_j(CNNT{}, 0, cur);
                        if (k == 7) init_j(CNNT{}, 1, cur);
                        if (k == 8) init_j(CNNT{}, 2, cur);
                    }
                }
                if constexpr (!PARTIAL) soft_slot(cur, 1, k, 4, Pb, Pa, 4);      // (chunk 12: both of its pairs were finished in phase 1)
                if constexpr (NEXT2) {
                    if (k == 11) init_j(CNNT{}, 3, cur);
                }
                SOC_SLOT_END;
            }
            // ---- the list ends on a whole chunk: its second k-step is still due
            if constexpr (!NEXT && !PARTIAL) {
                finish(Pb, 3);
                SOC_SLOT_END;
#pragma unroll
                for (int k = 0; k < 6; ++k) mf_pv(k, vlo, vhi, Pb);
                SOC_SLOT_END;
            }
        });
#undef SOC_SLOT_END
        return sum + sum2;
    };

    // ------------------------------------------------------------------------------------------------------------------
    // The generic chunk forms (runtime chunk index): the two-pass tile for scores outside the safe range, the shared tile.
    // ------------------------------------------------------------------------------------------------------------------
    auto chunk_scores = [&](int c, const bf16x8 (&qs)[2][3], unsigned qb, int qreg) -> f32x16 {
        const int nj = c == SNC - 1 ? 1 : 4;
        f32x16 s;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 4 * c + (j < nj ? j : 0);
            const int dy = (col * 37) >> 8, dx = col - dy * 7;
            lds_cfloat* bp = (lds_cfloat*)(uintptr_t)(qb - (unsigned)(60 * (13 * dy + dx)));
            float pen = 0.f;
            if (SHIFTED && has_mask) {
                const int gc = gtab[2 * col + hh];
                pen = ((gc >> 16) != qreg) ? PEN : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * j + e] = j < nj ? bp[e] + pen : -INFINITY;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 kf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) kf[pl] = k_frag(pl, kk, (unsigned)(c * SCHB));
            MFMA6_BIG(s, kf, qs[kk]);
        }
        if (nj < 4) {                                          // keys that do not exist
#pragma unroll
            for (int i = 4; i < 16; ++i) s[i] = -INFINITY;
        }
        return s;
    };
    // P = 2^(s - m), row sum, O^T += V^T . P^T for one chunk
    auto chunk_apply = [&](int c, const f32x16& s, float m, float& sum, f32x16& O) {
        const bool last = c == SNC - 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && last) break;
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = __builtin_amdgcn_exp2f(s[8 * half + i] - m);      // -inf -> 0
            sum += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
            bf16x8 ps[3], vf[3];
            split8p(pv, ps[0], ps[1], ps[2]);
            const unsigned off = (unsigned)(c * SCHB + half * 1024);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const bf16x4 lo = v_read(pl, 0, (int)off);
                const bf16x4 hi = last ? v_read(pl, 0, (int)off) : v_read(pl, 1, (int)off);
                vf[pl] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            MFMA6_BIG(O, vf, ps);
        }
    };
    auto row_max = [&](const f32x16& s, float mx) -> float {
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[i]), s[i + 1]);      // compiler-visible (v_max3_f32): reads MFMA results
        return mx;
    };
    auto slow_tile = [&](const bf16x8 (&qs)[2][3], unsigned qb, int qreg, f32x16& O) -> float {
        float mx = -INFINITY;
#pragma unroll 1
        for (int c = 0; c < SNC; ++c) mx = row_max(chunk_scores(c, qs, qb, qreg), mx);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = 0.f;
#pragma unroll 1
        for (int c = 0; c < SNC; ++c) {
            const f32x16 s = chunk_scores(c, qs, qb, qreg);
            chunk_apply(c, s, mx, sum, O);
        }
        return sum;
    };
    auto store_tile = [&](int qsrc, const f32x16& O, float tot) {
        if (qsrc >= 0) {
            const float inv = 1.f / tot;
            float* orow = out + (long)qsrc * p.C + head * HD + 4 * hh;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<float4*>(orow + 8 * k) =
                    make_float4(O[4 * k] * inv, O[4 * k + 1] * inv, O[4 * k + 2] * inv, O[4 * k + 3] * inv);
        }
    };

    // `carry`: the NEXT tile's raw Q in the waves that own two tiles (waves 0..3 of an unsplit workgroup, any wave of a part), and in
    // waves 4..7 of an unsplit workgroup -- which own one tile and never prefetch -- their partial O of the shared tile: one set of
    // 16 registers for both.
    float carry[16];
    float part_sum = 0.f, part_mx = 0.f;
    float* xsum = reinterpret_cast<float*>(gtab + FNT * 4);          // [4 waves][8 queries], in the slack behind the group table
    auto shared_partial = [&](const bool use_max) __attribute__((always_inline)) {
        // tile 12 (queries 384..391 in lanes n32 < 8): wave w takes key chunks w - 4, w, w + 4 (, 12)
        if (use_max) load_q(SNQ - 1, q_shared, qsrc_shared, qcode_shared);        // (the redo: the registers were given up)
        bf16x8 qs[2][3];
        split_q(q_shared, qs);
        const unsigned sh_qb = tbase + (unsigned)((qcode_shared & 0xFFFF) + C0 - 32 + 16 * hh);
        const int sh_qreg = qcode_shared >> 16;
        f32x16 O;
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] = 0.f;
        float mx = 0.f, sum = 0.f;
        if (use_max) {
            mx = -INFINITY;
#pragma unroll 1
            for (int c = wave - 4; c < SNC; c += 4) mx = row_max(chunk_scores(c, qs, sh_qb, sh_qreg), mx);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll 1
            for (int c = wave - 4; c < SNC; c += 4) {
                const f32x16 sc = chunk_scores(c, qs, sh_qb, sh_qreg);
                chunk_apply(c, sc, mx, sum, O);
            }
        } else {
            // the pipelined form over this wave's chunks (the chunk-at-a-time loop took 6 k cycles per chunk beside a streaming wave)
            auto run = [&](auto list_tag) {
                sum = (SHIFTED && has_mask) ? fast_tile(std::true_type{}, list_tag, qs, sh_qb, sh_qreg, O)
                                            : fast_tile(std::false_type{}, list_tag, qs, sh_qb, sh_qreg, O);
            };
            if (wave == 4) run(ChunkList<0, 4, 8, 12>{});
            else if (wave == 5) run(ChunkList<1, 5, 9>{});
            else if (wave == 6) run(ChunkList<2, 6, 10>{});
            else run(ChunkList<3, 7, 11>{});
        }
        part_sum = sum + __shfl_xor(sum, 32);
        if (use_max) part_mx = mx;                          // (0 otherwise: not a value that lives across the tile loop)
#pragma unroll
        for (int i = 0; i < 16; ++i) carry[i] = O[i];
    };
    if (light) {
        // FIRST, not last: these chunk-at-a-time passes stall on every LDS read and MFMA chain, which costs nothing while the SIMD's
        // other wave streams its whole tiles -- at the end of the workgroup they were 12-17 k cycles with the other wave idle
        shared_partial(false);
        if (lane < 8) xsum[(wave - 4) * 8 + lane] = part_sum;
    }
    STAMP(stamp_slot); ++stamp_slot;
    {
        int qsrc_n = qsrc_first, qcode_n = qcode_first;
        bf16x8 qs[2][3];
        if (qt < qend) split_q(q_first, qs);
        for (; qt < qend; qt += qstep) {
            const int qsrc = qsrc_n;
            const unsigned qb = tbase + (unsigned)((qcode_n & 0xFFFF) + C0 - 32 + 16 * hh);
            const int qreg = qcode_n >> 16;
            const bool more = qt + qstep < qend;                             // never in waves 4..7 of an unsplit workgroup
            if (more) load_q(qt + qstep, carry, qsrc_n, qcode_n);            // next tile's Q: in flight behind this tile
            f32x16 O;
            using All = ChunkList<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12>;
            float sum = (SHIFTED && has_mask) ? fast_tile(std::true_type{}, All{}, qs, qb, qreg, O)
                                              : fast_tile(std::false_type{}, All{}, qs, qb, qreg, O);
            float tot = sum + __shfl_xor(sum, 32);
            // 2^score without a max is legal iff the row sum says so (NaN fails the test too); lanes of queries that do not exist
            // (the 24 surplus lanes of tile 12 in a split pair: Q = 0, and with a shift mask possibly every key masked) do not vote
#if (defined(SOC_K1_DBG) && (SOC_K1_DBG == 1 || SOC_K1_DBG >= 3)) || defined(SOC_K1_VAR)
            if (false) {
#elif defined(SOC_K1_DBG) && SOC_K1_DBG == 2
            if (true) {
#else
            if (!__all(qsrc == -2 || (tot > 0x1p-60f && tot < 0x1p60f))) {
#endif
                sum = slow_tile(qs, qb, qreg, O);
                tot = sum + __shfl_xor(sum, 32);
            }
#if defined(SOC_K1_DBG) && SOC_K1_DBG == 3       // raw O, no normalisation
            store_tile(qsrc, O, 1.0f);
#elif defined(SOC_K1_DBG) && SOC_K1_DBG == 4     // the row sum in every dim
            { f32x16 T; for (int i = 0; i < 16; ++i) T[i] = tot; store_tile(qsrc, T, 1.0f); }
#else
            store_tile(qsrc, O, tot);
#endif
            if (more) split_q(carry, qs);
            STAMP(stamp_slot); ++stamp_slot;
        }
    }

    if (share_last) {
        // The shared tile's partial (O, sum) of waves 4..7 are merged through LDS (the scratch aliases the K planes: every wave must
        // be done with them first).  Like the whole tiles it was done WITHOUT a max; the four partial row sums sit in a few words
        // beside the group table, and only if their total says that was not legal the four waves redo their chunks with the max of
        // THEIR chunks subtracted (the K / V planes are still intact then) and the merge rescales the parts.
        __syncthreads();                                   // every wave has finished its whole tiles; the partial sums are there
        bool legal = true;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float tq = (xsum[q] + xsum[8 + q]) + (xsum[16 + q] + xsum[24 + q]);
            legal = legal && tq > 0x1p-60f && tq < 0x1p60f;             // (NaN fails too)
        }
#ifdef SOC_K1_VAR
        legal = true;
#endif
        if (!legal) {                                      // block-uniform: every thread read the same 32 words
            if (light) shared_partial(true);
            __syncthreads();                               // ... and now every wave is done with the K planes
        }
        if (light && n32 < 8) {
            float* pw = Pw + (wave - 4) * SQ_PWS;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<float4*>(pw + n32 * SQ_PRS + 8 * k + 4 * hh) = make_float4(carry[4 * k], carry[4 * k + 1], carry[4 * k + 2], carry[4 * k + 3]);
            if (hh == 0) {
                pw[8 * SQ_PRS + n32] = part_mx;
                pw[8 * SQ_PRS + 8 + n32] = part_sum;
            }
        }
        __syncthreads();
        if (tid < 8 * 32) {
            const int q = tid >> 5, d = tid & 31;
            int reg, cc;
            const int osrc = slot_info((SNQ - 1) * 32 + q, reg, cc);
            if (osrc >= 0) {
                float M = -INFINITY;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) M = fmaxf(M, Pw[w4 * SQ_PWS + 8 * SQ_PRS + q]);
                float num = 0.f, den = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const float f = __builtin_amdgcn_exp2f(Pw[w4 * SQ_PWS + 8 * SQ_PRS + q] - M);       // 1 when no max was taken
                    num += f * Pw[w4 * SQ_PWS + q * SQ_PRS + d];
                    den += f * Pw[w4 * SQ_PWS + 8 * SQ_PRS + 8 + q];
                }
                out[(long)osrc * p.C + head * HD + d] = num / den;
            }
        }
    }
    // The waves retire together (the co-residence rule of the bf16-MFMA kernels).
    __syncthreads();
    STAMP(stamp_slot);
    STAMP_RT(31);
}

int launch_full(const float* qkv, const float* qkv_bias, const float* table, float* out,
                const WinParams& p, long blocks, hipStream_t st) {
    const size_t lds = FULL_LDS_BYTES;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];   // per device: the attribute is device state
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_full_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_full_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);   // idempotent: a racing thread just sets it again
    }
    if (p.shifted)
        hipLaunchKernelGGL((win_attn3d_full_kernel<true>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    else
        hipLaunchKernelGGL((win_attn3d_full_kernel<false>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    return soc_check_launch();
}

int launch_split(const float* qkv, const float* qkv_bias, const float* table, float* out,
                 const WinParams& p, long blocks, hipStream_t st) {
    const size_t lds = SPLIT_LDS_BYTES;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_split_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_split_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (p.shifted)
        hipLaunchKernelGGL((win_attn3d_split_kernel<true>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    else
        hipLaunchKernelGGL((win_attn3d_split_kernel<false>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    return soc_check_launch();
}

int launch_stream(const float* qkv, const float* qkv_bias, const float* table, float* out,
                  const WinParams& p, long blocks, hipStream_t st) {
    const size_t lds = SPLIT_LDS_BYTES;                        // + 256 B static (the prefetch sink)
    static_assert(SPLIT_LDS_BYTES + 256 <= 160 * 1024, "LDS");
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_stream_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_stream_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess)
            return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (p.shifted)
        hipLaunchKernelGGL((win_attn3d_stream_kernel<true>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    else
        hipLaunchKernelGGL((win_attn3d_stream_kernel<false>), dim3((unsigned)blocks), dim3(THREADS), lds, st, qkv,
                           qkv_bias, table, out, p);
    return soc_check_launch();
}

template <int NT, int NT_PREV>
int launch_nt(const float* qkv, const float* qkv_bias, const float* table, float* out,
              const WinParams& p, long blocks, hipStream_t st) {
    const size_t lds = (size_t)(2 * NT * 16 * RS + ((p.table_len + 3) & ~3)) * sizeof(float) +
                       (3 * NT * 16 + 8) * sizeof(int);
    if (lds > 160 * 1024) return SOC_EUNSUPPORTED;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];   // per instantiation and per device
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_kernel<NT, NT_PREV, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn3d_kernel<NT, NT_PREV, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    if (p.shifted)
        hipLaunchKernelGGL((win_attn3d_kernel<NT, NT_PREV, true>), dim3((unsigned)blocks), dim3(THREADS), lds, st,
                           qkv, qkv_bias, table, out, p);
    else
        hipLaunchKernelGGL((win_attn3d_kernel<NT, NT_PREV, false>), dim3((unsigned)blocks), dim3(THREADS), lds, st,
                           qkv, qkv_bias, table, out, p);
    return soc_check_launch();
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)


// ---------------------------------------------------------------------------------------------
// Workgroup schedule.  One workgroup per (window, head) pair is the efficient form (K/V staged once), but
// a pair is an indivisible ~48 us of one CU, so a grid that is not a multiple of the CU count idles most
// of the chip in its last round (897 pairs on 256 CUs: 4 rounds for 3.5 rounds of work).  The last pairs
// are therefore split q ways over query tiles (each part re-stages K/V).  (n_main, q) are chosen by
// simulating the dispatcher -- workgroups are handed to CUs in index order as CUs free up -- with a cost
// model fitted to MI355X measurements (tools/k1_probe.py --stamps / --sweep; the model ranks the forced
// schedules of the sweep within 2-3 us of their measured times): in units of one tile-slot (four SIMDs x one
// 16-query tile, ~15.9k cycles), staging a pair's K/V + bias column costs SIGMA = 0.86, a part that owns n whole
// tiles SIGMA + ceil(n / 4), an unsplit full-window workgroup (25th tile shared) SIGMA + 25/4 + 0.27.
// ---------------------------------------------------------------------------------------------
struct Plan { int n_main, qsplit; };

struct CostModel { double sigma, share; };
constexpr CostModel COST_F32{0.86, 0.27};
// split kernel: a tile-slot is ~2.6x shorter, staging (global-load latency + the operand split) is not
constexpr CostModel COST_SPLIT{2.3, 0.6};
// streaming kernel (round 6): the unit is one 32-query tile on one SIMD (13 chunks, ~11 k cycles); an unsplit workgroup is
// staging + 3 tiles per SIMD + the shared 8-query tile
constexpr CostModel COST_STREAM{0.9, 0.3};        // stamps (tools/k1_probe.py --stamps): staging 14.6 k cycles, 3 tiles per SIMD ~50 k, tail ~5 k

double simulate_tail(long pairs, int NT, int cus, int n_main, int q, bool shared_last, std::vector<double>& heap,
                     const CostModel& cm) {
    const double SIGMA = cm.sigma, SHARE = cm.share;
    // an unsplit workgroup: the shared last tile is a fraction of a tile-slot on top of the whole ones
    const double full = SIGMA + (shared_last ? (NT - 1) / 4 + ((NT - 1) % 4) / 4.0 + (NT == 25 ? 0.25 : 0.0) + SHARE : (double)((NT + 3) / 4));
    const long rounds = n_main / cus, extra = n_main % cus;
    heap.assign(cus, rounds * full);
    for (long i = 0; i < extra; ++i) heap[i] += full;
    std::make_heap(heap.begin(), heap.end(), std::greater<double>());
    double part_cost[32];
    for (int pt = 0; pt < q; ++pt) {
        const int n = (NT - pt + q - 1) / q;              // tiles pt, pt+q, ... < NT
        part_cost[pt] = n > 0 ? SIGMA + (n + 3) / 4 : 0.05;
    }
    double makespan = rounds * full + (extra ? full : 0.0);
    for (long pr = n_main; pr < pairs; ++pr)
        for (int pt = 0; pt < q; ++pt) {
            std::pop_heap(heap.begin(), heap.end(), std::greater<double>());
            const double t = heap.back() + part_cost[pt];
            heap.back() = t;
            std::push_heap(heap.begin(), heap.end(), std::greater<double>());
            if (t > makespan) makespan = t;
        }
    // the real dispatcher is not an ideal list scheduler: among equal makespans prefer fewer workgroups
    return makespan + 1e-4 * (double)(n_main + (pairs - n_main) * q);
}

Plan plan_schedule(long pairs, int NT, int cus, bool shared_last, int split) {      // split: 0 f32 form, 1 streaming, 2 round-3 split form
    static std::mutex mu;
    static std::map<std::tuple<long, int, int, bool, int>, Plan> cache;     // one entry per launch geometry
    const auto key = std::make_tuple(pairs, NT, cus, shared_last, split);
    const CostModel& cm = split == 1 ? COST_STREAM : split == 2 ? COST_SPLIT : COST_F32;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    std::vector<double> heap;
    Plan best{(int)pairs, 1};
    double best_t = simulate_tail(pairs, NT, cus, (int)pairs, 1, shared_last, heap, cm);
    const int max_q = NT < 8 ? NT : 8;
    // only the last two rounds' worth of pairs are candidates for splitting: earlier rounds are full anyway
    const long lo = pairs > 2L * cus ? (pairs - 2L * cus) / 8 * 8 : 0;
    for (int q = 2; q <= max_q; ++q)
        for (long nm = lo; nm < pairs; nm += 8) {
            const double t = simulate_tail(pairs, NT, cus, (int)nm, q, shared_last, heap, cm);
            if (t < best_t - 1e-9) { best_t = t; best = Plan{(int)nm, q}; }
        }
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = best;
    return best;
}

#ifdef SOC_K1_TUNE
int g_force_n_main = 0, g_force_qsplit = 0;
#endif

}  // namespace

#ifdef SOC_K1_TUNE
extern "C" void soc_debug_force_k1_plan(int n_main, int qsplit) { g_force_n_main = n_main; g_force_qsplit = qsplit; }
#endif

extern "C" int soc_win_attn3d_f32(const float* qkv, const float* qkv_bias, const float* bias_table,
                                  float* out, int B, int D, int H, int W, int C, int n_heads,
                                  int win_d, int win_h, int win_w, int shift_d, int shift_h,
                                  int shift_w, int tab_d, int tab_h, int tab_w, int split_arith, void* stream) {
    if (!qkv || !qkv_bias || !bias_table || !out) return SOC_EINVAL;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || n_heads <= 0) return SOC_EINVAL;
    if (win_d <= 0 || win_h <= 0 || win_w <= 0 || tab_d <= 0 || tab_h <= 0 || tab_w <= 0) return SOC_EINVAL;
    if (shift_d < 0 || shift_h < 0 || shift_w < 0 || shift_d >= win_d || shift_h >= win_h || shift_w >= win_w)
        return SOC_EINVAL;
    if (C != n_heads * HD) return SOC_EUNSUPPORTED;
    WinParams p;
    p.B = B; p.D = D; p.H = H; p.W = W; p.C = C; p.nH = n_heads;
    p.wd = win_d; p.wh = win_h; p.ww = win_w;
    p.sd = shift_d; p.sh = shift_h; p.sw = shift_w;
    p.td = tab_d; p.th = tab_h; p.tw = tab_w;
    p.N = win_d * win_h * win_w;
    if (p.N > NP_MAX || p.N > tab_d * tab_h * tab_w) return SOC_EUNSUPPORTED;
    p.NT = (p.N + 15) / 16;
    p.nwd = (D + win_d - 1) / win_d; p.nwh = (H + win_h - 1) / win_h; p.nww = (W + win_w - 1) / win_w;
    p.Dp = p.nwd * win_d; p.Hp = p.nwh * win_h; p.Wp = p.nww * win_w;
    p.table_len = (2 * tab_d - 1) * (2 * tab_h - 1) * (2 * tab_w - 1);
    p.shifted = (shift_d | shift_h | shift_w) != 0;
#ifdef SOC_K1_STAMPS
    p.dbg = g_dbg;
    if (!p.dbg) return SOC_EINVAL;
#endif
    if ((long)B * D * H * W * 3 * C >= (1L << 31)) return SOC_EUNSUPPORTED;  // int token offsets
    const long pairs = (long)B * p.nwd * p.nwh * p.nww * n_heads;
    const bool full_window = win_d == 8 && win_h == 7 && win_w == 7 && tab_d == 8 && tab_h == 7 && tab_w == 7;
    // full 8x7x7 windows only; a launch argument, not process state: 1 = the streaming form (round 6), 2 = the round-3 split form
    const int split = !full_window ? 0 : split_arith == 2 ? 2 : split_arith != 0 ? 1 : 0;
    {
        const Plan pl = plan_schedule(pairs, split == 1 ? SNQ : p.NT, num_cus((hipStream_t)stream), full_window, split);
        p.n_main = pl.n_main;
        p.qsplit = pl.qsplit;
    }
#ifdef SOC_K1_TUNE   // diagnostic build only (tools/k1_probe.py --sweep): force a schedule
    if (g_force_qsplit > 0) {
        p.qsplit = g_force_qsplit;
        p.n_main = g_force_n_main < 0 ? 0 : (g_force_n_main > pairs ? (int)pairs : g_force_n_main);
        if (p.qsplit == 1) p.n_main = (int)pairs;
    }
#endif
    const long blocks = p.n_main + (pairs - p.n_main) * p.qsplit;
    hipStream_t st = (hipStream_t)stream;
    // key/query tiles are a compile-time constant (fully unrolled MFMA schedule); a window with
    // fewer tokens runs on the next larger instantiation with the surplus keys masked out.
    if (split == 1)
        return launch_stream(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (split == 2)
        return launch_split(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (full_window)
        return launch_full(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (p.NT <= 7) return launch_nt<7, 0>(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (p.NT <= 10) return launch_nt<10, 7>(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (p.NT <= 13) return launch_nt<13, 10>(qkv, qkv_bias, bias_table, out, p, blocks, st);
    if (p.NT <= 19) return launch_nt<19, 13>(qkv, qkv_bias, bias_table, out, p, blocks, st);
    return launch_nt<25, 19>(qkv, qkv_bias, bias_table, out, p, blocks, st);
}
