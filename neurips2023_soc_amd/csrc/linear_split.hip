// K20: f32 linear layers on the bf16 matrix cores by exact operand splitting (gfx950).
//
//   out[M, N] = mul * act( LN(x [+ x_add])[M, K] . w[N, K]^T + bias ) + residual          (f32 in, f32 out)
//
// Why.  Every pixel-sized linear layer of the path (reference models/video_swin_transformer.py:144-166 qkv / proj,
// :24-37 Mlp; models/deformable_transformer.py:253-263 FFN; models/ops/modules/ms_deform_attn.py:93-116 value / output
// projections) is an f32 GEMM, and gfx950's f32-input MFMA runs at the f32 VECTOR rate: 157 TFLOP/s, 1/16 of the bf16
// matrix rate, with the vector ALU blocked while it runs.  An f32 number is EXACTLY the sum of three bf16 numbers
// (8 + 8 + 8 significant bits, round-to-nearest splits):  a = a0 + a1 + a2,  b = b0 + b1 + b2, and a bf16 x bf16
// product is exact in f32.  The kernel accumulates, in f32 accumulators, the six products
//       a0 b0  +  (a0 b1 + a1 b0)  +  (a0 b2 + a1 b1 + a2 b0)
// and drops a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|, i.e. the size of ONE f32 rounding of the product: the result carries
// f32-level error (tests/test_gpu_kernels.py::test_linear_split_*: error against an f64 reference no larger than the
// f32 library GEMM's), at 6/16 of the f32 MFMA time, and with the vector ALU free beside the matrix cores -- which is
// what makes the fused LayerNorm prologue / GELU / residual epilogues cheap here (on the f32 MFMA path they are
// matrix-pipe time).
//
// Structure.  512-thread workgroups (8 waves = WM x WN), tile BM x BN = 32 MT WM x 32 NT WN, K-steps of 32.
//   * weights are split ONCE per model into a packed image [K/32][3 planes][N pad][32 bf16] (soc_linear_split_pack_f32)
//     whose rows are already in LDS order: a k-step's B tile is three contiguous runs copied straight to LDS;
//   * activations stay f32 in HBM: a thread loads 8 consecutive k of a row (two 16-B loads), splits in registers (11 VALU
//     instructions per 2 elements) and writes one 16-B piece per plane;
//   * a LayerNorm in front of the layer is applied BEHIND it, exactly: the host folds gamma into the weight image and
//     beta into the bias, and with the row statistics (soc_row_stats_f32) and the column sums of the scaled weights
//       LN(x) W^T + b = rstd (x (W diag gamma)^T - mean colsum) + (b + W beta);
//   * LDS rows are 64 B (32 bf16); 16-B chunk c of row r sits at chunk c ^ ((r >> 2) & 3): the ds_read_b128 fragment
//     reads of v_mfma_f32_32x32x16_bf16 (lane = row, 8 consecutive k) and the ds_write_b128 staging writes are both
//     bank-conflict free;
//   * double-buffered LDS, ONE barrier per K-step, the next step's global loads in flight during the MFMAs; the
//     workgroups are persistent and the pipeline runs across tile boundaries (the first K-step of the next tile is
//     fetched under the last MFMAs of the current one), which is what short-K layers (K = 96 ... 384) need;
//   * an XCD works through a contiguous range of tiles (column tiles of one row panel next to each other), so a row
//     panel of x is fetched once per XCD L2.
#include "soc_common.h"
#include <math.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // k per step
constexpr int ROWB = 64;      // bytes per LDS / packed row (32 bf16)
constexpr int THREADS = 512;
constexpr int NPAD_UNIT = 768;   // packed images pad N to a multiple of lcm(256, 96): every tile width divides it

struct SplitParams {
    const float* x;          // [M, K]
    const float* x_add;      // [M, K] or null: A = x + x_add
    const float* stats;      // [M, 2] (mean, rstd) or null: LayerNorm statistics of the rows of x, applied in the epilogue
    const float* colsum;     // [N] column sums of the (gamma-scaled) weights; with stats
    const unsigned char* wp; // packed split weights
    const float* bias;       // [N] or null
    const float* residual;   // [M, N] or null
    const float* mul;        // [M, N] or null
    float* out;              // [M, N], or [M, n_split] when out2 is set
    float* out2;             // [M, N - n_split] or null: columns >= n_split go here (two layers that share their input)
    int n_split;
    long M;
    int N, K, KT, Npad, act;
    int tiles_n, total_tiles;
#ifdef SOC_K20_STAMPS
    unsigned long long* dbg; // diagnostic build only: [block][wave][8] cycle sums per phase
#endif
};

#ifdef SOC_K20_STAMPS
static unsigned long long* g_dbg20 = nullptr;
extern "C" void soc_debug_set_buffer_k20(void* ptr) { g_dbg20 = (unsigned long long*)ptr; }
#define STAMP20(i)                                                     \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        phase_sum[i] += now_ - last_;                                  \
        last_ = now_;                                                  \
    } while (0)
#else
#define STAMP20(i) do {} while (0)
#endif

// exact (erf) GELU with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, ~1 ulp of the 1 + erf it is added to),
// branch-free (libm's erff is piecewise: 16 divergent branches per tile row group) -- the form K13 uses
// The constants live in vector registers (GeluConsts, filled through an opaque v_mov): the compiler would otherwise feed
// them to v_pk_fma_f32 from SGPR pairs, the operand form that is unsafe beside LDS-DMA + MFMA waves (see the kernel).
struct GeluConsts { float rs2, a0, one, c5, c4, c3, c2, c1, nlog2e, half; };
__device__ __forceinline__ float in_vgpr(float c) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(c));
    return r;
}
__device__ __forceinline__ GeluConsts gelu_consts() {
    return {in_vgpr(0.70710678118654752f), in_vgpr(0.3275911f), in_vgpr(1.0f), in_vgpr(1.061405429f), in_vgpr(-1.453152027f),
            in_vgpr(1.421413741f), in_vgpr(-0.284496736f), in_vgpr(0.254829592f), in_vgpr(-1.4426950408889634f),
            in_vgpr(0.5f)};
}
__device__ __forceinline__ float gelu_erf(float x, const GeluConsts& k) {
    const float z = fabsf(x) * k.rs2;
    const float t = __builtin_amdgcn_rcpf(fmaf(k.a0, z, k.one));
    float p = fmaf(k.c5, t, k.c4);
    p = fmaf(p, t, k.c3);
    p = fmaf(p, t, k.c2);
    p = fmaf(p, t, k.c1);
    const float e = __builtin_amdgcn_exp2f(z * z * k.nlog2e);
    const float erf_abs = fmaf(-p * t, e, k.one);         // erf(|x| / sqrt 2)
    const float half = k.half * x;
    return fmaf(half, copysignf(erf_abs, x), half);        // 0.5 x (1 + erf(x / sqrt 2))
}

// f32 x 8 -> three bf16 x 8 with a0 + a1 + a2 == a exactly (round-to-nearest at every level)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
    bf16x8 h0, h1, h2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        h0[i] = a0; h1[i] = a1; h2[i] = (__bf16)r2;
    }
    p0 = __builtin_bit_cast(u32x4, h0);
    p1 = __builtin_bit_cast(u32x4, h1);
    p2 = __builtin_bit_cast(u32x4, h2);
}

constexpr int SCRW = 36;                       // floats per row of a wave's 32 x 32 epilogue scratch tile
constexpr int SCR_BYTES = 32 * SCRW * 4;       // 4 608 B per wave

template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(THREADS, 2) void linear_split_kernel(const SplitParams p) {
    static_assert(WM * WN == THREADS / 64, "8 waves");
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int A_ITEMS = BM * 4 / THREADS;                     // (row, chunk) items per thread
    static_assert(BM * 4 % THREADS == 0, "A tile must divide over the threads");
    constexpr int B_PIECES = 3 * BN * 4;                          // 16-B pieces per k-step
    static_assert((BN * 4) % 64 == 0, "a wave's 64 pieces must stay inside one plane");
    constexpr int B_ITEMS = (B_PIECES + THREADS - 1) / THREADS;
    constexpr int BUF_BYTES = 3 * (BM + BN) * ROWB;
    static_assert(8 * SCR_BYTES <= BUF_BYTES, "epilogue scratch lives in a tile buffer");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [buf 0][buf 1]
#ifndef SOC_K20_SHARE_CU
    // The kernel owns its CUs.  Measured on MI355X (tools/experiments/pk_mfma_probe.hip): while a wave that issues
    // LDS-DMA loads next to bf16 MFMAs is resident, v_pk_*_f32 instructions with an SGPR source that OTHER waves of the
    // same SIMD execute return wrong low halves in lanes 48..63 -- other kernels included (the dynamic mask head beside
    // this kernel in the pipelined replay: 20-30 % of its launches).  Claiming all 256 architectural VGPRs makes the two
    // waves per SIMD of one workgroup fill the 512-entry register file, so nothing else can be resident beside them;
    // this kernel's own VALU code keeps packed f32 arithmetic away from SGPR operands (checked in the build's ISA test).
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const GeluConsts gk = gelu_consts();
#ifdef SOC_K20_STAMPS
    unsigned long long phase_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_start_ = last_;
#endif

    // ---- this workgroup's tiles: XCD x (blocks b = x mod 8) owns the contiguous tile range [x T8, (x+1) T8) and its
    // blocks stride through it, so concurrently running workgroups of an XCD sit on neighbouring tiles
    const int nb = gridDim.x;
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int stride = (nb + 7 - xcd) >> 3;                       // blocks with this b mod 8
    const int T8 = (p.total_tiles + 7) / 8;
    const int t_begin = xcd * T8, t_end = min((xcd + 1) * T8, p.total_tiles);
    int tile = t_begin + bidx;
    if (tile >= t_end) return;

    // ---- loader roles: item u of a thread = (row, 16-B chunk) of the A tile
    const int a_chunk = tid & 3;                                  // same for every item (THREADS % 4 == 0)
    const int a_row0 = tid >> 2;                                  // + 128 u
    float4 a_raw[A_ITEMS][2], a_add[A_ITEMS][2];
    const bool has_add = p.x_add != nullptr;
    // (keep_addr / keep_alive: the address registers of the per-lane loads stay live until the barrier whose vmcnt wait
    // covers the loads -- free, the kernel claims all 256 VGPRs anyway)
    const void* keep_addr[A_ITEMS][2] = {};
    auto keep_alive = [&]() {
#pragma unroll
        for (int u = 0; u < A_ITEMS; ++u) asm volatile("" :: "v"(keep_addr[u][0]), "v"(keep_addr[u][1]));
    };

    auto issue_a = [&](int t, int kt) {       // A rows of step (t, kt): global -> registers
        const int tm = t / p.tiles_n;
        const int kk = min(kt * BK + 8 * a_chunk, p.K - 8);       // K-tail chunks re-read the row's last chunk (zeroed below)
#pragma unroll
        for (int u = 0; u < A_ITEMS; ++u) {
            const long gm = min((long)tm * BM + a_row0 + (THREADS / 4) * u, p.M - 1);
            const float4* src = reinterpret_cast<const float4*>(p.x + gm * p.K + kk);
            a_raw[u][0] = src[0]; a_raw[u][1] = src[1];
            keep_addr[u][0] = src;
            if (has_add) {
                const float4* s2 = reinterpret_cast<const float4*>(p.x_add + gm * p.K + kk);
                a_add[u][0] = s2[0]; a_add[u][1] = s2[1];
                keep_addr[u][1] = s2;
            }
        }
    };

    // B: the three planes of a k-step's weight tile are contiguous runs of the packed image, copied straight into the
    // LDS buffer by LDS-DMA (no registers; a wave instruction lands 64 pieces = 1 KB contiguously)
    auto dma_b = [&](int t, int kt, int nbuf) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        const unsigned char* wsrc = p.wp + ((long)kt * 3 * p.Npad + tn * BN) * ROWB;
        unsigned char* bdst = lds + nbuf * BUF_BYTES + 3 * BM * ROWB;
#pragma unroll
        for (int u = 0; u < B_ITEMS; ++u) {
            const int q0 = wave * 64 + THREADS * u;               // first piece of this wave instruction (wave-uniform)
            if (B_PIECES % THREADS == 0 || q0 < B_PIECES) {
                const int pl = q0 / (BN * 4);                     // wave-uniform plane
                const int r = q0 - pl * (BN * 4) + lane;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wsrc + (long)pl * p.Npad * ROWB + r * 16),
                    (__attribute__((address_space(3))) void*)(bdst + q0 * 16), 16, 0, 0);
            }
        }
    };

    auto commit = [&](int nbuf, int kt) {      // A registers -> split -> LDS
        unsigned char* base = lds + nbuf * BUF_BYTES;
        const int kc = kt * BK + 8 * a_chunk;
#pragma unroll
        for (int u = 0; u < A_ITEMS; ++u) {
            float v[8] = {a_raw[u][0].x, a_raw[u][0].y, a_raw[u][0].z, a_raw[u][0].w,
                          a_raw[u][1].x, a_raw[u][1].y, a_raw[u][1].z, a_raw[u][1].w};
            if (has_add) {
                const float w[8] = {a_add[u][0].x, a_add[u][0].y, a_add[u][0].z, a_add[u][0].w,
                                    a_add[u][1].x, a_add[u][1].y, a_add[u][1].z, a_add[u][1].w};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += w[i];
            }
            if (kc >= p.K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 0.f;
            }
            u32x4 p0, p1, p2;
            split8(v, p0, p1, p2);
            const int r = a_row0 + (THREADS / 4) * u;
            unsigned char* dst = base + r * ROWB + ((a_chunk ^ ((r >> 2) & 3)) << 4);
            *reinterpret_cast<u32x4*>(dst) = p0;
            *reinterpret_cast<u32x4*>(dst + BM * ROWB) = p1;
            *reinterpret_cast<u32x4*>(dst + 2 * BM * ROWB) = p2;
        }
    };

    // ---- MFMA fragment addresses (byte offsets inside a buffer), one per k-substep
    const int lrow = lane & 31, lh = lane >> 5;
    const int sw = (lrow >> 2) & 3;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        a_off[s] = (wm * 32 * MT + lrow) * ROWB + (((2 * s + lh) ^ sw) << 4);
        b_off[s] = 3 * BM * ROWB + (wn * 32 * NT + lrow) * ROWB + (((2 * s + lh) ^ sw) << 4);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto half = [&](int buf, int s) {        // the MFMAs of k-substep s (16 of the step's 32 k)
        const unsigned char* base = lds + buf * BUF_BYTES;
        bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[i][pl] = *reinterpret_cast<const bf16x8*>(base + a_off[s] + (pl * BM + 32 * i) * ROWB);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[j][pl] = *reinterpret_cast<const bf16x8*>(base + b_off[s] + (pl * BN + 32 * j) * ROWB);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };

    // ---- epilogue: each wave turns its 32 x 32 accumulator tiles (lane = column, registers = rows) into rows of four
    // consecutive columns per lane through a private LDS scratch tile, so that bias / activation / mul / residual are
    // float4 operations and every global access is a full 128-B line (8 rows x 128 B per wave instruction).  The scratch
    // lives in the tile buffer the workgroup has just finished reading (the caller has passed a barrier).
    auto epilogue = [&](int t, int buf) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        float* scr = reinterpret_cast<float*>(lds + buf * BUF_BYTES + wave * SCR_BYTES);
        const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = tn * BN + wn * 32 * NT + 32 * j + c4;       // N % 4 == 0: a lane's four columns are in or out together
            const bool n_ok = n < p.N;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = bv;
            if (p.bias && n_ok) bv = *reinterpret_cast<const float4*>(p.bias + n);
            if (p.stats && n_ok) cs = *reinterpret_cast<const float4*>(p.colsum + n);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * SCRW + lrow] = acc[i][j][e];
                    acc[i][j][e] = 0.f;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const long m_base = (long)tm * BM + wm * 32 * MT + 32 * i + r8;
                float4 v[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const float4*>(scr + (8 * it + r8) * SCRW + c4);
                if (p.stats) {
                    // LayerNorm in front of the layer, applied behind it: with W' = W diag(gamma) in the weight image,
                    //   LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (b + W beta)
                    // exactly -- the loader streams the raw rows, and the per-row statistics are needed only here
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const long m = min(m_base + 8 * it, p.M - 1);
                        const float2 st = *reinterpret_cast<const float2*>(p.stats + 2 * m);
                        v[it].x = (v[it].x - st.x * cs.x) * st.y; v[it].y = (v[it].y - st.x * cs.y) * st.y;
                        v[it].z = (v[it].z - st.x * cs.z) * st.y; v[it].w = (v[it].w - st.x * cs.w) * st.y;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) { v[it].x += bv.x; v[it].y += bv.y; v[it].z += bv.z; v[it].w += bv.w; }
                if (p.act == 1) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        v[it].x = fmaxf(v[it].x, 0.f); v[it].y = fmaxf(v[it].y, 0.f);
                        v[it].z = fmaxf(v[it].z, 0.f); v[it].w = fmaxf(v[it].w, 0.f);
                    }
                } else if (p.act == 2) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        v[it].x = gelu_erf(v[it].x, gk); v[it].y = gelu_erf(v[it].y, gk);
                        v[it].z = gelu_erf(v[it].z, gk); v[it].w = gelu_erf(v[it].w, gk);
                    }
                }
                if (p.mul) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const long m = min(m_base + 8 * it, p.M - 1);
                        const float4 mu = *reinterpret_cast<const float4*>(p.mul + m * p.N + (n_ok ? n : 0));
                        v[it].x *= mu.x; v[it].y *= mu.y; v[it].z *= mu.z; v[it].w *= mu.w;
                    }
                }
                if (p.residual) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const long m = min(m_base + 8 * it, p.M - 1);
                        const float4 rr = *reinterpret_cast<const float4*>(p.residual + m * p.N + (n_ok ? n : 0));
                        v[it].x += rr.x; v[it].y += rr.y; v[it].z += rr.z; v[it].w += rr.w;
                    }
                }
                // two output tensors: columns [0, n_split) -> out, [n_split, N) -> out2 (n_split % 4 == 0)
                const bool second = p.out2 != nullptr && n >= p.n_split;
                float* obase = second ? p.out2 : p.out;
                const int ocols = p.out2 ? (second ? p.N - p.n_split : p.n_split) : p.N;
                const int ocol = second ? n - p.n_split : n;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const long m = m_base + 8 * it;
                    if (n_ok && m < p.M) *reinterpret_cast<float4*>(obase + m * ocols + ocol) = v[it];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    };

    // ---- flattened (tile, k-step) pipeline.  Step k of this workgroup's sequence computes from LDS buffer k & 1 while
    //   * the A rows of step k+1 -- in registers since step k-1 -- are split and written to the other buffer,
    //   * the A rows of step k+2 are fetched into the freed registers and the B planes of step k+1 are DMA-ed,
    // so a global load has a whole step to land and nothing in a step waits for memory but the closing barrier.  The two
    // waves of a SIMD (w and w + 4) run the memory part at different times: waves 4-7 put the first half of their MFMAs
    // in front of it, waves 0-3 behind it, so one of them feeds the matrix pipe while the other issues loads and stores.
    auto adv = [&](int& t, int& kt) { if (++kt == p.KT) { kt = 0; t += stride; } };
    int tc = tile, kc = 0;
    int t1 = tc, k1 = kc; adv(t1, k1);
    int t2 = t1, k2 = k1; adv(t2, k2);
    const bool late = wave >= 4;
    issue_a(tc, kc);
    dma_b(tc, kc, 0);
    commit(0, kc);
    keep_alive();                         // commit() has waited for the loads of issue_a
    if (t1 < t_end) issue_a(t1, k1);
    __syncthreads();
    keep_alive();
    int buf = 0;
    STAMP20(0);
    while (true) {
        const bool has1 = t1 < t_end, has2 = has1 && t2 < t_end;
        if (late) half(buf, 0);
        STAMP20(3);
        if (has1) commit(buf ^ 1, k1);
        STAMP20(1);
        if (has2) issue_a(t2, k2);
        if (has1) dma_b(t1, k1, buf ^ 1);
        STAMP20(2);
        if (!late) half(buf, 0);
        half(buf, 1);
        STAMP20(3);
        if (kc == p.KT - 1) {
            __syncthreads();              // every wave has finished reading `buf`: it becomes the epilogue scratch
            STAMP20(4);
            epilogue(tc, buf);
            STAMP20(5);
        }
        if (!has1) break;
        __syncthreads();
        keep_alive();
        STAMP20(6);
        buf ^= 1;
        tc = t1; kc = k1; t1 = t2; k1 = k2; adv(t2, k2);
    }
#ifdef SOC_K20_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) d[i] = phase_sum[i];
        d[7] = __builtin_amdgcn_s_memtime() - t_start_;
    }
#endif
}

// ---- weight packing: w [N][K] f32 -> [KT][3][Npad][32 bf16], chunk c of row n at chunk c ^ ((n >> 2) & 3)
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                         int N, int K, int KT, int Npad) {
    const long total = (long)KT * Npad * 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx & 3);
        const long rn = idx >> 2;
        const int n = (int)(rn % Npad), kt = (int)(rn / Npad);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kt * BK + 8 * c + i;
            v[i] = (n < N && k < K) ? w[(long)n * K + k] : 0.f;
        }
        u32x4 p0, p1, p2;
        split8(v, p0, p1, p2);
        unsigned char* dst = out + (((long)kt * 3) * Npad + n) * ROWB + ((c ^ ((n >> 2) & 3)) << 4);
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + (long)Npad * ROWB) = p1;
        *reinterpret_cast<u32x4*>(dst + 2L * Npad * ROWB) = p2;
    }
}

// ---- LayerNorm row statistics: (mean, rstd) per row, two-pass over the row held in registers (as nn.LayerNorm: biased
// variance, eps inside the square root).  G lanes share a row.
template <int G>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, long M,
                                                        int K, float eps) {
    constexpr int MAXV = 8;                                     // float4 per lane: K <= 4 * G * MAXV
    const long row = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (row >= M) return;
    const float4* src = reinterpret_cast<const float4*>(x + row * K);
    const int nv = K / 4;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int q = sub + G * i;
        v[i] = q < nv ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)K;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (sub + G * i < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q2 += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) q2 += __shfl_xor(q2, o);
    if (sub == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = 1.0f / sqrtf(q2 / (float)K + eps);
    }
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)

template <int MT, int NT, int WM, int WN>
int launch_cfg(const SplitParams& p0, hipStream_t st) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    SplitParams p = p0;
    const long tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    if (tiles_m * p.tiles_n >= (1L << 31)) return SOC_EUNSUPPORTED;
    p.total_tiles = (int)(tiles_m * p.tiles_n);
#ifndef SOC_K20_DBG_PAD
#define SOC_K20_DBG_PAD 0
#endif
    const size_t lds = 2 * 3 * (BM + BN) * ROWB + SOC_K20_DBG_PAD;
    if (lds > 160 * 1024) return SOC_EUNSUPPORTED;
    const void* fn = reinterpret_cast<const void*>(linear_split_kernel<MT, NT, WM, WN>);
    static bool attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SOC_ELAUNCH;
        attr_set[dev] = true;
    }
    long blocks = num_cus(st);
#ifdef SOC_K20_DBG_BLOCKS_PER_CU
    blocks *= SOC_K20_DBG_BLOCKS_PER_CU;
#endif
    if (blocks > p.total_tiles) blocks = p.total_tiles;
    hipLaunchKernelGGL((linear_split_kernel<MT, NT, WM, WN>), dim3((unsigned)blocks), dim3(THREADS), lds, st, p);
    return soc_check_launch();
}

}  // namespace

extern "C" size_t soc_linear_split_packed_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    const long KT = (K + BK - 1) / BK, Npad = ((long)N + NPAD_UNIT - 1) / NPAD_UNIT * NPAD_UNIT;
    return (size_t)(KT * 3 * Npad * ROWB);
}

extern "C" int soc_linear_split_pack_f32(const float* w, void* packed, int N, int K, void* stream) {
    if (!w || !packed || N <= 0 || K <= 0) return SOC_EINVAL;
    const int KT = (K + BK - 1) / BK, Npad = (N + NPAD_UNIT - 1) / NPAD_UNIT * NPAD_UNIT;
    const long total = (long)KT * Npad * 4;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<unsigned char*>(packed), N, K, KT, Npad);
    return soc_check_launch();
}

extern "C" int soc_row_stats_f32(const float* x, float* stats, long M, int K, float eps, void* stream) {
    if (!x || !stats || M < 0 || K <= 0) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (K % 4 != 0 || K > 4 * 32 * 8 || ((uintptr_t)x & 15)) return SOC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (K <= 256) {
        const long blocks = (M * 8 + 255) / 256;
        hipLaunchKernelGGL(row_stats_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, x, stats, M, K, eps);
    } else {
        const long blocks = (M * 32 + 255) / 256;
        hipLaunchKernelGGL(row_stats_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, x, stats, M, K, eps);
    }
    return soc_check_launch();
}

extern "C" int soc_linear_split_f32(const float* x, const float* x_add, const float* row_stats, const float* w_colsum,
                                    const void* w_packed, const float* bias,
                                    const float* residual, const float* mul, float* out, float* out2, int n_split,
                                    long M, int N, int K, int act, int tile_cfg, void* stream) {
    if (!x || !w_packed || !out || M < 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if ((row_stats != nullptr) != (w_colsum != nullptr)) return SOC_EINVAL;
    if (out2 && (n_split <= 0 || n_split >= N || n_split % 4 != 0 || residual || mul)) return SOC_EINVAL;
    if (K % 8 != 0 || N % 4 != 0) return SOC_EUNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)(x_add ? x_add : x)) & 15) return SOC_EUNSUPPORTED;
    SplitParams p;
    p.x = x; p.x_add = x_add; p.stats = row_stats; p.colsum = w_colsum;
    p.wp = reinterpret_cast<const unsigned char*>(w_packed);
    p.bias = bias; p.residual = residual; p.mul = mul; p.out = out; p.out2 = out2; p.n_split = out2 ? n_split : 0;
    p.M = M; p.N = N; p.K = K; p.KT = (K + BK - 1) / BK;
    p.Npad = (N + NPAD_UNIT - 1) / NPAD_UNIT * NPAD_UNIT;
    p.act = act; p.tiles_n = 0; p.total_tiles = 0;
#ifdef SOC_K20_STAMPS
    p.dbg = g_dbg20;
#endif
    hipStream_t st = (hipStream_t)stream;
    switch (tile_cfg) {
        case 0: return launch_cfg<2, 2, 2, 4>(p, st);    // 128 x 256
        case 1: return launch_cfg<2, 2, 4, 2>(p, st);    // 256 x 128
        case 2: return launch_cfg<2, 1, 2, 4>(p, st);    // 128 x 128
        case 3: return launch_cfg<1, 3, 8, 1>(p, st);    // 256 x 96
        case 4: return launch_cfg<1, 1, 4, 2>(p, st);    // 128 x 64
        default: return SOC_EINVAL;
    }
}
