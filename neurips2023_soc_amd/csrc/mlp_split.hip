// K23: a two-layer perceptron block in one kernel, on the bf16 matrix cores (exact three-way operand split).
//
//   out[M, C] = LN2(act(LN(x)[M, C] . W1[F, C]^T + b1) . W2[C, F]^T + b2 + residual)               (f32 in, f32 out)
//
// Replaces  x + mlp(norm2(x))  = norm2 -> fc1 -> GELU -> fc2 -> shortcut of SwinTransformerBlock3D.forward_part2 (reference
// models/video_swin_transformer.py:24-37, 262-272; optionally together with norm1 of the NEXT block, :219) and
// norm2(src + linear2(relu(linear1(src)))) of the deformable encoder's forward_ffn (models/deformable_transformer.py:253-263).
// The [M, F] hidden tensor (177 MB per block at Video-Swin stage 0, 316 MB per encoder layer) never exists.
//
// Data flow:
//   * a wave owns RT 16-row tiles of x and keeps them, normalised and split, as bf16 MFMA fragments (K = C);
//   * the hidden layer is produced 32 columns (one chunk) at a time: H^T[32 x 16] = W1_chunk . x^T on
//     v_mfma_f32_16x16x32_bf16 (six products per k-step, smallest first), starting from b1; lane (row m, kq) of the two
//     16 x 16 accumulator tiles holds hidden columns 4 kq .. 4 kq + 3 of each -- once activated and split, that IS the B operand
//     of the second product: eight k values per lane, in an order the packed image of W2 mirrors (k-permutation at pack time);
//   * out^T[C x 16] += W2_chunk . H_chunk^T over the chunks in C / 4 accumulator registers per tile; b2 and the residual are
//     added last, then (optionally) LayerNorm over the row, which lives in the four lanes (m, kq = 0..3); 16-B stores.
// The weights stream through LDS:
//   * per chunk a W1 block and a W2 block, pre-split, pre-permuted, laid out as the fragments are read (soc_mlp_split_pack_f32),
//     travel through a RING of NSLOT slots filled by LDS-DMA up to NSLOT - 1 pieces ahead (a piece = 1 / SB of a block) and
//     handed over with a COUNTED vmcnt wait: only the piece needed next must have landed;
//   * the LDS-DMA is issued from inline assembly (see dma below): with a compiler-visible LDS-DMA in flight every LDS wait of
//     the loop degrades to lgkmcnt(0) and fragment reads cannot run ahead of the MFMAs;
//   * weight fragments are read PF groups ahead of the MFMAs that consume them (sched_group_barrier pins the order).
// Every row is taken: a workgroup owns a balanced contiguous share of the row tiles and deals the tiles of a part-filled pass
// round-robin over its waves (one per SIMD first); for few rows the hidden dimension is split over `nfs` workgroup columns
// that write partial sums, which a reduce kernel adds in a fixed order (deterministic) together with b2 / residual / LN2.
// C <= 256: eight 256-register waves (two per SIMD); C = 384 / 512 (Swin-B stage 2): four 512-register waves, blocks travel as
// half-block pieces.
// Measured null and kept as diagnostic variants only (SOC_K23_VARIANTS): waves 4..7 one piece behind waves 0..3, the
// activation at raised wave priority, deeper fragment prefetch, four-wave forms for C <= 256 (tools/experiments/README.md).
// Co-residence rule: the whole register file is claimed, waves retire behind a barrier, packed f32 code
// has VGPR operands only (tests/test_isa_rules.py).
#include "soc_common.h"
#include "split_math.h"
#include <atomic>
#include <type_traits>

namespace {

using namespace soc_split;
constexpr int MAX_THREADS = 512;       // the image pads a block to whole DMA rounds of 512 threads (256-thread forms take two)

template <int C>
struct Geo {
    static constexpr int KS = C / 32;                                   // k-steps of the first product
    static constexpr int OT = C / 16;                                   // output tiles of the second product
    static constexpr int BLK_U4 = OT * 3 * 64;                          // 16-B pieces of a weight block (W1: 2 KS groups, W2: OT)
    static constexpr int BLKP_U4 = (BLK_U4 + MAX_THREADS - 1) / MAX_THREADS * MAX_THREADS;     // a block in the image / a ring slot
    static_assert(2 * KS == OT, "both block kinds hold the same number of fragment groups");
};

template <int N>
__device__ __forceinline__ void handoff() {        // my part of the next block has landed; everyone is done with this one
#ifdef SOC_K23_NO_BARRIER          // diagnostic build only (tools/experiments/k23_time.py, K23_EXTRA_FLAGS): what the ring would cost
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");         // WITHOUT its hand-off barrier -- results are wrong (races)
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// DBG (diagnostics only) bit 0: no LDS-DMA inside the block loop, bit 1: no MFMA work (stream ceiling), bit 2: no fragment
// reads after a piece's first group, bit 3: no activation.  PF: fragment groups read ahead.  STAG bit 0: waves 4..7 run one ring
// piece behind waves 0..3, bit 1: the activation runs at raised wave priority, bit 2: the LDS-DMA instructions of a piece are
// issued between the MFMA groups of the step instead of in front of them.
// NW waves per workgroup: 8 (two per SIMD, 256 registers each) or 4 (one per SIMD, 512 registers: wider rows or more row tiles
// per wave, so that a weight fragment read from LDS feeds more MFMAs).  RT row tiles per wave.
template <int C, int ACT, bool HAS_LN, int NW, int RT, int NSLOT, int SB, int STAG, int PF, int DBG>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp_split_kernel(
    const float* __restrict__ x, const u32x4* __restrict__ img, const float* __restrict__ b1, const float* __restrict__ b2,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, const float* __restrict__ res,
    const float* __restrict__ gamma2, const float* __restrict__ beta2, float eps2, float* __restrict__ out,
    float* __restrict__ out_sum, long M, int F, int nrg, int nfs, int res_ln) {
    using G = Geo<C>;
    constexpr int THREADS = NW * 64;
    // a block travels in SB ring pieces (a piece = a slot = one DMA round set): P LDS-DMA instructions per thread and piece
    constexpr int KS = G::KS, OT = G::OT, BLKP = G::BLKP_U4, SLOT = BLKP / SB, P = SLOT / THREADS;
    static_assert(OT % SB == 0 && SLOT % THREADS == 0 && (SB == 1 || BLKP == G::BLK_U4), "pieces are whole fragment groups");
    static_assert((STAG & 3) == 0 || NW == 8, "the stagger pairs the two waves of a SIMD");
    constexpr int D = NSLOT - 1 - ((STAG & 1) ? 1 : 0);                       // pieces in flight ahead of the one being consumed
    static_assert(D >= 1 && (D - 1) * P < 64, "ring depth");
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* slots = lds;
    float* gs = reinterpret_cast<float*>(lds + NSLOT * SLOT);           // gamma [C], beta [C], then this range's b1
    float* bs = gs + C;
    float* b1s = bs + C;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                       // own the CU
    if (NW == 4) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");   // one wave per SIMD: the whole 512-entry file
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int fs = blockIdx.x % nfs, g = blockIdx.x / nfs;              // blocks b and b + 8 share an XCD: with nfs | 8 an XCD
    const int NC = F / 32;                                              // streams one hidden range only
    const int c0 = (int)((long)fs * NC / nfs), c1 = (int)((long)(fs + 1) * NC / nfs);
    const int nb = 2 * (c1 - c0), nq = nb * SB;                         // blocks / ring pieces of this range
    const long ntiles = (M + 15) >> 4;
    const long t0 = (long)g * ntiles / nrg, t1 = (long)(g + 1) * ntiles / nrg;
    float* o = out + (nfs > 1 ? (long)fs * M * C : 0);
    for (int i = tid; i < (c1 - c0) * 32; i += THREADS) b1s[i] = b1[c0 * 32 + i];
    if (HAS_LN)
        for (int i = tid; i < C; i += THREADS) { gs[i] = gamma[i]; bs[i] = beta[i]; }
    // block `blk` of this hidden range -> ring slot `slot`: wave-uniform base + 16 B per lane.  The LDS-DMA is issued from
    // inline assembly on purpose: while the compiler knows of an LDS-DMA in flight it turns every LDS wait of the loop into
    // lgkmcnt(0) (fragment reads could not run ahead of the MFMAs) and would pair its own vmcnt(0) with LDS reads; the ring's
    // waits are the counted ones in handoff().  m0 is touched by these statements only (tests/test_isa_rules.py).
    const char* ibase = reinterpret_cast<const char*>(img + (long)c0 * 2 * BLKP);
    const unsigned voff = (unsigned)tid * 16u;
    const unsigned lds_slots = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)slots + (unsigned)wave * 1024u;
    auto dma = [&](int piece, int slot) {
        const char* src = ibase + (long)piece * (SLOT * 16);
        const unsigned dst = lds_slots + (unsigned)slot * (SLOT * 16);
#pragma unroll
        for (int u = 0; u < P; ++u)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         ::"s"(dst + u * (THREADS * 16)), "v"(voff), "s"(src + u * (THREADS * 16)) : "memory");
    };
    auto dma_one = [&](int piece, int slot, int u) {      // instruction u of a piece's P (STAG bit 2: spread between the MFMAs)
        const char* src = ibase + (long)piece * (SLOT * 16);
        const unsigned dst = lds_slots + (unsigned)slot * (SLOT * 16);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     ::"s"(dst + u * (THREADS * 16)), "v"(voff), "s"(src + u * (THREADS * 16)) : "memory");
    };
    auto next_slot = [](int s) { return s + 1 == NSLOT ? 0 : s + 1; };
    __syncthreads();
    // One pass = up to 8 RT row tiles through the whole hidden range.  NRT (compile time) = this wave's tiles in the pass:
    // the accumulator updates are straight-line code for every count, a wave without tiles only feeds the ring.
    auto pass = [&](auto nrt_c, auto late_c, long pt) {
        constexpr int NRT = decltype(nrt_c)::value;
        constexpr bool LATE = decltype(late_c)::value;
        constexpr int NX = NRT > 0 ? NRT : 1;
        bf16x8 xb[NX][KS][3];
        float ln_mean[NX], ln_rstd[NX];         // kept for a residual that is LN(x) itself (res_ln: the encoder's norm1 folded in)
        {
            float4 xn[NX][KS][2];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const long m = min((pt + NW * rt + wave) * 16 + r, M - 1);
                const float4* xp = reinterpret_cast<const float4*>(x + m * C + 8 * kq);
#pragma unroll
                for (int s = 0; s < KS; ++s) { xn[rt][s][0] = xp[8 * s]; xn[rt][s][1] = xp[8 * s + 1]; }
            }
            int ps = 0;
#pragma unroll
            for (int b = 0; b < D; ++b)
                if (b < nq) { dma(b, ps); ps = next_slot(ps); }
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                float v[KS][8];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    v[s][0] = xn[rt][s][0].x; v[s][1] = xn[rt][s][0].y; v[s][2] = xn[rt][s][0].z; v[s][3] = xn[rt][s][0].w;
                    v[s][4] = xn[rt][s][1].x; v[s][5] = xn[rt][s][1].y; v[s][6] = xn[rt][s][1].z; v[s][7] = xn[rt][s][1].w;
                }
                if (HAS_LN) {   // as nn.LayerNorm: two-pass mean / variance over the row, which lives in lanes (r, kq = 0..3)
                    const float inv_c = in_vgpr(1.0f / C), eps_v = in_vgpr(eps);
                    float sm = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int i = 0; i < 8; ++i) sm += v[s][i];
                    sm += __shfl_xor(sm, 16);
                    sm += __shfl_xor(sm, 32);
                    const float mean = sm * inv_c;
                    float q = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int i = 0; i < 8; ++i) { v[s][i] -= mean; q = fmaf(v[s][i], v[s][i], q); }
                    q += __shfl_xor(q, 16);
                    q += __shfl_xor(q, 32);
                    const float rstd = rsqrtf(fmaf(q, inv_c, eps_v));
                    ln_mean[rt] = mean; ln_rstd[rt] = rstd;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float4* gp = reinterpret_cast<const float4*>(gs + 32 * s + 8 * kq);
                        const float4* ep = reinterpret_cast<const float4*>(bs + 32 * s + 8 * kq);
                        const float4 ga = gp[0], gb = gp[1], ea = ep[0], eb = ep[1];
                        const float gg[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                        const float bb[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[s][i] = fmaf(v[s][i] * rstd, gg[i], bb[i]);
                    }
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) split8(v[s], xb[rt][s][0], xb[rt][s][1], xb[rt][s][2]);
            }
        }
        f32x4 acc2[NX][OT];
#pragma unroll
        for (int rt = 0; rt < NX; ++rt)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc2[rt][ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 acc1[NX][2];          // a chunk of the hidden layer between its two blocks: H^T tiles j = 0, 1, bias included
        bf16x8 hb[NX][3];           // ... and once activated and split: the B operand of the second product
        // ---- the two kinds of block, one ring piece (1 / SB of a block: NGS fragment groups) at a time.  Fragment groups are
        // read PF groups ahead of the MFMAs that consume them.
        constexpr int NGS = OT / SB;
        constexpr bool SPREAD = (STAG & 4) != 0 && NRT > 0 && !(DBG & 2);
        int dq = -1, dsl = 0;                   // the piece whose LDS-DMA instructions this step still has to issue (SPREAD)
        auto spread = [&](int gi) {             // instruction u goes out behind fragment group (u + 1) NGS / (P + 1) - 1
            if constexpr (SPREAD) {
#pragma unroll
                for (int u = 0; u < P; ++u)
                    if (gi == ((u + 1) * NGS) / (P + 1) && dq >= 0) dma_one(dq, dsl, u);
            }
        };
        auto frag = [&](bf16x8 (&wf)[3], const u32x4* wl, int gi) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[pl] = __builtin_bit_cast(bf16x8, wl[(gi * 3 + pl) * 64]);
        };
        // piece `sb` of W1 block `blk` (even) in slot `sl`: hidden columns [16 blk, 16 blk + 32) of this range; fragment groups
        // (j, s) = (g / KS, g % KS), j = hidden tile
        auto first = [&](int blk, auto sb_c, int sl) {
            constexpr int sb = decltype(sb_c)::value;
            if ((DBG & 2) || NRT == 0) return;
            const u32x4* wl = slots + sl * SLOT + lane;
            if (sb == 0) {      // the accumulators start from b1 (lane (r, kq) holds hidden columns 4 kq .. + 3 of each tile)
                const f32x4* bp = reinterpret_cast<const f32x4*>(b1s + 16 * blk + 4 * kq);
                const f32x4 bq0 = bp[0], bq1 = bp[4];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) { acc1[rt][0] = bq0; acc1[rt][1] = bq1; }
            }
            bf16x8 wf[PF + 1][3];
#pragma unroll
            for (int q = 0; q < PF; ++q) frag(wf[q], wl, q);
#pragma unroll
            for (int gi = 0; gi < NGS; ++gi) {
                if (gi + PF < NGS && !((DBG & 4) && gi > 0)) frag(wf[(gi + PF) % (PF + 1)], wl, gi + PF);
                const int g = sb * NGS + gi, j = g / KS, s = g % KS;
                const int cur = (DBG & 4) ? 0 : gi % (PF + 1);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) mfma6(acc1[rt][j], wf[cur], xb[rt][s][0], xb[rt][s][1], xb[rt][s][2]);
                // the reads go out in front of this group's MFMAs; PF groups ahead, not all of them
                if (gi + PF < NGS && !(DBG & 4)) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * NRT, 0);
                __builtin_amdgcn_sched_barrier(0);
                spread(gi);
            }
        };
        // piece `sb` of a W2 block in slot `sl`: (activation and split of the chunk first), then its output tiles get the
        // chunk's 32 hidden columns
        auto second = [&](auto sb_c, int sl) {
            constexpr int sb = decltype(sb_c)::value;
            if ((DBG & 2) || NRT == 0) return;
            const u32x4* wl = slots + sl * SLOT + lane;
            bf16x8 wf[PF + 1][3];
#pragma unroll
            for (int q = 0; q < PF; ++q) frag(wf[q], wl, q);
            if (sb == 0) {
                if (STAG & 2) __builtin_amdgcn_s_setprio(1);   // the vector work goes in front of the partner wave's MFMAs
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = acc1[rt][i >> 2][i & 3];
                    if (DBG & 8) {
                    } else if (ACT == 1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                    } else {
                        const GeluK gk = gelu_k();
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i], gk);
                    }
                    split8(v, hb[rt][0], hb[rt][1], hb[rt][2]);
                }
                if (STAG & 2) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int gi = 0; gi < NGS; ++gi) {
                if (gi + PF < NGS && !((DBG & 4) && gi > 0)) frag(wf[(gi + PF) % (PF + 1)], wl, gi + PF);
                const int ot = sb * NGS + gi;
                const int cur = (DBG & 4) ? 0 : gi % (PF + 1);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) mfma6(acc2[rt][ot], wf[cur], hb[rt][0], hb[rt][1], hb[rt][2]);
                if (gi + PF < NGS && !(DBG & 4)) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * NRT, 0);
                __builtin_amdgcn_sched_barrier(0);
                spread(gi);
            }
        };
        if (nq >= D) handoff<(D - 1) * P>(); else handoff<0>();         // piece 0 has landed
        // Ring step t: the piece D ahead goes out, a piece is consumed, the next one is waited for.  Waves 0..3 consume piece t;
        // with the stagger (LATE) waves 4..7 consume piece t - 1, whose slot is `pslot`: on every SIMD one wave then starts a
        // step with the activation (vector ALU, at raised priority) while its partner starts with MFMAs.
        int slot = 0, pslot = 0, dslot = D % NSLOT, q = 0;
        // piece K of a chunk: K < SB -> W1 piece K, else W2 piece K - SB
        auto piece = [&](auto k_c, int blk, int sl) {
            constexpr int K_ = decltype(k_c)::value;
            if constexpr (K_ < SB) first(blk, std::integral_constant<int, K_>{}, sl);
            else second(std::integral_constant<int, K_ - SB>{}, sl);
        };
        auto step = [&](auto k_c, int blk) {
            constexpr int K_ = decltype(k_c)::value;
            const bool more = q + D < nq;
            if (!(DBG & 1) && more) {
                if constexpr (SPREAD && !LATE) { dq = q + D; dsl = dslot; }
                else dma(q + D, dslot);
                dslot = next_slot(dslot);
            } else {
                dq = -1;
            }
            if constexpr (!LATE) piece(k_c, blk, slot);
            else if (K_ > 0) piece(std::integral_constant<int, (K_ > 0 ? K_ - 1 : 0)>{}, blk, pslot);
            else if (blk > 0) piece(std::integral_constant<int, 2 * SB - 1>{}, blk - 2, pslot);
            if (q + 1 < nq) {
                if (more) handoff<(D - 1) * P>(); else handoff<0>();
            }
            pslot = slot;
            slot = next_slot(slot);
            ++q;
        };
        for (int blk = 0; blk < nb; blk += 2) {
            step(std::integral_constant<int, 0>{}, blk);
            step(std::integral_constant<int, 1>{}, blk);
            if constexpr (SB >= 2) {
                step(std::integral_constant<int, 2>{}, blk);
                step(std::integral_constant<int, 3>{}, blk);
            }
            if constexpr (SB >= 4) {
                step(std::integral_constant<int, 4>{}, blk);
                step(std::integral_constant<int, 5>{}, blk);
                step(std::integral_constant<int, 6>{}, blk);
                step(std::integral_constant<int, 7>{}, blk);
            }
        }
        if constexpr (LATE) piece(std::integral_constant<int, 2 * SB - 1>{}, nb - 2, pslot);
        // ---- lane (r, kq) holds out[m][16 ot + 4 kq .. + 3] of its tiles.  b2 and the residual are added LAST, to the finished
        // sum of products (starting the accumulators from them would round every product at the residual's magnitude); the
        // loads of a tile go out together
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const long m = (pt + NW * rt + wave) * 16 + r;
            if (m < M) {
                long mo = m * C + 4 * kq;
                asm volatile("" : "+v"(mo));       // worked out here, not carried through the block loop
                if (nfs == 1) {
                    f32x4 bq[OT];
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) bq[ot] = *reinterpret_cast<const f32x4*>(b2 + 16 * ot + 4 * kq);
                    if (res) {
                        f32x4 rr[OT];
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) rr[ot] = *reinterpret_cast<const f32x4*>(res + mo + 16 * ot);
                        if (HAS_LN && res_ln) {     // the shortcut is LN(x), not x: the prologue's formula on the output layout
#pragma unroll
                            for (int ot = 0; ot < OT; ++ot) {
                                const f32x4 g1 = *reinterpret_cast<const f32x4*>(gs + 16 * ot + 4 * kq);
                                const f32x4 e1 = *reinterpret_cast<const f32x4*>(bs + 16 * ot + 4 * kq);
#pragma unroll
                                for (int i = 0; i < 4; ++i) rr[ot][i] = fmaf((rr[ot][i] - ln_mean[rt]) * ln_rstd[rt], g1[i], e1[i]);
                            }
                        }
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) acc2[rt][ot] = (acc2[rt][ot] + bq[ot]) + rr[ot];
                    } else {
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) acc2[rt][ot] += bq[ot];
                    }
                }
                if (nfs == 1 && gamma2) {
                    // LayerNorm behind the block (the encoder's norm2, or norm1 of the next Video-Swin block with the sum kept as
                    // the shortcut: out_sum): the row lives in lanes (r, kq = 0..3), two-pass
                    if (out_sum) {
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<f32x4*>(out_sum + mo + 16 * ot) = acc2[rt][ot];
                    }
                    const float inv_c = in_vgpr(1.0f / C), eps_v = in_vgpr(eps2);
                    float sm = 0.f;
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) sm += (acc2[rt][ot][0] + acc2[rt][ot][1]) + (acc2[rt][ot][2] + acc2[rt][ot][3]);
                    sm += __shfl_xor(sm, 16);
                    sm += __shfl_xor(sm, 32);
                    const float mean = sm * inv_c;
                    float q = 0.f;
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { acc2[rt][ot][i] -= mean; q = fmaf(acc2[rt][ot][i], acc2[rt][ot][i], q); }
                    q += __shfl_xor(q, 16);
                    q += __shfl_xor(q, 32);
                    const float rstd = rsqrtf(fmaf(q, inv_c, eps_v));
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        const f32x4 g2 = *reinterpret_cast<const f32x4*>(gamma2 + 16 * ot + 4 * kq);
                        const f32x4 e2 = *reinterpret_cast<const f32x4*>(beta2 + 16 * ot + 4 * kq);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc2[rt][ot][i] = fmaf(acc2[rt][ot][i] * rstd, g2[i], e2[i]);
                    }
                }
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<f32x4*>(o + mo + 16 * ot) = acc2[rt][ot];
            }
        }
    };
    for (long pt = t0; pt < t1; pt += NW * RT) {
        if (pt != t0) handoff<0>();                 // nobody still reads the slots the next pass's prologue refills
        // this wave's row tiles: pt + NW rt + wave (a part-filled pass leaves the later tiles of a wave empty first)
        int nrt = 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            if (pt + NW * rt + wave < t1) nrt = rt + 1;
        auto run = [&](auto late_c) {
            if (RT >= 4 && nrt == 4) pass(std::integral_constant<int, RT >= 4 ? 4 : 0>{}, late_c, pt);
            else if (RT >= 3 && nrt == 3) pass(std::integral_constant<int, RT >= 3 ? 3 : 0>{}, late_c, pt);
            else if (RT >= 2 && nrt == 2) pass(std::integral_constant<int, RT >= 2 ? 2 : 0>{}, late_c, pt);
            else if (nrt == 1) pass(std::integral_constant<int, 1>{}, late_c, pt);
            else pass(std::integral_constant<int, 0>{}, late_c, pt);
        };
        if ((STAG & 1) && wave >= NW / 2) run(std::integral_constant<bool, (STAG & 1) != 0>{});
        else run(std::false_type{});
    }
    __syncthreads();        // the waves retire together
}

// out = [LayerNorm](sum over the hidden ranges (fixed order) + b2 (+ residual)); a wave per row, up to two float4 per lane
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const float* __restrict__ part, int nfs, const float* __restrict__ b2,
                                                         const float* __restrict__ res, const float* __restrict__ gamma1,
                                                         const float* __restrict__ beta1, float eps1,
                                                         const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                         float eps2, float* __restrict__ out, float* __restrict__ out_sum,
                                                         long M, int C) {
    const int lane = threadIdx.x & 63;
    const long n4 = M * C / 4;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) {
        float4 a[2], rr[2];
        bool on[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c4 = lane + 64 * h;
            on[h] = c4 < C / 4;
            a[h] = rr[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on[h]) {
                const long i = m * (C / 4) + c4;
                a[h] = reinterpret_cast<const float4*>(part)[i];
                for (int f = 1; f < nfs; ++f) {
                    const float4 p = reinterpret_cast<const float4*>(part)[(long)f * n4 + i];
                    a[h].x += p.x; a[h].y += p.y; a[h].z += p.z; a[h].w += p.w;
                }
                const float4 bq = *reinterpret_cast<const float4*>(b2 + 4 * c4);
                a[h].x += bq.x; a[h].y += bq.y; a[h].z += bq.z; a[h].w += bq.w;
                if (res) rr[h] = reinterpret_cast<const float4*>(res)[i];
            }
        }
        if (res && gamma1) {       // the shortcut is LN(res) (two-pass, as the block kernel's prologue)
            float sm = ((rr[0].x + rr[0].y) + (rr[0].z + rr[0].w)) + ((rr[1].x + rr[1].y) + (rr[1].z + rr[1].w));
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) sm += __shfl_xor(sm, d);
            const float mean = sm / C;
            float q = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (on[h]) {
                    rr[h].x -= mean; rr[h].y -= mean; rr[h].z -= mean; rr[h].w -= mean;
                    q += rr[h].x * rr[h].x + rr[h].y * rr[h].y + rr[h].z * rr[h].z + rr[h].w * rr[h].w;
                }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) q += __shfl_xor(q, d);
            const float rstd = rsqrtf(q / C + eps1);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (on[h]) {
                    const float4 g1 = *reinterpret_cast<const float4*>(gamma1 + 4 * (lane + 64 * h));
                    const float4 e1 = *reinterpret_cast<const float4*>(beta1 + 4 * (lane + 64 * h));
                    rr[h].x = fmaf(rr[h].x * rstd, g1.x, e1.x); rr[h].y = fmaf(rr[h].y * rstd, g1.y, e1.y);
                    rr[h].z = fmaf(rr[h].z * rstd, g1.z, e1.z); rr[h].w = fmaf(rr[h].w * rstd, g1.w, e1.w);
                }
        }
        if (res) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { a[h].x += rr[h].x; a[h].y += rr[h].y; a[h].z += rr[h].z; a[h].w += rr[h].w; }
        }
        if (gamma2) {
            if (out_sum) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (on[h]) reinterpret_cast<float4*>(out_sum)[m * (C / 4) + lane + 64 * h] = a[h];
            }
            float sm = ((a[0].x + a[0].y) + (a[0].z + a[0].w)) + ((a[1].x + a[1].y) + (a[1].z + a[1].w));
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) sm += __shfl_xor(sm, d);
            const float mean = sm / C;
            float q = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (on[h]) {
                    a[h].x -= mean; a[h].y -= mean; a[h].z -= mean; a[h].w -= mean;
                    q += a[h].x * a[h].x + a[h].y * a[h].y + a[h].z * a[h].z + a[h].w * a[h].w;
                }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) q += __shfl_xor(q, d);
            const float rstd = rsqrtf(q / C + eps2);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (on[h]) {
                    const float4 g2 = *reinterpret_cast<const float4*>(gamma2 + 4 * (lane + 64 * h));
                    const float4 e2 = *reinterpret_cast<const float4*>(beta2 + 4 * (lane + 64 * h));
                    a[h].x = fmaf(a[h].x * rstd, g2.x, e2.x); a[h].y = fmaf(a[h].y * rstd, g2.y, e2.y);
                    a[h].z = fmaf(a[h].z * rstd, g2.z, e2.z); a[h].w = fmaf(a[h].w * rstd, g2.w, e2.w);
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (on[h]) reinterpret_cast<float4*>(out)[m * (C / 4) + lane + 64 * h] = a[h];
    }
}

// item = (chunk hc, block kind, group, lane): one 16-B piece per plane = 8 weights split three ways
template <int C>
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                       u32x4* __restrict__ img, int F) {
    using G = Geo<C>;
    constexpr int KS = G::KS, OT = G::OT;
    const long total = (long)(F / 32) * 2 * OT * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long rest = idx >> 6;
        const int sub = (int)(rest % OT);                              // W1: (j, s) = (sub / KS, sub % KS); W2: output tile
        rest /= OT;
        const int kind = (int)(rest & 1), hc = (int)(rest >> 1);
        const int n = lane & 15, kq = lane >> 4;
        float v[8];
        if (kind == 0) {
            const int j = sub / KS, s = sub % KS;
            const float* src = w1 + (long)(32 * hc + 16 * j + n) * C + 32 * s + 8 * kq;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[i];
        } else {
            // k order of a lane = the order the two hidden tiles of a chunk sit in the accumulators: 4 kq + i, 16 + 4 kq + i
            const float* src = w2 + (long)(16 * sub + n) * F + 32 * hc + 4 * kq;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = src[i]; v[4 + i] = src[16 + i]; }
        }
        bf16x8 h0, h1, h2;
        split8(v, h0, h1, h2);
        u32x4* dst = img + (long)(hc * 2 + kind) * G::BLKP_U4 + (sub * 3) * 64 + lane;
        dst[0] = __builtin_bit_cast(u32x4, h0);
        dst[64] = __builtin_bit_cast(u32x4, h1);
        dst[128] = __builtin_bit_cast(u32x4, h2);
    }
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)

// The shipped form of a width: waves per workgroup and row tiles per wave.  256-register waves (NW = 8) hold the x fragments
// (3 C / 8 registers per tile) and the output accumulators (C / 4 per tile) of RT tiles up to C = 256; C = 384 needs the 512
// registers of a one-wave-per-SIMD workgroup.
constexpr int form_nw(int C) { return C > 256 ? 4 : 8; }
constexpr int form_rt(int C) { return C <= 96 ? 2 : 1; }
constexpr int tiles_per_pass(int C) { return form_nw(C) * form_rt(C); }

template <int C>
size_t lds_bytes(int nslot, int sb, int F, int nfs) {
    const int nc = (F / 32 + nfs - 1) / nfs;
    return (size_t)nslot * (Geo<C>::BLKP_U4 / sb) * 16 + 8 * C + (size_t)nc * 128;
}

struct Args {
    const float *x, *b1, *b2, *gamma, *beta, *res, *gamma2, *beta2;
    float eps2;
    const u32x4* img;
    float eps;
    float* out;
    float* out_sum;
    long M;
    int F, nrg, nfs, res_ln;
    hipStream_t st;
};

template <int C, int ACT, bool HAS_LN, int NW, int RT, int NSLOT, int SB, int STAG, int PF, int DBG>
int launch(const Args& a) {
    const void* fn = reinterpret_cast<const void*>(mlp_split_kernel<C, ACT, HAS_LN, NW, RT, NSLOT, SB, STAG, PF, DBG>);
    const size_t lds = lds_bytes<C>(NSLOT, SB, a.F, a.nfs);
    if (lds > 160 * 1024) return SOC_EUNSUPPORTED;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((mlp_split_kernel<C, ACT, HAS_LN, NW, RT, NSLOT, SB, STAG, PF, DBG>), dim3((unsigned)(a.nrg * a.nfs)),
                       dim3(NW * 64), lds, a.st, a.x, a.img, a.b1, a.b2, a.gamma, a.beta, a.eps, a.res, a.gamma2, a.beta2, a.eps2,
                       a.out, a.out_sum, a.M, a.F, a.nrg, a.nfs, a.res_ln);
    return soc_check_launch();
}

// The ring of a width: a block of C <= 256 is one piece, three or four slots; the 72-KB blocks of C = 384 travel as halves (36 KB)
// through four slots -- two whole-block slots would leave one block in flight, too little to cover the L2 latency.
constexpr int form_sb(int C) { return C > 256 ? 2 : 1; }
template <int C>
constexpr int form_ns() {
    constexpr int ns = (160 * 1024 - 8 * C - 4096) / (Geo<C>::BLKP_U4 / form_sb(C) * 16);     // slots beside a 1024-wide b1 range
    return ns >= 4 ? 4 : ns;
}

// variant = NSLOT + 8 * log2(SB) + 32 * DBG + 512 * (PF - 1) + 1024 * STAG; 0 = the shipped choice for the width
template <int C, int ACT, bool HAS_LN>
int launch_variant(const Args& a, int variant) {
    static_assert(form_ns<C>() >= 3, "three slots at least");
    if (variant == 0) return launch<C, ACT, HAS_LN, form_nw(C), form_rt(C), form_ns<C>(), form_sb(C), 0, 1, 0>(a);
#ifdef SOC_K23_VARIANTS         // diagnostic build (tools/experiments/k23_time.py)
#define V(NS, LSB, ST, PFD, DB)                                                                                  \
    case NS + 8 * LSB + 32 * DB + 512 * (PFD - 1) + 1024 * ST:                                                   \
        if constexpr ((size_t)NS * (Geo<C>::BLKP_U4 >> LSB) * 16 + 8 * C + 4096 <= 160 * 1024 &&                 \
                      (Geo<C>::OT >> LSB) >= 1 && (LSB == 0 || Geo<C>::BLKP_U4 == Geo<C>::BLK_U4) &&             \
                      (Geo<C>::BLKP_U4 >> LSB) % (NW0 * 64) == 0 && ((ST & 3) == 0 || NW0 == 8))                       \
            return launch<C, ACT, HAS_LN, NW0, RT0, NS, (1 << LSB), ST, PFD, DB>(a);                             \
        else break;
    constexpr int NW0 = form_nw(C), RT0 = form_rt(C);
    switch (variant) {
        V(2, 0, 0, 1, 0) V(3, 0, 0, 1, 0) V(4, 1, 0, 1, 0) V(3, 1, 0, 1, 0)
        V(3, 0, 1, 1, 0) V(3, 0, 3, 1, 0) V(3, 0, 2, 1, 0) V(3, 0, 4, 1, 0) V(4, 0, 4, 1, 0) V(4, 1, 4, 1, 0) V(2, 0, 4, 1, 0)
        V(3, 0, 0, 1, 1) V(3, 0, 0, 1, 2) V(3, 0, 0, 1, 13) V(4, 1, 0, 1, 1) V(4, 1, 0, 1, 2) V(4, 1, 0, 1, 13)
        V(3, 1, 0, 1, 1) V(3, 1, 0, 1, 2) V(3, 1, 0, 1, 13) V(3, 1, 0, 1, 4) V(3, 1, 0, 1, 8) V(6, 2, 0, 1, 0) V(5, 2, 0, 1, 0) V(4, 2, 0, 1, 0)
        default: break;
    }
#undef V
#endif
    return SOC_EUNSUPPORTED;
}

template <int C>
int launch_c(const Args& a, int act, int variant) {
#ifdef SOC_K23_VARIANTS         // the diagnostic build instantiates the forms the model uses only
    if (act == 1 && !a.gamma) return launch_variant<C, 1, false>(a, variant);
    if (act == 2 && a.gamma) return launch_variant<C, 2, true>(a, variant);
#else
    if (act == 1) return a.gamma ? launch_variant<C, 1, true>(a, variant) : launch_variant<C, 1, false>(a, variant);
    if (act == 2) return a.gamma ? launch_variant<C, 2, true>(a, variant) : launch_variant<C, 2, false>(a, variant);
#endif
    return SOC_EUNSUPPORTED;
}

bool width_ok(int C) { return C == 96 || C == 128 || C == 192 || C == 256 || C == 384 || C == 512; }

template <int C> size_t packed_bytes(int F) { return (size_t)(F / 32) * 2 * Geo<C>::BLKP_U4 * 16; }

}  // namespace

extern "C" size_t soc_mlp_split_packed_bytes(int C, int F) {
    if (!width_ok(C) || F <= 0 || F % 32 != 0) return 0;
    switch (C) {
        case 96: return packed_bytes<96>(F);
        case 128: return packed_bytes<128>(F);
        case 192: return packed_bytes<192>(F);
        case 256: return packed_bytes<256>(F);
        case 384: return packed_bytes<384>(F);
        default: return packed_bytes<512>(F);
    }
}

extern "C" int soc_mlp_split_pack_f32(const float* w1, const float* w2, void* packed, int C, int F, void* stream) {
    if (!w1 || !w2 || !packed) return SOC_EINVAL;
    if (!width_ok(C) || F <= 0 || F % 32 != 0) return SOC_EUNSUPPORTED;
    const long total = (long)(F / 32) * 2 * (C / 16) * 64;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    u32x4* img = reinterpret_cast<u32x4*>(packed);
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 96: hipLaunchKernelGGL(mlp_pack_kernel<96>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
        case 128: hipLaunchKernelGGL(mlp_pack_kernel<128>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
        case 192: hipLaunchKernelGGL(mlp_pack_kernel<192>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
        case 256: hipLaunchKernelGGL(mlp_pack_kernel<256>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
        case 384: hipLaunchKernelGGL(mlp_pack_kernel<384>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
        default: hipLaunchKernelGGL(mlp_pack_kernel<512>, dim3(blocks), dim3(256), 0, st, w1, w2, img, F); break;
    }
    return soc_check_launch();
}

// How a launch is cut: `nrg` workgroup rows over the row tiles x `nfs` hidden ranges.  Many rows: one hidden range, a
// workgroup per CU.  Few rows (less than half a pass of every CU): the hidden dimension is split so that nrg * nfs fills the
// chip; nfs is a power of two <= 8 (an XCD then streams one range only).
static void plan_rows(long M, int C, int F, int cus, int* nrg_out, int* nfs_out) {
    const long ntiles = (M + 15) >> 4;
    const int per_pass = tiles_per_pass(C);
    const long groups = (ntiles + per_pass - 1) / per_pass;             // workgroup passes if every pass were full
    int nfs = 1;
    while (nfs < 8 && groups * nfs * 2 <= cus && (F / 32) % (nfs * 2) == 0) nfs *= 2;
    long nrg = groups < cus / nfs ? groups : cus / nfs;
    if (nrg < 1) nrg = 1;
    *nrg_out = (int)nrg;
    *nfs_out = nfs;
}

// A whole call: rows that fill whole rounds of the chip (every CU one full pass) run with one hidden range; a last round that
// would be less than 60 % filled is cut off and run as its own launch over split hidden ranges (a round streams both
// weight matrices into every CU whatever its fill).  Returns the row where the tail starts (M: no tail).
static long tail_start(long M, int C, int cus) {
    const long ntiles = (M + 15) >> 4;
    const long per_round = (long)cus * tiles_per_pass(C);
    const long full = ntiles / per_round, rem = ntiles - full * per_round;
    if (full >= 1 && rem > 0 && rem * 5 < per_round * 3) return full * per_round * 16;
    return M;
}

// Largest hidden width the one-range form (nfs == 1, what whole rounds run) holds in 160 KB of LDS beside its weight ring:
// ring + gamma / beta + 4 bytes of b1 per hidden unit (lds_bytes above).
template <int C>
static int max_hidden() {
    const size_t fixed = (size_t)form_ns<C>() * (Geo<C>::BLKP_U4 / form_sb(C)) * 16 + 8 * C;
    return (int)((160 * 1024 - fixed) / 128) * 32;
}

extern "C" int soc_mlp_split_max_hidden(int C) {
    switch (C) {
        case 96: return max_hidden<96>();
        case 128: return max_hidden<128>();
        case 192: return max_hidden<192>();
        case 256: return max_hidden<256>();
        case 384: return max_hidden<384>();
        case 512: return max_hidden<512>();
        default: return 0;
    }
}

extern "C" int soc_mlp_split_plan(long M, int C, int F, int* nrg_out, int* nfs_out, void* stream) {
    if (!width_ok(C) || F <= 0 || F % 32 != 0 || M <= 0 || !nrg_out || !nfs_out) return SOC_EUNSUPPORTED;
    plan_rows(M, C, F, num_cus((hipStream_t)stream), nrg_out, nfs_out);
    return SOC_OK;
}

extern "C" size_t soc_mlp_split_workspace_bytes(long M, int C, int F, void* stream) {
    if (!width_ok(C) || F <= 0 || F % 32 != 0 || M <= 0) return 0;
    const int cus = num_cus((hipStream_t)stream);
    const long m0 = tail_start(M, C, cus);
    int nrg = 0, nfs = 0;
    plan_rows(m0 < M ? M - m0 : M, C, F, cus, &nrg, &nfs);
    return nfs == 1 ? 0 : (size_t)nfs * (m0 < M ? M - m0 : M) * C * sizeof(float);
}

extern "C" int soc_mlp_split_variant_f32(const float* x, const void* packed, const float* b1, const float* b2,
                                         const float* ln_gamma, const float* ln_beta, float ln_eps, const float* residual,
                                         const float* post_gamma, const float* post_beta, float post_eps, float* out,
                                         float* out_sum, float* workspace, long M, int C, int F, int act, int residual_ln,
                                         int nrg, int nfs, int variant, void* stream) {
    if (M < 0 || F <= 0) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (!x || !packed || !b1 || !b2 || !out || (ln_gamma == nullptr) != (ln_beta == nullptr) ||
        (post_gamma == nullptr) != (post_beta == nullptr) || (out_sum && !post_gamma) ||
        (residual_ln && (!residual || !ln_gamma)))
        return SOC_EINVAL;
    // residual_ln: the shortcut is LN(x) -- the block kernel takes the row statistics from x, the reduce kernel from the
    // residual rows; the two agree only for residual == x, so anything else is refused rather than plan-dependent
    if (residual_ln && residual != x) return SOC_EINVAL;
    if (!width_ok(C) || F % 32 != 0 || (act != 1 && act != 2)) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)residual | (uintptr_t)out |
          (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)post_gamma | (uintptr_t)post_beta | (uintptr_t)workspace |
          (uintptr_t)out_sum) & 15) != 0)
        return SOC_EUNSUPPORTED;
    if (nrg <= 0 || nfs <= 0 || nfs > F / 32 || nrg > (M + 15) / 16) return SOC_EINVAL;
    if (nfs > 1 && !workspace) return SOC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Args a{x, b1, b2, ln_gamma, ln_beta, nfs > 1 ? nullptr : residual, nfs > 1 ? nullptr : post_gamma,
           nfs > 1 ? nullptr : post_beta, post_eps, reinterpret_cast<const u32x4*>(packed), ln_eps,
           nfs > 1 ? workspace : out, nfs > 1 ? nullptr : out_sum, M, F, nrg, nfs, residual_ln, st};
    int rc;
    switch (C) {
        case 96: rc = launch_c<96>(a, act, variant); break;
        case 128: rc = launch_c<128>(a, act, variant); break;
        case 192: rc = launch_c<192>(a, act, variant); break;
        case 256: rc = launch_c<256>(a, act, variant); break;
        case 384: rc = launch_c<384>(a, act, variant); break;
        default: rc = launch_c<512>(a, act, variant); break;
    }
    if (rc != SOC_OK || nfs == 1) return rc;
    const int blocks = (int)((M + 3) / 4 > 8192 ? 8192 : (M + 3) / 4);
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, nfs, b2, residual,
                       residual_ln ? ln_gamma : nullptr, residual_ln ? ln_beta : nullptr, ln_eps, post_gamma, post_beta, post_eps,
                       out, out_sum, M, C);
    return soc_check_launch();
}

extern "C" int soc_mlp_split_f32(const float* x, const void* packed, const float* b1, const float* b2, const float* ln_gamma,
                                 const float* ln_beta, float ln_eps, const float* residual, const float* post_gamma,
                                 const float* post_beta, float post_eps, float* out, float* out_sum, float* workspace,
                                 size_t workspace_bytes, long M, int C, int F, int act, int residual_ln, void* stream) {
    if (M < 0 || F <= 0) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (!width_ok(C) || F % 32 != 0) return SOC_EUNSUPPORTED;
    if (workspace_bytes < soc_mlp_split_workspace_bytes(M, C, F, stream)) return SOC_EWORKSPACE;
    const int cus = num_cus((hipStream_t)stream);
    const long m0 = tail_start(M, C, cus);
    int nrg = 0, nfs = 0;
    if (m0 < M) {       // whole rounds first, then the tail over split hidden ranges
        const int rc = soc_mlp_split_variant_f32(x, packed, b1, b2, ln_gamma, ln_beta, ln_eps, residual, post_gamma, post_beta,
                                                 post_eps, out, out_sum, nullptr, m0, C, F, act, residual_ln, cus, 1, 0, stream);
        if (rc != SOC_OK) return rc;
        plan_rows(M - m0, C, F, cus, &nrg, &nfs);
        return soc_mlp_split_variant_f32(x + m0 * C, packed, b1, b2, ln_gamma, ln_beta, ln_eps,
                                         residual ? residual + m0 * C : nullptr, post_gamma, post_beta, post_eps, out + m0 * C,
                                         out_sum ? out_sum + m0 * C : nullptr, workspace, M - m0, C, F, act, residual_ln, nrg, nfs, 0,
                                         stream);
    }
    plan_rows(M, C, F, cus, &nrg, &nfs);
    return soc_mlp_split_variant_f32(x, packed, b1, b2, ln_gamma, ln_beta, ln_eps, residual, post_gamma, post_beta, post_eps, out,
                                     out_sum, workspace, M, C, F, act, residual_ln, nrg, nfs, 0, stream);
}
