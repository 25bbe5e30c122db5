// K24 kernel templates shared by the two translation units that instantiate them (xs_linear_split.hip: K <= 384,
// xs_linear_split_wide.hip: K = 512 / 768 / 1024 -- one file took three minutes to compile).  See xs_linear_split.hip for
// the data flow.  Everything here has internal linkage; the wide unit exports three plain functions (namespace soc_xs).
#pragma once
#include "soc_common.h"
#include "split_math.h"
#include <atomic>
#include <type_traits>

namespace soc_xs {     // external linkage: the argument block crosses the two translation units
struct Args {
    const float *x, *bias, *gamma, *beta, *res;
    const soc_split::u32x4* img;
    float eps;
    float* out;
    long M;
    int N, nrg, ncr, rpw;       // ncr column spans of rpw ranges each
    hipStream_t st;
};
}  // namespace soc_xs

namespace {

using namespace soc_split;
constexpr int MAX_THREADS = 512;

template <int K>
struct Geo {
    static constexpr int KS = K / 32;                                   // k-steps of 32
    static constexpr int NW = K > 384 ? 4 : 8;                          // waves per workgroup
    static constexpr int SB = K > 768 ? 4 : (K > 384 ? 2 : 1);          // ring pieces per column tile (K split)
    static constexpr int CTP = K <= 256 ? 2 : 1;                        // column tiles per ring piece
    static constexpr int GROUPS = CTP * KS / SB;                        // fragment groups (16 columns x 32 k, three planes) per piece
    static constexpr int PIECE_U4 = (GROUPS * 3 * 64 + MAX_THREADS - 1) / MAX_THREADS * MAX_THREADS;   // whole DMA rounds
    static constexpr int BIAS_BYTES = 8192;                             // the bias of a workgroup's column span: <= 2048 columns
    static constexpr int NSLOT = (160 * 1024 - 8 * K - BIAS_BYTES) / (PIECE_U4 * 16) >= 4 ? 4 : 3;
    static_assert(KS % SB == 0 && (160 * 1024 - 8 * K - BIAS_BYTES) / (PIECE_U4 * 16) >= 3, "three ring slots at least");
};

template <int N_>
__device__ __forceinline__ void handoff() {        // my part of the next piece has landed; everyone is done with this one
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N_) : "memory");
}

#ifdef SOC_K24_STAMPS       // diagnostic build only (tools/experiments/k24_stamps.py): s_memtime at phase boundaries, [block][wave][8]
__device__ unsigned long long* g_xs_dbg = nullptr;
#define XS_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        if (g_xs_dbg && (threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                          \
            g_xs_dbg[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define XS_STAMP(slot) do {} while (0)
#endif

// NCT column tiles per range (compile time: the accumulators are registers).  A workgroup owns a SPAN of `rpw` consecutive
// ranges: its waves keep their split rows for the whole span and walk the ranges one after the other (round 5: with one range
// per workgroup the row prologue -- loads, LayerNorm, split, ring fill, about ten column tiles' worth of time -- was paid per
// 18 tiles at most; at the row counts of a launch group the chip fills without cutting N that finely).
template <int K, int ACT, bool HAS_LN, int NCT>
__global__ __launch_bounds__(Geo<K>::NW * 64, Geo<K>::NW / 4) void xs_linear_kernel(
    const float* __restrict__ x, const u32x4* __restrict__ img, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, const float* __restrict__ res,
    float* __restrict__ out, long M, int N, int nrg, int ncr, int xcd_rows, int rpw) {
    using G = Geo<K>;
    constexpr int NW = G::NW, THREADS = NW * 64, KS = G::KS, SB = G::SB, CTP = G::CTP, NSLOT = G::NSLOT;
    constexpr int SLOT = G::PIECE_U4, P = SLOT / THREADS, D = NSLOT - 1, NQ = NCT * SB / CTP, KSP = KS / SB;
    static_assert(NCT % CTP == 0 && (D - 1) * P < 64, "whole pieces");
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    u32x4* slots = lds;
    float* gs = reinterpret_cast<float*>(lds + NSLOT * SLOT);           // gamma [K], beta [K], this range's bias [16 NCT]
    float* bs = gs + K;
    float* bias_s = bs + K;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                       // own the CU
    if (NW == 4) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    // Workgroup -> (row group g, column range cr).  Workgroups are dealt to the 8 XCDs round-robin (b, b + 8, ... share an XCD
    // and its private 4 MB L2).  Default: consecutive workgroups are the column ranges of one row group, so an XCD streams
    // 1 / gcd(ncr, 8) of the weight image and sees gcd(ncr, 8) / 8 of the rows -- the rows of x are fetched from HBM once per
    // column range.  xcd_rows (chosen by the launcher when the whole weight image fits an L2 beside the rows): the ncr
    // workgroups of a row group sit on ONE XCD, x comes from HBM once and from that L2 for the other ranges, every XCD streams
    // the whole image (stage-2 qkv of Video-Swin-T: 50 -> 33 MB of HBM traffic per launch).
    int cr, g;
    if (xcd_rows) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cr = j % ncr;
        g = (j / ncr) * 8 + xcd;
        if (g >= nrg) return;           // the grid is padded to whole groups of 8 row groups: a surplus workgroup has no rows
    } else {
        cr = blockIdx.x % ncr;
        g = blockIdx.x / ncr;
    }
    const int n00 = cr * rpw * NCT * 16;                                // first column of this workgroup's span (ncr spans)
    const long ntiles = (M + 15) >> 4;
    const long t0 = (long)g * ntiles / nrg, t1 = (long)(g + 1) * ntiles / nrg;
    if (HAS_LN)
        for (int i = tid; i < K; i += THREADS) { gs[i] = gamma[i]; bs[i] = beta[i]; }
    // the bias of the column range waits in LDS: read from global memory in the epilogue it was a dependent load in front of
    // the stores of every wave (tools/experiments/k24_stamps.py)
    for (int i = tid; i < rpw * NCT * 16; i += THREADS) bias_s[i] = bias ? bias[n00 + i] : 0.f;
    // LDS-DMA from inline assembly, as K23 (mlp_split.hip): the compiler must not see an LDS-DMA in flight
    const char* const ibase0 = reinterpret_cast<const char*>(img + (long)cr * rpw * NQ * SLOT);
    const char* ibase = ibase0;                                         // the current range's pieces
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
    __syncthreads();
    auto pass = [&](auto act_c, long pt) {
        constexpr bool ACTIVE = decltype(act_c)::value;                 // this wave has a row tile in the pass
        bf16x8 xb[KS][3];
        XS_STAMP(0);
        ibase = ibase0;
        {
            float4 xn[KS][2];
            const long m = min((pt + wave) * 16 + r, M - 1);
            const float4* xp = reinterpret_cast<const float4*>(x + m * K + 8 * kq);
            if (ACTIVE) {
#pragma unroll
                for (int s = 0; s < KS; ++s) { xn[s][0] = xp[8 * s]; xn[s][1] = xp[8 * s + 1]; }
            }
#pragma unroll
            for (int b = 0; b < D; ++b)
                if (b < NQ) dma(b, b);
            XS_STAMP(1);
            if (ACTIVE) {
                float v[KS][8];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    v[s][0] = xn[s][0].x; v[s][1] = xn[s][0].y; v[s][2] = xn[s][0].z; v[s][3] = xn[s][0].w;
                    v[s][4] = xn[s][1].x; v[s][5] = xn[s][1].y; v[s][6] = xn[s][1].z; v[s][7] = xn[s][1].w;
                }
                if (HAS_LN) {   // as nn.LayerNorm: two-pass mean / variance over the row, which lives in lanes (r, kq = 0..3)
                    // every rounding below is written out: left to contract on its own, the compiler rounded 0.07 % of the
                    // normalised values one ulp apart between two instantiations (tools/experiments/k24_dbg.py), and a cut
                    // must not change the bits
#pragma clang fp contract(off)
                    const float inv_k = in_vgpr(1.0f / K), eps_v = in_vgpr(eps);
                    float sm = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int i = 0; i < 8; ++i) sm += v[s][i];
                    sm += __shfl_xor(sm, 16);
                    sm += __shfl_xor(sm, 32);
                    const float mean = sm * inv_k;
                    float q = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int i = 0; i < 8; ++i) { v[s][i] -= mean; q = fmaf(v[s][i], v[s][i], q); }
                    q += __shfl_xor(q, 16);
                    q += __shfl_xor(q, 32);
                    const float rstd = rsqrtf(fmaf(q, inv_k, eps_v));
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
                for (int s = 0; s < KS; ++s) {
                    split8(v[s], xb[s][0], xb[s][1], xb[s][2]);
                    // four 512-register waves: half of the file is accumulation registers, which the matrix cores read as
                    // well as the others.  The first fragments go there as they are made (the row's f32 values still hold 192
                    // of the 256 others at that point); left to itself the allocator spilled six fragments here and re-loaded
                    // them at the top of every range of the span.
#ifndef SOC_K24_NO_PIN          // diagnostic build: tools/experiments/k24_dbg.py
                    if (NW == 4 && s < (256 - 4 * NCT - 4) / 12)
                        asm volatile("" : "+a"(xb[s][0]), "+a"(xb[s][1]), "+a"(xb[s][2]));
#endif
                }
            }
        }
#pragma nounroll
        for (int rg = 0; rg < rpw; ++rg) {
        f32x4 acc[NCT];
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        XS_STAMP(2);
        // piece 0 has landed (a later range of the span: behind the previous range's stores, which share the counter)
        if (NQ >= D && rg == 0) handoff<(D - 1) * P>(); else handoff<0>();
        XS_STAMP(3);
        // ---- the ring: piece q = column tiles [CTP (q / SB), + CTP) x k-steps [KSP (q % SB), + KSP); fully unrolled (the
        // accumulators are registers), fragment groups read one ahead of the MFMAs that consume them
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            constexpr int dummy = 0; (void)dummy;
            if (q + D < NQ) dma(q + D, (q + D) % NSLOT);
            if (ACTIVE) {
                const u32x4* wl = slots + (q % NSLOT) * SLOT + lane;
                constexpr int NGP = CTP * KSP;                          // fragment groups of the piece, column tile major
                bf16x8 wf[2][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[0][pl] = __builtin_bit_cast(bf16x8, wl[pl * 64]);
#pragma unroll
                for (int gi = 0; gi < NGP; ++gi) {
                    if (gi + 1 < NGP) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[(gi + 1) & 1][pl] = __builtin_bit_cast(bf16x8, wl[((gi + 1) * 3 + pl) * 64]);
                    }
                    const int j = CTP * (q / SB) + gi / KSP, s = KSP * (q % SB) + gi % KSP;
                    mfma6(acc[j], wf[gi & 1], xb[s][0], xb[s][1], xb[s][2]);
                    if (gi + 1 < NGP) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (q + 1 < NQ) {
                if (q + D < NQ) handoff<(D - 1) * P>(); else handoff<0>();
            }
        }
        // ---- the next range of the span: its first pieces travel while this range's results leave
        if (rg + 1 < rpw) {
            handoff<0>();                           // every wave is done with the slots
            ibase += (long)NQ * (SLOT * 16);
#pragma unroll
            for (int b = 0; b < D; ++b)
                if (b < NQ) dma(b, b);
        }
        // ---- lane (r, kq) holds out[m][n0 + 16 j + 4 kq .. + 3]: bias, activation, residual, 16-B stores
        XS_STAMP(4);
        const long m = (pt + wave) * 16 + r;
        if (ACTIVE && m < M) {
            const int nr = rg * (NCT * 16);         // first column of the range within the span
            long mo = m * N + n00 + nr + 4 * kq;
            asm volatile("" : "+v"(mo));
            // The split rows stay in registers for the next range of the span, so the epilogue has what the fragments leave: at
            // K = 384 (144 of 256 registers) and K >= 768 (288 / 384 of 512) the range leaves in rounds of EC column tiles --
            // bias, activation, shortcut, stores of one round before the next (all 18 tiles at once: the compiler kept 72
            // registers each of bias, shortcut and accumulator copies and spilled 100 registers of fragments, re-loaded through
            // the vector-memory counter at the top of every range).
            constexpr int EC = K > 768 ? NCT / 2 : ((K == 384 || K == 768) && NCT > 12) ? (NCT % 3 == 0 ? NCT / 3 : NCT / 2) : NCT;
            const GeluK gk = gelu_k();
#pragma unroll
            for (int j0 = 0; j0 < NCT; j0 += EC) {
#pragma unroll
                for (int j = j0; j < j0 + EC; ++j) acc[j] += *reinterpret_cast<const f32x4*>(bias_s + nr + 16 * j + 4 * kq);
                if (ACT == 1) {
#pragma unroll
                    for (int j = j0; j < j0 + EC; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = fmaxf(acc[j][i], 0.f);
                } else if (ACT == 2) {
#pragma unroll
                    for (int j = j0; j < j0 + EC; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[j][i] = gelu_erf(acc[j][i], gk);
                }
                if (res) {
                    f32x4 rr[EC];
#pragma unroll
                    for (int j = 0; j < EC; ++j) rr[j] = *reinterpret_cast<const f32x4*>(res + mo + 16 * (j0 + j));
#pragma unroll
                    for (int j = 0; j < EC; ++j) acc[j0 + j] += rr[j];
                }
#pragma unroll
                for (int j = j0; j < j0 + EC; ++j) *reinterpret_cast<f32x4*>(out + mo + 16 * j) = acc[j];
                if (EC < NCT) __builtin_amdgcn_sched_barrier(0);
            }
        }
        }   // ranges of the span
        XS_STAMP(5);
    };
    for (long pt = t0; pt < t1; pt += NW) {
        if (pt != t0) handoff<0>();                 // nobody still reads the slots the next pass's prologue refills
        if (pt + wave < t1) pass(std::true_type{}, pt);
        else pass(std::false_type{}, pt);
    }
    __syncthreads();        // the waves retire together
    XS_STAMP(6);
}

// item = (piece, group, lane): one 16-B piece per plane = 8 weights split three ways.  Image: [N / 16 / CTP x SB pieces][PIECE_U4]
template <int K>
__global__ __launch_bounds__(256) void xs_pack_kernel(const float* __restrict__ w, u32x4* __restrict__ img, int N) {
    using G = Geo<K>;
    constexpr int KSP = G::KS / G::SB, NGP = G::GROUPS;
    const long npieces = (long)(N / 16 / G::CTP) * G::SB;
    const long total = npieces * NGP * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long rest = idx >> 6;
        const int gi = (int)(rest % NGP);
        const long q = rest / NGP;
        const int j = G::CTP * (int)(q / G::SB) + gi / KSP, s = KSP * (int)(q % G::SB) + gi % KSP;
        const int n = lane & 15, kq = lane >> 4;
        const float* src = w + (long)(16 * j + n) * K + 32 * s + 8 * kq;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[i];
        bf16x8 h0, h1, h2;
        split8(v, h0, h1, h2);
        u32x4* dst = img + q * G::PIECE_U4 + (gi * 3) * 64 + lane;
        dst[0] = __builtin_bit_cast(u32x4, h0);
        dst[64] = __builtin_bit_cast(u32x4, h1);
        dst[128] = __builtin_bit_cast(u32x4, h2);
    }
}

int num_cus(hipStream_t st) { return soc_num_cus(st); }      // CUs the launch stream may use (soc_capi.hip)

using soc_xs::Args;

// HBM traffic of the two workgroup -> XCD mappings (see the kernel): rows x gcd + 8 images / gcd against rows + 8 images.  The
// second one needs the whole split image (6 B per weight) in an XCD's 4 MB L2 beside its share of the rows (the grid is
// padded to whole groups of 8 row groups, one per XCD; surplus workgroups return at once).
inline int xcd_rows_pays(long M, int N, int K, int nrg, int ncr) {
#ifdef SOC_K24_NO_XCD_ROWS          // diagnostic build: the round-4 mapping everywhere (tools/experiments/k24_xcd_rows.py)
    return 0;
#endif
    if (ncr <= 1 || nrg < 8) return 0;
    const double x_bytes = 4.0 * (double)M * K, w_bytes = 6.0 * (double)N * K;
    if (w_bytes > 3.0 * 1024 * 1024) return 0;
    int g = 8;
    while (ncr % g != 0) g >>= 1;                       // gcd(ncr, 8)
    return x_bytes + 8.0 * w_bytes < x_bytes * g + 8.0 * w_bytes / g ? 1 : 0;
}

template <int K, int ACT, bool HAS_LN, int NCT>
int launch(const Args& a) {
    using G = Geo<K>;
    const void* fn = reinterpret_cast<const void*>(xs_linear_kernel<K, ACT, HAS_LN, NCT>);
    if ((size_t)a.rpw * NCT * 64 > (size_t)G::BIAS_BYTES) return SOC_EUNSUPPORTED;
    const size_t lds = (size_t)G::NSLOT * G::PIECE_U4 * 16 + 8 * K + (size_t)a.rpw * NCT * 64;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int xcd_rows = xcd_rows_pays(a.M, a.N, K, a.nrg, a.ncr);
    const int groups = xcd_rows ? (a.nrg + 7) / 8 * 8 : a.nrg;         // xcd_rows: whole groups of 8 row groups (one per XCD)
    hipLaunchKernelGGL((xs_linear_kernel<K, ACT, HAS_LN, NCT>), dim3((unsigned)(groups * a.ncr)), dim3(G::NW * 64), lds, a.st,
                       a.x, a.img, a.bias, a.gamma, a.beta, a.eps, a.res, a.out, a.M, a.N, a.nrg, a.ncr, xcd_rows, a.rpw);
    return soc_check_launch();
}

// column tiles per range the kernels are built for (the accumulators are registers: 4 per tile; K = 384 has room for 18,
// K = 768 in 512 registers likewise)
constexpr int NCTS[] = {18, 16, 12, 8, 6, 4};

// K = 1024 keeps 384 registers of x fragments: ranges of more than 8 column tiles would spill into scratch (checked on the
// assembly by tests/test_isa_rules.py), they are not built
constexpr int max_nct(int K) { return K > 768 ? 8 : 18; }

template <int K, int ACT, bool HAS_LN, int NCT>
int launch_if_built(const Args& a) {
    if constexpr (NCT <= max_nct(K)) return launch<K, ACT, HAS_LN, NCT>(a);
    else return SOC_EUNSUPPORTED;
}

template <int K, int ACT, bool HAS_LN>
int launch_nct(const Args& a, int nct) {
    switch (nct) {
        case 18: return launch_if_built<K, ACT, HAS_LN, 18>(a);
        case 16: return launch_if_built<K, ACT, HAS_LN, 16>(a);
        case 12: return launch_if_built<K, ACT, HAS_LN, 12>(a);
        case 8: return launch_if_built<K, ACT, HAS_LN, 8>(a);
        case 6: return launch_if_built<K, ACT, HAS_LN, 6>(a);
        case 4: return launch_if_built<K, ACT, HAS_LN, 4>(a);
        default: return SOC_EUNSUPPORTED;
    }
}

template <int K>
int launch_k(const Args& a, int act, int nct) {
    if (a.gamma) {
        if (act == 0) return launch_nct<K, 0, true>(a, nct);
        if (act == 2) return launch_nct<K, 2, true>(a, nct);
        return SOC_EUNSUPPORTED;
    }
    if (act == 0) return launch_nct<K, 0, false>(a, nct);
    if (act == 1) return launch_nct<K, 1, false>(a, nct);
    if (act == 2) return launch_nct<K, 2, false>(a, nct);
    return SOC_EUNSUPPORTED;
}

template <int K> size_t packed_bytes(int N) { return (size_t)(N / 16 / Geo<K>::CTP) * Geo<K>::SB * Geo<K>::PIECE_U4 * 16; }

}  // namespace

namespace soc_xs {     // defined in xs_linear_split_wide.hip
size_t packed_bytes_wide(int K, int N);
void pack_wide(int K, const float* w, void* packed, int N, int blocks, hipStream_t st);
int launch_wide(int K, const Args& a, int act, int nct);
}  // namespace soc_xs
