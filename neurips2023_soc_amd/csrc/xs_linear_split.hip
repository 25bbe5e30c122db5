// K24: an x-stationary linear layer on the bf16 matrix cores (exact three-way operand split) -- K23's first product
// (mlp_split.hip) as a kernel of its own:
//
//   out[M, N] = act(LN(x)[M, K] . W[N, K]^T + b) + residual                                          (f32 in, f32 out)
//
// Replaces the pixel-sized nn.Linear calls of widths K = 192 / 256 / 384 / 512 / 768 / 1024 that K13b (weights stationary in LDS, rows
// streamed and split again for every column range) ran at 0.3 of their ceiling: Video-Swin qkv / proj of stages 1-3 (reference
// models/video_swin_transformer.py:144-166, with norm1 :219 in front and the shortcut :254-259 behind), the patch-merging
// reduction into stage 2 (:277-312), the deformable encoder's value_proj / output_proj (models/ops/modules/ms_deform_attn.py:
// 95,114), the fusion blocks' query projection (models/vla.py:18-24) and input_proj of level 1 (models/soc.py:226-230).
//
// Data flow: a wave owns a 16-row tile of x and keeps it, normalised and split ONCE, as bf16 MFMA fragments; the weights of
// the workgroup's column range stream through the LDS ring of K23 (pre-split, laid out as the fragments are read:
// soc_xs_linear_pack_f32; LDS-DMA from inline assembly, counted vmcnt hand-offs, fragment reads one group ahead); out^T tiles
// [16 columns x 16 rows] accumulate in registers for the whole range (NCT tiles) and leave once, behind bias / activation /
// residual, as 16-B stores.  A launch is cut into `nrg` workgroup rows x `ncr` column ranges (no partial sums: a range owns
// its columns), so that short inputs still fill the chip; blocks b and b + 8 share an XCD, which then streams one range.
// K <= 384: eight 256-register waves; K = 512 / 768 / 1024 (xs_linear_split_wide.hip): four 512-register waves, a column tile
// travels as two (four at K = 1024) partial-K pieces.
// Co-residence rule as K23: whole register file, final barrier, VGPR operands in packed f32 code.
#include "xs_linear_split.h"
#include <cstdlib>

namespace {

bool width_ok(int K) { return K == 192 || K == 256 || K == 384 || K == 512 || K == 768 || K == 1024; }
int waves_of(int K) { return K > 384 ? 4 : 8; }
int ctp_of(int K) { return K <= 256 ? 2 : 1; }

// The cut of M rows x N columns: nrg workgroup rows (NW row tiles per pass each) x ncr column SPANS; a span is rpw ranges of
// one of the built widths, walked one after the other by waves that keep their split rows.  Cost model
// (tools/experiments/k24_time.py): a workgroup pass costs a fixed part (row loads, LayerNorm, split, ring fill: about ten
// column tiles' worth of MFMA time -- tools/experiments/k24_main_tail.py: 38 560 x 256 x 256 takes 41 us as (256, 1) and 48 us
// as (128, 2)) plus its column tiles, plus two tiles' worth per further range of the span (the ring drains and refills behind
// the stores); a launch costs the passes of its busiest workgroup.  Fewest passes x (tiles + fixed) wins, ties go to the
// wider range and the longer span (x is split once per span).
constexpr int SPAN_MAX_COLUMNS = 2048;          // Geo<K>::BIAS_BYTES / 4: the span's bias waits in LDS

// widest built range that divides a span of `tiles` column tiles (0: none)
int range_of(int K, int tiles) {
    for (int nct : NCTS)
        if (tiles % nct == 0 && nct % ctp_of(K) == 0 && nct <= max_nct(K)) return nct;
    return 0;
}

// diagnostic switch SOC_K24_SPANS=0: one range per workgroup, the cut of round 4 (read once)
bool spans_enabled() {
    static const bool on = [] { const char* e = std::getenv("SOC_K24_SPANS"); return !(e && e[0] == '0'); }();
    return on;
}

bool plan(long M, int N, int K, int cus, int* nrg_out, int* ncr_out, int* nct_out) {
    const long ntiles = (M + 15) >> 4;
    const int nw = waves_of(K), nct_all = N / 16;
    const long groups = (ntiles + nw - 1) / nw;
    int best_span = 0;
    long best_cost = 0, best_nrg = 0;
    for (int nct : NCTS) {
        if (nct_all % nct != 0 || nct % ctp_of(K) != 0 || nct > max_nct(K)) continue;
        const int ranges = nct_all / nct;
        for (int rpw = 1; rpw <= (spans_enabled() ? ranges : 1); ++rpw) {
            if (ranges % rpw != 0 || rpw * nct * 16 > SPAN_MAX_COLUMNS || (rpw > 1 && range_of(K, rpw * nct) != nct)) continue;
            const long ncr = ranges / rpw;
            long nrg = cus / ncr > 0 ? cus / ncr : 1;
            if (nrg > groups) nrg = groups;
            const long tiles_per_wg = (ntiles + nrg - 1) / nrg;
            const long passes = (tiles_per_wg + nw - 1) / nw;
            const long rounds = (nrg * ncr + cus - 1) / cus;
            const long cost = passes * rounds * (rpw * nct + 10 + 2 * (rpw - 1));
            if (best_span == 0 || cost < best_cost || (cost == best_cost && rpw * nct > best_span)) {
                best_span = rpw * nct; best_cost = cost; best_nrg = nrg;
            }
        }
    }
    if (best_span == 0) return false;
    *nrg_out = (int)best_nrg;
    *ncr_out = nct_all / best_span;
    *nct_out = best_span;
    return true;
}

}  // namespace

#ifdef SOC_K24_STAMPS
extern "C" int soc_xs_debug_set_buffer(void* ptr) {
    unsigned long long* p = (unsigned long long*)ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_xs_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t soc_xs_linear_packed_bytes(int N, int K) {
    if (!width_ok(K) || N <= 0 || N % 32 != 0) return 0;
    switch (K) {
        case 192: return packed_bytes<192>(N);
        case 256: return packed_bytes<256>(N);
        case 384: return packed_bytes<384>(N);
        default: return soc_xs::packed_bytes_wide(K, N);
    }
}

extern "C" int soc_xs_linear_pack_f32(const float* w, void* packed, int N, int K, void* stream) {
    if (!w || !packed) return SOC_EINVAL;
    if (!width_ok(K) || N <= 0 || N % 32 != 0) return SOC_EUNSUPPORTED;
    const long total = (long)N * K / 8;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    u32x4* img = reinterpret_cast<u32x4*>(packed);
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 192: hipLaunchKernelGGL(xs_pack_kernel<192>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
        case 256: hipLaunchKernelGGL(xs_pack_kernel<256>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
        case 384: hipLaunchKernelGGL(xs_pack_kernel<384>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
        default: soc_xs::pack_wide(K, w, packed, N, blocks, st); break;
    }
    return soc_check_launch();
}

extern "C" int soc_xs_linear_plan(long M, int N, int K, int* nrg, int* ncr, int* nct, void* stream) {
    if (!width_ok(K) || N <= 0 || N % 32 != 0 || M <= 0 || !nrg || !ncr || !nct) return SOC_EUNSUPPORTED;
    return plan(M, N, K, num_cus((hipStream_t)stream), nrg, ncr, nct) ? SOC_OK : SOC_EUNSUPPORTED;
}

extern "C" int soc_xs_linear_f32(const float* x, const void* packed, const float* bias, const float* ln_gamma,
                                 const float* ln_beta, float ln_eps, const float* residual, float* out, long M, int N, int K,
                                 int act, int nrg, int ncr, void* stream) {
    if (M < 0 || N <= 0) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (!x || !packed || !out || (ln_gamma == nullptr) != (ln_beta == nullptr)) return SOC_EINVAL;
    if (!width_ok(K) || N % 32 != 0 || act < 0 || act > 2) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)ln_gamma |
          (uintptr_t)ln_beta) & 15) != 0)
        return SOC_EUNSUPPORTED;
    int span = 0;               // column tiles per workgroup: rpw ranges of nct tiles
    if (nrg <= 0 || ncr <= 0) {
        if (!plan(M, N, K, num_cus((hipStream_t)stream), &nrg, &ncr, &span)) return SOC_EUNSUPPORTED;
    } else {
        if ((N / 16) % ncr != 0) return SOC_EINVAL;
        span = N / 16 / ncr;
    }
    if (nrg > (M + 15) / 16 || span % ctp_of(K) != 0) return SOC_EINVAL;
    const int nct = range_of(K, span);
    if (nct == 0 || span * 16 > SPAN_MAX_COLUMNS) return SOC_EUNSUPPORTED;
    Args a{x, bias, ln_gamma, ln_beta, residual, reinterpret_cast<const u32x4*>(packed), ln_eps, out, M, N, nrg, ncr,
           span / nct, (hipStream_t)stream};
    switch (K) {
        case 192: return launch_k<192>(a, act, nct);
        case 256: return launch_k<256>(a, act, nct);
        case 384: return launch_k<384>(a, act, nct);
        default: return soc_xs::launch_wide(K, a, act, nct);
    }
}
