// K22: the deformable encoder's feed-forward block in one kernel, on the bf16 matrix cores (exact three-way split).
//
//   out[M, 256] = ReLU(x[M, 256] . W1[F, 256]^T + b1) . W2[256, F]^T + b2 (+ residual)            (f32 in, f32 out)
//
// Replaces linear1 -> ReLU -> linear2 of DeformableTransformerEncoderLayer.forward_ffn
// (reference models/deformable_transformer.py:253-263; F = dim_feedforward = 2048, 38 560 tokens at the BASELINE config):
// two GEMMs of 40 GFLOP each with a 316 MB hidden tensor written and read between them.  Here the hidden layer never
// leaves the registers.  K13b's data flow (ws_linear_split.hip), chained:
//   * a wave owns 16 rows of x and keeps them as split bf16 fragments (K = 256: 8 steps x 3 planes = 96 VGPRs);
//   * the hidden layer is produced 32 columns at a time: H^T[32 x 16] = W1_chunk . x^T on v_mfma_f32_16x16x32_bf16 (six
//     products per step, smallest first), + b1, ReLU, and the two 16 x 16 accumulator tiles -- lane (row m, kq) holds hidden
//     columns 4 kq .. 4 kq + 3 of each -- ARE the B operand of the next product once split: eight k values per lane, in an
//     order the packed image of W2 mirrors (k-permutation at pack time, free);
//   * out^T[256 x 16] += W2_chunk . H_chunk^T: 16 output tiles x six products, accumulated over the F / 32 chunks in 64
//     VGPRs; + b2 (+ residual) and 16-B stores at the end.
// The weights stream through LDS: per chunk a 48-KB W1 block and a 48-KB W2 block (pre-split, pre-permuted, laid out as
// the fragments are read: soc_ffn_split_pack_f32), copied by LDS-DMA into two alternating slots, one block ahead of the
// MFMAs, one barrier per block.  All 8 waves of a workgroup consume the same block for their own rows.
// Arithmetic: linear_split.hip (a = a0 + a1 + a2 exactly; a2 b0, a1 b1, a0 b2, a1 b0, a0 b1, a0 b0; dropped <= 2^-23 |a b|).
// Co-residence rule: all 256 VGPRs claimed, waves retire behind a barrier, no SGPR operands in packed
// f32 code (tests/test_isa_rules.py).
#include "soc_common.h"
#include <atomic>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512;
constexpr int C = 256;                     // model width (K of linear1, N of linear2)
constexpr int KS = C / 32;                 // 8 k-steps of linear1
constexpr int OT = C / 16;                 // 16 output tiles of linear2
constexpr int BLOCK_U4 = 2 * KS * 3 * 64;  // 16-B pieces per weight block (= OT * 3 * 64): 3072 -> 48 KB
static_assert(BLOCK_U4 == OT * 3 * 64, "both block kinds have the same size");
constexpr int BLOCK_BYTES = BLOCK_U4 * 16;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        h0[i] = a0; h1[i] = a1; h2[i] = (__bf16)r2;
    }
}

#define MFMA6(acc, wa, xb0, xb1, xb2)                                                        \
    do {                                                                                     \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[2], xb0, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], xb1, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xb2, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], xb0, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xb1, acc, 0, 0, 0);             \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], xb0, acc, 0, 0, 0);             \
    } while (0)

// img: [F / 32 chunks][2 blocks: W1 then W2][BLOCK_U4] 16-B pieces
__global__ __launch_bounds__(THREADS, 2) void ffn_split_kernel(
    const float* __restrict__ x, const u32x4* __restrict__ img, const float* __restrict__ b1,
    const float* __restrict__ b2, const float* __restrict__ res, float* __restrict__ out, long M, int F) {
    extern __shared__ __attribute__((aligned(16))) u32x4 slots[];      // two slots of BLOCK_U4
    asm volatile("v_mov_b32 v255, 0" ::: "v255");                      // own the CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int nblocks = 2 * (F / 32);
    const long ntiles = (M + 15) >> 4;
    auto dma = [&](int blk, int slot) {                                // 48 KB: 6 x (512 threads x 16 B)
        const u32x4* src = img + (long)blk * BLOCK_U4;
        u32x4* dst = slots + slot * BLOCK_U4;
#pragma unroll
        for (int u = 0; u < BLOCK_U4 / THREADS; ++u)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + u * THREADS + wave * 64 + lane),
                (__attribute__((address_space(3))) void*)(dst + u * THREADS + wave * 64), 16, 0, 0);
    };
    for (long t0 = (long)blockIdx.x * 8; t0 < ntiles; t0 += (long)gridDim.x * 8) {
        const long tile = t0 + wave;                                   // this wave's rows (clamped: a dead wave still
        const long m = min(tile * 16 + r, M - 1);                      // takes part in the barriers)
        bf16x8 xb[KS][3];
        {
            const float4* xp = reinterpret_cast<const float4*>(x + m * C + 8 * kq);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 lo = xp[8 * s], hi = xp[8 * s + 1];
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                split8(v, xb[s][0], xb[s][1], xb[s][2]);
            }
        }
        f32x4 acc2[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc2[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 hb[3];
        dma(0, 0);
        __syncthreads();                                               // block 0 has landed (vmcnt(0) precedes the barrier)
        for (int blk = 0; blk < nblocks; ++blk) {
            if (blk + 1 < nblocks) dma(blk + 1, (blk + 1) & 1);        // the next block flies during this block's MFMAs
            const u32x4* wl = slots + (blk & 1) * BLOCK_U4 + lane;
            const int hc = blk >> 1;
            if ((blk & 1) == 0) {
                // ---- linear1, hidden columns [32 hc, 32 hc + 32): H^T tiles j = 0, 1; accumulators start from b1
                f32x4 acc1[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                // b1 is added behind the MFMAs: its load has the whole block to land
                const float4 bq0 = *reinterpret_cast<const float4*>(b1 + 32 * hc + 4 * kq);
                const float4 bq1 = *reinterpret_cast<const float4*>(b1 + 32 * hc + 16 + 4 * kq);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bf16x8 wa[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wa[pl] = __builtin_bit_cast(bf16x8, wl[((j * KS + s) * 3 + pl) * 64]);
                        MFMA6(acc1[j], wa, xb[s][0], xb[s][1], xb[s][2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float v[8] = {fmaxf(acc1[0][0] + bq0.x, 0.f), fmaxf(acc1[0][1] + bq0.y, 0.f), fmaxf(acc1[0][2] + bq0.z, 0.f),
                                    fmaxf(acc1[0][3] + bq0.w, 0.f), fmaxf(acc1[1][0] + bq1.x, 0.f), fmaxf(acc1[1][1] + bq1.y, 0.f),
                                    fmaxf(acc1[1][2] + bq1.z, 0.f), fmaxf(acc1[1][3] + bq1.w, 0.f)};
                split8(v, hb[0], hb[1], hb[2]);
            } else {
                // ---- linear2: every output tile gets this chunk's 32 hidden columns
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    bf16x8 wa[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wa[pl] = __builtin_bit_cast(bf16x8, wl[(ot * 3 + pl) * 64]);
                    MFMA6(acc2[ot], wa, hb[0], hb[1], hb[2]);
                    if ((ot & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // a few fragment reads ahead, not all 48
                }
            }
            __syncthreads();                   // every wave is done with this slot; the next block has landed
        }
        // ---- lane (m = r, kq) holds out[m][16 ot + 4 kq .. + 3]
        if (tile < ntiles && tile * 16 + r < M) {
            long mo = m * C + 4 * kq;
            asm volatile("" : "+v"(mo));       // the epilogue's addresses are worked out here, not carried through the block loop
            float* orow = out + mo;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const float4 bq = *reinterpret_cast<const float4*>(b2 + 16 * ot + 4 * kq);
                float4 o = make_float4(acc2[ot][0] + bq.x, acc2[ot][1] + bq.y, acc2[ot][2] + bq.z, acc2[ot][3] + bq.w);
                if (res) {
                    const float4 rr = *reinterpret_cast<const float4*>(res + mo + 16 * ot);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
                *reinterpret_cast<float4*>(orow + 16 * ot) = o;
            }
        }
    }
    __syncthreads();        // the waves retire together
}

// item = (chunk hc, block kind, piece): one 16-B piece of the image = 8 weights split into three planes
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                       u32x4* __restrict__ img, int F) {
    const long total = (long)(F / 32) * 2 * (BLOCK_U4 / 3);            // pieces per plane
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        long rest = idx >> 6;
        const int sub = (int)(rest % 16);                              // W1: (j, s) = (sub >> 3, sub & 7); W2: output tile
        rest /= 16;
        const int kind = (int)(rest & 1), hc = (int)(rest >> 1);
        const int n = lane & 15, kq = lane >> 4;
        float v[8];
        if (kind == 0) {
            const int j = sub >> 3, s = sub & 7;
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
        u32x4* dst = img + ((long)(hc * 2 + kind) * BLOCK_U4) + (sub * 3) * 64 + lane;
        dst[0] = __builtin_bit_cast(u32x4, h0);
        dst[64] = __builtin_bit_cast(u32x4, h1);
        dst[128] = __builtin_bit_cast(u32x4, h2);
    }
}

int num_cus() {
    const int dev = soc_current_device();
    int v = 0;
    if (dev >= 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    return 256;
}

}  // namespace

extern "C" size_t soc_ffn_split_packed_bytes(int C_, int F) {
    if (C_ != C || F <= 0 || F % 32 != 0) return 0;
    return (size_t)(F / 32) * 2 * BLOCK_BYTES;
}

extern "C" int soc_ffn_split_pack_f32(const float* w1, const float* w2, void* packed, int C_, int F, void* stream) {
    if (!w1 || !w2 || !packed) return SOC_EINVAL;
    if (C_ != C || F <= 0 || F % 32 != 0) return SOC_EUNSUPPORTED;
    const long total = (long)(F / 32) * 2 * (BLOCK_U4 / 3);
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(ffn_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w1, w2,
                       reinterpret_cast<u32x4*>(packed), F);
    return soc_check_launch();
}

extern "C" int soc_ffn_split_f32(const float* x, const void* packed, const float* b1, const float* b2, const float* residual,
                                 float* out, long M, int C_, int F, void* stream) {
    if (M < 0 || F <= 0) return SOC_EINVAL;
    if (M == 0) return SOC_OK;
    if (!x || !packed || !b1 || !b2 || !out) return SOC_EINVAL;
    if (C_ != C || F % 32 != 0) return SOC_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)residual | (uintptr_t)out) & 15) != 0)
        return SOC_EUNSUPPORTED;
    static std::atomic<bool> attr_set[SOC_MAX_DEVICES];
    const int dev = soc_current_device();
    if (dev < 0) return SOC_ELAUNCH;
    const void* fn = reinterpret_cast<const void*>(ffn_split_kernel);
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SOC_ELAUNCH;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const long ntiles = (M + 15) >> 4;
    long blocks = (ntiles + 7) / 8;
    const int cus = num_cus();
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(ffn_split_kernel, dim3((unsigned)blocks), dim3(THREADS), 2 * BLOCK_BYTES, (hipStream_t)stream, x,
                       reinterpret_cast<const u32x4*>(packed), b1, b2, residual, out, M, F);
    return soc_check_launch();
}
