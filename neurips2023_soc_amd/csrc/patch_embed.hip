// K21: Video-Swin patch embedding, (1,4,4)/(1,4,4) convolution + LayerNorm in one pass (HBM-bound).
//
//   frames [N, 3, H, W] f32 (N = batch * frames)  ->  out [N, ceil(H/4), ceil(W/4), C] token-major
//   out[n, y, x, :] = LayerNorm_C( bias + sum_{c,kh,kw} w[:, c, kh, kw] * frames[n, c, 4y + kh, 4x + kw] )
// (pixels past the right / bottom edge read as zeros), i.e. PatchEmbed3D.forward of the reference
// (models/video_swin_transformer.py:438-456: F.pad, the Conv3d `proj` with kernel = stride = patch size, flatten,
// `norm`) for the (1,4,4) patch SOC uses (models/video_swin_transformer.py:676).  Through the library this was an
// im2col copy (22 MB), a K = 48 GEMM that ran at 7.6 TFLOP/s (140 us at the BASELINE config) and a LayerNorm pass over
// the 44 MB token map; here the 22 MB clip is read once and the normalised token map is written once.
//
// A workgroup (4 waves) owns tiles of 64 consecutive tokens.  The weights [48][C] stay in LDS for the workgroup's life;
// a tile's pixels are loaded with 16-B loads (lane = token: consecutive tokens of an image row are consecutive 16-B
// pieces) into a [48][64] LDS tile; wave q computes channels [q C/4, (q+1) C/4) of the 64 tokens -- lane = token, the
// input value of a step is one conflict-free LDS word, the weights are wave-uniform 16-B broadcast reads -- the row
// statistics are exchanged between the four waves through LDS (two-pass mean / variance, as torch's LayerNorm) and the
// finished [64][C] tile leaves through LDS as whole 16-B-per-lane rows (64 tokens x C floats are contiguous in `out`).
#include "soc_common.h"

namespace {

constexpr int TOK = 64;          // tokens per tile
constexpr int KIN = 48;          // 3 x 4 x 4 inputs per token

template <int C>
__global__ __launch_bounds__(256, 2) void patch_embed_ln_kernel(
    const float* __restrict__ frames, const float* __restrict__ weight, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
    long tokens, int H, int W, int Hp, int Wp, float eps) {
    constexpr int CQ = C / 4;                       // channels per wave
    constexpr int OS = C + 4;                       // output tile row stride (floats): 16-B aligned, conflict-light
    __shared__ __attribute__((aligned(16))) float Wl[KIN * C];      // [k][c]
    // the pixel tile [k][token] and, once every wave is through the convolution (two barriers later), the output tile
    // [token][c] share one buffer
    __shared__ __attribute__((aligned(16))) float U[TOK * OS > KIN * TOK ? TOK * OS : KIN * TOK];
    float* Xt = U;
    float* Ol = U;
    __shared__ float red[2][4][TOK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);

    // weights: reference layout [C][3][1][4][4] = [c][k]; LDS holds the transpose
    for (int i = tid; i < KIN * C; i += 256) {
        const int c = i / KIN, k = i - c * KIN;
        Wl[k * C + c] = weight[i];
    }
    float bq[CQ], gq[CQ], eq[CQ];
#pragma unroll
    for (int j = 0; j < CQ; ++j) {
        bq[j] = bias ? bias[q * CQ + j] : 0.f;
        gq[j] = gamma[q * CQ + j];
        eq[j] = beta[q * CQ + j];
    }
    const long tiles = (tokens + TOK - 1) / TOK;
    const bool vec_ok = (W & 3) == 0;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();                            // the previous tile's Ol / Xt readers are done (and Wl is staged)
        // ---- stage the pixels: item = (ck = c * 4 + kh, token), 4 pixels each
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int item = tid + 256 * it;
            const int ck = item >> 6, tk = item & 63;
            const long t = tile * TOK + tk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < tokens) {
                const int xp = (int)(t % Wp);
                const int yp = (int)((t / Wp) % Hp);
                const long n = t / ((long)Wp * Hp);
                const int c = ck >> 2, y = 4 * yp + (ck & 3), x = 4 * xp;
                if (y < H) {
                    const float* src = frames + ((n * 3 + c) * H + y) * (long)W + x;
                    if (vec_ok) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (x + 1 < W) v.y = src[1];
                        if (x + 2 < W) v.z = src[2];
                        if (x + 3 < W) v.w = src[3];
                    }
                }
            }
            Xt[(ck * 4 + 0) * TOK + tk] = v.x;
            Xt[(ck * 4 + 1) * TOK + tk] = v.y;
            Xt[(ck * 4 + 2) * TOK + tk] = v.z;
            Xt[(ck * 4 + 3) * TOK + tk] = v.w;
        }
        __syncthreads();
        // ---- the convolution: lane = token, this wave's CQ channels
        float acc[CQ];
#pragma unroll
        for (int j = 0; j < CQ; ++j) acc[j] = bq[j];
#pragma unroll 4
        for (int k = 0; k < KIN; ++k) {
            const float x = Xt[k * TOK + lane];
            const float4* wr = reinterpret_cast<const float4*>(Wl + k * C + q * CQ);
#pragma unroll
            for (int j4 = 0; j4 < CQ / 4; ++j4) {
                const float4 w4 = wr[j4];
                acc[4 * j4 + 0] = fmaf(x, w4.x, acc[4 * j4 + 0]);
                acc[4 * j4 + 1] = fmaf(x, w4.y, acc[4 * j4 + 1]);
                acc[4 * j4 + 2] = fmaf(x, w4.z, acc[4 * j4 + 2]);
                acc[4 * j4 + 3] = fmaf(x, w4.w, acc[4 * j4 + 3]);
            }
        }
        // ---- LayerNorm over the C channels of a token: four waves hold a quarter each
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CQ; ++j) s += acc[j];
        red[0][q][lane] = s;
        __syncthreads();
        const float mean = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) * (1.0f / C);
        float v2 = 0.f;
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const float d = acc[j] - mean;
            v2 = fmaf(d, d, v2);
        }
        red[1][q][lane] = v2;
        __syncthreads();
        const float var = ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) * (1.0f / C);
        const float rstd = rsqrtf(var + eps);
        float4* orow = reinterpret_cast<float4*>(Ol + lane * OS + q * CQ);
#pragma unroll
        for (int j4 = 0; j4 < CQ / 4; ++j4)
            orow[j4] = make_float4((acc[4 * j4 + 0] - mean) * rstd * gq[4 * j4 + 0] + eq[4 * j4 + 0],
                                   (acc[4 * j4 + 1] - mean) * rstd * gq[4 * j4 + 1] + eq[4 * j4 + 1],
                                   (acc[4 * j4 + 2] - mean) * rstd * gq[4 * j4 + 2] + eq[4 * j4 + 2],
                                   (acc[4 * j4 + 3] - mean) * rstd * gq[4 * j4 + 3] + eq[4 * j4 + 3]);
        __syncthreads();
        // ---- the tile is TOK * C contiguous floats of `out`
        const long base = tile * TOK;
        const int live = (int)min((long)TOK, tokens - base);
        float4* dst = reinterpret_cast<float4*>(out + base * C);
        for (int i = tid; i < live * (C / 4); i += 256) {
            const int tk = i / (C / 4), c4 = i - tk * (C / 4);
            dst[i] = *reinterpret_cast<const float4*>(Ol + tk * OS + 4 * c4);
        }
    }
}

}  // namespace

extern "C" int soc_patch_embed_layernorm_f32(const float* frames, const float* weight, const float* bias,
                                             const float* gamma, const float* beta, float* out, int N, int H, int W,
                                             int C, float eps, void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0) return SOC_EINVAL;
    if (N == 0) return SOC_OK;
    if (!frames || !weight || !gamma || !beta || !out) return SOC_EINVAL;
    if (C != 96 && C != 128) return SOC_EUNSUPPORTED;
    const int Hp = (H + 3) / 4, Wp = (W + 3) / 4;
    const long tokens = (long)N * Hp * Wp;
    const long tiles = (tokens + TOK - 1) / TOK;
    const unsigned blocks = (unsigned)(tiles < 1024 ? tiles : 1024);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C == 96)
        hipLaunchKernelGGL(patch_embed_ln_kernel<96>, dim3(blocks), dim3(256), 0, st, frames, weight, bias, gamma, beta,
                           out, tokens, H, W, Hp, Wp, eps);
    else
        hipLaunchKernelGGL(patch_embed_ln_kernel<128>, dim3(blocks), dim3(256), 0, st, frames, weight, bias, gamma, beta,
                           out, tokens, H, W, Hp, Wp, eps);
    return soc_check_launch();
}
