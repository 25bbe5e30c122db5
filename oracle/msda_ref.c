/*
 * Plain-C restatement of the reference's multi-scale deformable attention forward --
 * TEST INFRASTRUCTURE ONLY (see oracle/soc_oracle.py header); never linked into the product.
 *
 * Follows, per output scalar, reference models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299
 * (loop over levels then points, h_im = loc_h*H - 0.5, w_im = loc_w*W - 0.5, in-range test
 * (-1,H)x(-1,W)) and :33-84 (4-tap bilinear, per-tap zero padding, weights hh*hw, hh*lw, lh*hw, lh*lw).
 * The reference's own CPU entry point only raises (src/cpu/ms_deform_attn_cpu.cpp:16-40), and its
 * CUDA sources cannot be compiled in this image, so parity is pinned through tests/golden/msda_cases.npz
 * (generated with the reference's ms_deform_attn_core_pytorch, functions/ms_deform_attn_func.py:41-61).
 */
#include <math.h>
#include <stdint.h>

#define DEFINE_MSDA(NAME, T, FLOOR)                                                              \
    void NAME(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,          \
              const T* attw, T* out, int N, int S, int M, int D, int L, int Lq, int P) {        \
        const long rs = (long)M * D;                                                             \
        for (long n = 0; n < N; ++n)                                                             \
            for (long q = 0; q < Lq; ++q)                                                        \
                for (long m = 0; m < M; ++m) {                                                   \
                    const long gi = (n * Lq + q) * M + m;                                        \
                    for (long c = 0; c < D; ++c) {                                               \
                        T col = 0;                                                               \
                        for (int l = 0; l < L; ++l) {                                            \
                            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];        \
                            const T* v = value + (n * S + lsi[l]) * rs + m * D + c;              \
                            for (int p = 0; p < P; ++p) {                                        \
                                const long pi = (gi * L + l) * P + p;                            \
                                const T h = loc[2 * pi + 1] * H - (T)0.5;                        \
                                const T w = loc[2 * pi] * W - (T)0.5;                            \
                                if (!(h > -1 && w > -1 && h < H && w < W)) continue;             \
                                const int h0 = (int)FLOOR(h), w0 = (int)FLOOR(w);                \
                                const T lh = h - h0, lw = w - w0, hh = 1 - lh, hw = 1 - lw;      \
                                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                \
                                if (h0 >= 0 && w0 >= 0) v1 = v[((long)h0 * W + w0) * rs];        \
                                if (h0 >= 0 && w0 + 1 <= W - 1) v2 = v[((long)h0 * W + w0 + 1) * rs]; \
                                if (h0 + 1 <= H - 1 && w0 >= 0) v3 = v[((long)(h0 + 1) * W + w0) * rs]; \
                                if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1)                           \
                                    v4 = v[((long)(h0 + 1) * W + w0 + 1) * rs];                  \
                                col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * attw[pi]; \
                            }                                                                    \
                        }                                                                        \
                        out[gi * D + c] = col;                                                   \
                    }                                                                            \
                }                                                                                \
    }

DEFINE_MSDA(soc_oracle_msda_f32, float, floorf)
DEFINE_MSDA(soc_oracle_msda_f64, double, floor)
