// Device helpers shared by the one-row-per-workgroup kernels (K15 csrc/dec_cross_attn.hip, K16 csrc/row_mlp.hip):
// a wave computes R (16 or 8) dot products of length 256 at a time -- R coalesced 1-KB weight-row loads in flight, the
// R x 64 partial products reduced by a transposing butterfly (17 cross-lane moves instead of 16 x 6 for R = 16).
#pragma once
#include <hip/hip_runtime.h>

constexpr int ROW_DM = 256;

// One halving step of the transposing reduction: lanes whose bit `BIT` is clear keep the lower half of the values, the
// others the upper half; each adds what its partner (lane ^ BIT) held of the same half.
template <int N, int BIT>
__device__ __forceinline__ void fold(const float (&v)[2 * N], float (&o)[N], const int lane) {
    const bool up = lane & BIT;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float send = up ? v[i] : v[i + N];
        const float keep = up ? v[i + N] : v[i];
        o[i] = keep + __shfl_xor(send, BIT);
    }
}

// v[i] (i < R) of every lane -> sum over the 64 lanes of v[idx]; idx = the top log2(R) bits of the lane id (bit 5 the
// most significant); the lanes that share those bits all end up with the same value.  row_index(lane) gives idx.
template <int R>
__device__ __forceinline__ float reduce_rows(float (&v)[R], const int lane);

template <>
__device__ __forceinline__ float reduce_rows<16>(float (&v)[16], const int lane) {
    float a[8], b[4], c[2], d[1];
    fold<8, 32>(v, a, lane);
    fold<4, 16>(a, b, lane);
    fold<2, 8>(b, c, lane);
    fold<1, 4>(c, d, lane);
    float r = d[0];
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 1);
    return r;
}

template <>
__device__ __forceinline__ float reduce_rows<8>(float (&v)[8], const int lane) {
    float a[4], b[2], c[1];
    fold<4, 32>(v, a, lane);
    fold<2, 16>(a, b, lane);
    fold<1, 8>(b, c, lane);
    float r = c[0];
    r += __shfl_xor(r, 4);
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 1);
    return r;
}

template <int R>
__device__ __forceinline__ int row_index(const int lane) { return R == 16 ? (lane >> 2) & 15 : (lane >> 3) & 7; }
template <int R>
__device__ __forceinline__ bool row_writer(const int lane) { return R == 16 ? (lane & 3) == 0 : (lane & 7) == 0; }

// Bits (5,4,3,2) of the lane id as a number with bit 5 the MOST significant = (lane >> 2) & 15 read MSB-first: the fold
// order above maps bit 5 to the top half first, so the natural binary value of those bits is the row index.

// dst[j0 + i] = act(dot(W[j0 + i, 0:256], x) + bias[j0 + i] * bscale) for i < R and j0 + i < n_out (rows past the end
// re-read the last row and are not stored); x4 = x[4*lane .. 4*lane+3]; bias may be null.
template <int R>
__device__ __forceinline__ void matvec_rows(const float* __restrict__ W, const float* __restrict__ bias, const float bscale,
                                            const int j0, const int n_out, const float4 x4, float* dst, const int lane,
                                            const int relu) {
    float4 w[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int j = min(j0 + i, n_out - 1);
        w[i] = reinterpret_cast<const float4*>(W + (long)j * ROW_DM)[lane];
    }
    float v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = (w[i].x * x4.x + w[i].y * x4.y) + (w[i].z * x4.z + w[i].w * x4.w);
    const float r = reduce_rows<R>(v, lane);
    if (row_writer<R>(lane)) {
        const int j = j0 + row_index<R>(lane);
        if (j < n_out) {
            float y = r + (bias ? bias[j] * bscale : 0.f);
            if (relu) y = fmaxf(y, 0.f);
            dst[j] = y;
        }
    }
}
