// Does VALU work of one kernel stay correct while another kernel's MFMAs run on the same SIMDs?
//
// "victim" kernels run a deterministic chain of FMAs in which every lane of every wave computes the SAME values
// (weights are wave-uniform, inputs are identical), so any lane whose result differs from the reference result is a
// wrong result.  They are launched on one stream while an "aggressor" kernel (a loop of bf16 or f32 MFMAs, or plain
// VALU FMAs) occupies the chip from another stream with spare registers and no LDS, so that the two co-reside.
// Reported: wrong results per victim launch, by 16-lane group and by half of the packed pair.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_mfma_probe pk_mfma_probe.hip && ./pk_mfma_probe [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NW = 160;      // weights per round (K4 has 169 per instance)

// KIND 0: packed f32x2 FMAs, weights read through the scalar cache and broadcast to both halves (K4's form)
// KIND 1: the same arithmetic as two independent scalar FMAs per step
// KIND 2: packed f32x2 FMAs with the weights in vector registers (loaded per lane)
template <int KIND>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ wts, f32x2* __restrict__ out, int rounds) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 h[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) h[c] = (f32x2){0.25f + 0.125f * c, -0.5f + 0.0625f * c};
    for (int r = 0; r < rounds; ++r) {
        const float* __restrict__ W = wts + (r & 7) * NW;      // wave-uniform address
        f32x2 g[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (KIND == 0) {
                f32x2 a = (f32x2){W[128 + o], W[128 + o]};
#pragma unroll
                for (int c = 0; c < 8; ++c) a = __builtin_elementwise_fma((f32x2){W[o * 8 + c], W[o * 8 + c]}, h[c], a);
                g[o] = __builtin_elementwise_max(a, (f32x2){-4.f, -4.f});
            } else if (KIND == 4) {
                f32x2 a = (f32x2){W[128 + o], W[128 + o]};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    f32x2 m = (f32x2){W[o * 8 + c], W[o * 8 + c]} * h[c];
                    asm volatile("" : "+v"(m));                  // keep the multiply and the add apart
                    a += m;
                }
                g[o] = __builtin_elementwise_max(a, (f32x2){-4.f, -4.f});
            } else if (KIND == 1) {
                float ax = W[128 + o], ay = W[128 + o];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ax = __builtin_fmaf(W[o * 8 + c], h[c].x, ax);
                    ay = __builtin_fmaf(W[o * 8 + c], h[c].y, ay);
                    asm volatile("" : "+v"(ax), "+v"(ay));       // keep them scalar FMAs
                }
                g[o] = (f32x2){fmaxf(ax, -4.f), fmaxf(ay, -4.f)};
            } else {
                const volatile float* WV = W;
                float wv[9];
#pragma unroll
                for (int c = 0; c < 8; ++c) wv[c] = WV[o * 8 + c + (threadIdx.x & 0)];
                wv[8] = WV[128 + o];
                f32x2 a = (f32x2){wv[8], wv[8]};
#pragma unroll
                for (int c = 0; c < 8; ++c) a = __builtin_elementwise_fma((f32x2){wv[c], wv[c]}, h[c], a);
                g[o] = __builtin_elementwise_max(a, (f32x2){-4.f, -4.f});
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = g[c] * (f32x2){0.5f, 0.5f};
    }
    f32x2 s = h[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) s += h[c] * (f32x2){1.f + c, 1.f + c};
    out[tid] = s;
}

// KIND 3: no arithmetic -- eight global_load_dword per lane from a table in which every entry holds the same value, summed
__global__ __launch_bounds__(256) void victim_loads(const float* __restrict__ table, f32x2* __restrict__ out, int n) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a += table[(size_t)c * n + tid];
        b += table[(size_t)c * n + ((tid + 4096) % n)];
    }
    out[tid] = (f32x2){a, b};
}

// AGG 3: LDS-DMA (global_load_lds, 16 bytes per lane) in a loop, no MFMA; 4: LDS-DMA + bf16 MFMA
// EXCL: the kernel claims all 256 architectural VGPRs, so that its 2 waves per SIMD fill the 512-entry register file and
// no wave of another kernel can be resident on the same CU.
template <bool MFMA, bool EXCL = false>
__global__ __launch_bounds__(512, 2) void aggressor_dma(const float* __restrict__ src, float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    if (EXCL) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc0 = {};
    const float s = 1.0f + threadIdx.x * 1e-3f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(s + i); b[i] = (__bf16)(0.5f * s - i); }
    const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + ((size_t)(blockIdx.x & 63) * 512 + threadIdx.x) * 16;
    float t = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 4 + u) & 63) * 524288),
                                             (__attribute__((address_space(3))) void*)(lds + u * 8192 + wave * 1024), 16, 0, 0);
        if (MFMA) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
        __syncthreads();
        t += reinterpret_cast<const float*>(lds)[(lane * 17 + it) & 8191];
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc0[e];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

// LDS reads or writes (ordinary ds_read_b128 / ds_write_b128, no DMA) next to MFMAs.
//   MF 0: no MFMA, 1: bf16 MFMA, 2: f32 MFMA;  LD 1: ds_read_b128, 2: ds_write_b128
template <int MF, int LD>
__global__ __launch_bounds__(512, 2) void aggressor_lds(float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) float4 lds[2048];
    f32x16 acc0 = {};
    const float s = 1.0f + threadIdx.x * 1e-3f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(s + i); b[i] = (__bf16)(0.5f * s - i); }
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = make_float4(s, s, s, s);
    __syncthreads();
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (LD == 1) {
                const float4 v = lds[(threadIdx.x + 64 * u + it) & 2047];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            } else {
                lds[(threadIdx.x + 512 * u) & 2047] = t;
                t.x += 1.f;
            }
            if (MF == 1) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc0, 0, 0, 0);
            } else if (MF == 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 0.5f * s, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.5f * s, s, acc0, 0, 0, 0);
            }
        }
    }
    float r = t.x + t.y + t.z + t.w;
#pragma unroll
    for (int e = 0; e < 16; ++e) r += acc0[e];
    if (r == 12345.678f) sink[threadIdx.x] = r + lds[threadIdx.x].x;
}

// AGG 0: v_mfma_f32_32x32x16_bf16; 1: v_mfma_f32_32x32x2_f32; 2: v_fma_f32
template <int AGG>
__global__ __launch_bounds__(512, 2) void aggressor(float* __restrict__ sink, int iters) {
    f32x16 acc0 = {}, acc1 = {};
    const float s = 1.0f + threadIdx.x * 1e-3f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(s + i); b[i] = (__bf16)(0.5f * s - i); }
    for (int it = 0; it < iters; ++it) {
        if (AGG == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
            }
        } else if (AGG == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, 0.5f * s, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.25f * s, s, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc0[e] = __builtin_fmaf(acc0[e], 0.999f, s);
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc0[e] + acc1[e];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

template <int KIND>
static void run_victim(const float* w, f32x2* out, int blocks, int rounds, hipStream_t st) {
    hipLaunchKernelGGL(victim<KIND>, dim3(blocks), dim3(256), 0, st, w, out, rounds);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 100;
    const int blocks = 2048, rounds = 48, nthreads = blocks * 256;
    std::vector<float> hw(8 * NW);
    unsigned lcg = 12345u;
    for (auto& v : hw) { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f; }
    float *dw, *sink, *dsrc, *dtab;
    CK(hipMalloc(&dsrc, 64u * 524288u + 65536u));
    CK(hipMemset(dsrc, 0, 64u * 524288u + 65536u));
    f32x2* dout;
    CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&dout, (size_t)nthreads * 8));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    std::vector<f32x2> host(nthreads);
    {
        std::vector<float> tab((size_t)8 * nthreads, 1.25f);
        CK(hipMalloc(&dtab, tab.size() * 4));
        CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    const char* vn[5] = {"pk_fma sgpr-weights", "scalar fma sgpr-weights", "pk_fma vgpr-weights", "global loads only",
                         "pk_mul+pk_add sgpr-weights"};
    const char* an[12] = {"none", "bf16 mfma", "f32 mfma", "valu fma", "lds-dma", "lds-dma+mfma", "lds-dma+mfma, 256 VGPRs",
                          "ds_read+bf16 mfma", "ds_read+f32 mfma", "ds_write+bf16 mfma", "ds_read only", "ds_write+f32 mfma"};
    for (int kind = 0; kind < 5; ++kind) {
        // reference: the victim alone
        if (kind == 0) run_victim<0>(dw, dout, blocks, rounds, sv);
        if (kind == 1) run_victim<1>(dw, dout, blocks, rounds, sv);
        if (kind == 2) run_victim<2>(dw, dout, blocks, rounds, sv);
        if (kind == 3) hipLaunchKernelGGL(victim_loads, dim3(blocks), dim3(256), 0, sv, dtab, dout, nthreads);
        if (kind == 4) run_victim<4>(dw, dout, blocks, rounds, sv);
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(host.data(), dout, (size_t)nthreads * 8, hipMemcpyDeviceToHost));
        const f32x2 ref = host[0];
        long selfbad = 0;
        for (int i = 0; i < nthreads; ++i) selfbad += (host[i].x != ref.x) + (host[i].y != ref.y);
        printf("victim %-24s alone: ref (%.9g, %.9g), lanes differing from lane 0: %ld\n", vn[kind], ref.x, ref.y, selfbad);
        for (int agg = 0; agg < 12; ++agg) {
            if (agg >= 7 && kind != 0 && kind != 1) continue;
            long bad_launches = 0, grp[4] = {0, 0, 0, 0}, half[2] = {0, 0}, launches = 0;
            float ms_a = 0.f;
            for (int r = 0; r < reps; ++r) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, sa));
                if (agg == 1) hipLaunchKernelGGL(aggressor<0>, dim3(512), dim3(512), 0, sa, sink, 3000);
                if (agg == 2) hipLaunchKernelGGL(aggressor<1>, dim3(512), dim3(512), 0, sa, sink, 400);
                if (agg == 3) hipLaunchKernelGGL(aggressor<2>, dim3(512), dim3(512), 0, sa, sink, 1500);
                if (agg == 4) hipLaunchKernelGGL(aggressor_dma<false>, dim3(512), dim3(512), 0, sa, dsrc, sink, 1500);
                if (agg == 5) hipLaunchKernelGGL(aggressor_dma<true>, dim3(512), dim3(512), 0, sa, dsrc, sink, 1500);
                if (agg == 6) hipLaunchKernelGGL((aggressor_dma<true, true>), dim3(512), dim3(512), 0, sa, dsrc, sink, 1500);
                if (agg == 7) hipLaunchKernelGGL((aggressor_lds<1, 1>), dim3(512), dim3(512), 0, sa, sink, 3000);
                if (agg == 8) hipLaunchKernelGGL((aggressor_lds<2, 1>), dim3(512), dim3(512), 0, sa, sink, 1000);
                if (agg == 9) hipLaunchKernelGGL((aggressor_lds<1, 2>), dim3(512), dim3(512), 0, sa, sink, 3000);
                if (agg == 10) hipLaunchKernelGGL((aggressor_lds<0, 1>), dim3(512), dim3(512), 0, sa, sink, 6000);
                if (agg == 11) hipLaunchKernelGGL((aggressor_lds<2, 2>), dim3(512), dim3(512), 0, sa, sink, 1000);
                CK(hipEventRecord(e1, sa));
                for (int v = 0; v < 4; ++v) {
                    if (kind == 0) run_victim<0>(dw, dout, blocks, rounds, sv);
                    if (kind == 1) run_victim<1>(dw, dout, blocks, rounds, sv);
                    if (kind == 2) run_victim<2>(dw, dout, blocks, rounds, sv);
                    if (kind == 3) hipLaunchKernelGGL(victim_loads, dim3(blocks), dim3(256), 0, sv, dtab, dout, nthreads);
                    if (kind == 4) run_victim<4>(dw, dout, blocks, rounds, sv);
                    CK(hipMemcpyAsync(host.data(), dout, (size_t)nthreads * 8, hipMemcpyDeviceToHost, sv));
                    CK(hipStreamSynchronize(sv));
                    long b = 0;
                    for (int i = 0; i < nthreads; ++i) {
                        const bool bx = host[i].x != ref.x, by = host[i].y != ref.y;
                        if (bx || by) { ++b; ++grp[(i & 63) >> 4]; half[0] += bx; half[1] += by; }
                    }
                    bad_launches += b != 0;
                    ++launches;
                }
                CK(hipStreamSynchronize(sa));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_a += ms;
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
            printf("  beside %-24s (%.2f ms each): %ld of %ld victim launches wrong; wrong lanes by group [%ld %ld %ld %ld], by half [%ld %ld]\n",
                   an[agg], ms_a / reps, bad_launches, launches, grp[0], grp[1], grp[2], grp[3], half[0], half[1]);
            fflush(stdout);
        }
    }
    return 0;
}
