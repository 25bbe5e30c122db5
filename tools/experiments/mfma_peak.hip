// What the bf16 matrix cores of THIS box sustain (tools/experiments: not part of the library): a bare
// v_mfma_f32_16x16x32_bf16 loop, operands in registers, 16 independent accumulators per wave, one or two waves per SIMD, on
// random and on all-zero operands, after two seconds of back-to-back launches (the chip lowers its clock under load --
// MI355X_MICROARCH.md "DVFS give-back").  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_peak tools/experiments/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ in, float* __restrict__ out,
                                                 unsigned long long* __restrict__ clk, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, in[(i * 64 + lane)]);
        b[i] = __builtin_bit_cast(bf16x8, in[((4 + i) * 64 + lane)]);
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int cus = 256, iters = 4096;
    uint4* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 8 * 64 * sizeof(uint4)); hipMalloc(&out, (size_t)cus * 4 * 512 * sizeof(float)); hipMalloc(&clk, cus * 4 * 2 * sizeof(unsigned long long));
    std::vector<unsigned short> h(8 * 64 * 8);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (auto& v : h) {          // bf16 values in [-2, 2): random sign, exponent 125..128, random mantissa
            v = zero ? 0 : (unsigned short)(((rand() & 1) << 15) | ((125 + (rand() & 3)) << 7) | (rand() & 127));
        }
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int waves : {4, 8}) {
            const int blocks = cus * 4;            // four rounds of workgroups per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0.f; int n = 0; double total = 0;
            // two seconds of warm-up at load, then 20 timed launches
            hipEventRecord(e0);
            do { hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(waves * 64), 0, 0, in, out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
                 hipEventElapsedTime(&ms, e0, e1); } while (ms < 2000.f);
            hipEventRecord(e0);
            for (n = 0; n < 20; ++n) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(waves * 64), 0, 0, in, out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            total = 20.0 * blocks * waves * (double)iters * 16 * (2.0 * 16 * 16 * 32);
            std::vector<unsigned long long> hc(blocks * 2);
            hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> ghz;
            for (int i = 0; i < blocks; ++i) ghz.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1);
            std::sort(ghz.begin(), ghz.end());
            printf("%s operands, %d waves per CU (%d per SIMD): %.0f TFLOP/s bf16 dense  (= %.0f TFLOP/s of f32-grade work at six products)  in-kernel clock %.2f GHz (median)\n",
                   zero ? "all-zero" : "random  ", waves, waves / 4, total / (ms * 1e-3) / 1e12, total / (ms * 1e-3) / 6e12, ghz[ghz.size() / 2]);
        }
    }
    return 0;
}
