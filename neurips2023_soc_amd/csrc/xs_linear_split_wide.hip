// K24, wide inputs: the K = 512 / 768 / 1024 instantiations of xs_linear_split.h (four 512-register waves per workgroup),
// a translation unit of their own so that the two halves of K24 compile side by side.  K = 512 / 1024 are Video-Swin-B's
// stage 2 / 3 widths (reference models/video_swin_transformer.py:144-166, 219, 254-259), K = 768 Swin-T / -S stage 3.
#include "xs_linear_split.h"

namespace soc_xs {

size_t packed_bytes_wide(int K, int N) {
    switch (K) {
        case 512: return packed_bytes<512>(N);
        case 768: return packed_bytes<768>(N);
        default: return packed_bytes<1024>(N);
    }
}

void pack_wide(int K, const float* w, void* packed, int N, int blocks, hipStream_t st) {
    u32x4* img = reinterpret_cast<u32x4*>(packed);
    switch (K) {
        case 512: hipLaunchKernelGGL(xs_pack_kernel<512>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
        case 768: hipLaunchKernelGGL(xs_pack_kernel<768>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
        default: hipLaunchKernelGGL(xs_pack_kernel<1024>, dim3(blocks), dim3(256), 0, st, w, img, N); break;
    }
}

int launch_wide(int K, const Args& a, int act, int nct) {
    switch (K) {
        case 512: return launch_k<512>(a, act, nct);
        case 768: return launch_k<768>(a, act, nct);
        default: return launch_k<1024>(a, act, nct);
    }
}

}  // namespace soc_xs
