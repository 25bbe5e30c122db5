/*
 * soc_host.h -- C ABI of libsoc_host.so: host-side (CPU) helpers of the output side of SOC's inference drivers.  Plain C,
 * no Python / torch types, no global state, thread-safe (the drivers call it from a pool of writer threads).
 *
 * The reference writes every predicted mask as a PNG through Pillow (infer_refytb.py:269-277: 8-bit 'L', 0 / 255;
 * infer_davis.py:285-291: 8-bit palette label maps).  At 160 clips/s per GPU and 8 frames per clip that is 10 000 PNGs per
 * second on an 8-GPU node; zlib at its fastest level costs 1.9 ms of CPU per 720p mask (profiles/r04_files_to_png.json), which
 * made the files -> PNG driver host-bound on the 16-CPU quota.  Masks and label maps are piecewise constant, so a PNG encoder
 * that only ever looks for byte runs is enough.
 */
#ifndef SOC_HOST_H
#define SOC_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOC_HOST_ABI_VERSION 1
int soc_host_abi_version(void);

/* Upper bound of the bytes soc_png_encode_u8 writes for an h x w image (with or without a palette). */
size_t soc_png_bound(int h, int w);

/*
 * 8-bit image [h][w] (row stride `row_stride` bytes) -> a complete PNG file in `out` (capacity `cap`).
 *   palette_rgb == NULL: colour type 0 (grayscale, what Pillow writes for mode 'L');
 *   palette_rgb != NULL: colour type 3 with a PLTE chunk of n_colors (1..256) RGB triples (mode 'P').
 *   binarize != 0: every non-zero input byte is written as 255 (a bool / 0-1 mask -> the reference's 0 / 255 image,
 *   infer_refytb.py:272-275 `mask.astype(np.float32) * 255`), 0 stays 0.
 * Lossless: any PNG reader returns exactly the (binarized) pixels.  Encoding: every scanline with the "Up" filter, ONE
 * zlib stream of one fixed-Huffman deflate block whose only matches are byte runs (distance 1).  Piecewise-constant images
 * (masks, label maps) come out at a few KB per 720p frame; arbitrary images are valid but barely compressed.
 * Returns the number of bytes written, or a negative value: -1 bad argument, -2 `cap` too small, -3 out of memory.
 */
long soc_png_encode_u8(const uint8_t* img, int h, int w, long row_stride, int binarize, const uint8_t* palette_rgb,
                       int n_colors, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
