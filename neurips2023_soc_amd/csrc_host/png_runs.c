/* libsoc_host.so: a run-length PNG encoder for masks and label maps (include/soc_host.h).  Plain C99, no dependencies. */
#include "soc_host.h"

#include <stdlib.h>
#include <string.h>

int soc_host_abi_version(void) { return SOC_HOST_ABI_VERSION; }

size_t soc_png_bound(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    const size_t raw = (size_t)h * ((size_t)w + 1);
    return raw + raw / 8 + 1024 + 3 * 256;          /* 9 bits per literal at worst + chunk framing + palette */
}

/* ---- checksums ---------------------------------------------------------------------------------------------------- */
static uint32_t crc_table[8][256];
static volatile int crc_ready = 0;

static void crc_init(void) {                          /* idempotent: racing threads write the same values */
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[0][n] = c;
    }
    for (uint32_t n = 0; n < 256; ++n)
        for (int t = 1; t < 8; ++t) crc_table[t][n] = (crc_table[t - 1][n] >> 8) ^ crc_table[0][crc_table[t - 1][n] & 0xFF];
    __sync_synchronize();
    crc_ready = 1;
}

static uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    if (!crc_ready) crc_init();
    crc = ~crc;
    while (n >= 8) {                                   /* slicing-by-8 */
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = crc_table[7][a & 0xFF] ^ crc_table[6][(a >> 8) & 0xFF] ^ crc_table[5][(a >> 16) & 0xFF] ^ crc_table[4][a >> 24] ^
              crc_table[3][b & 0xFF] ^ crc_table[2][(b >> 8) & 0xFF] ^ crc_table[1][(b >> 16) & 0xFF] ^ crc_table[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = crc_table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

/* Adler-32 advanced over `len` copies of byte v in one step (the encoder walks the data run by run anyway):
 * a' = a + v len,  b' = b + a len + v len (len + 1) / 2   (mod 65521); len < 2^32, so everything fits 64 bits. */
typedef struct { uint64_t a, b; } adler;
static inline void adler_run(adler* s, uint8_t v, uint64_t len) {
    s->b = (s->b + (s->a % 65521u) * (len % 65521u) + (uint64_t)v * ((len * (len + 1) / 2) % 65521u)) % 65521u;
    s->a = (s->a + (uint64_t)v * len) % 65521u;
}

/* ---- bit writer (deflate packs bits LSB first; Huffman codes go in MSB first, i.e. bit-reversed) -------------------- */
typedef struct { uint8_t* p; uint8_t* end; uint64_t acc; int n; int overflow; } bitw;

static inline void bw_put(bitw* w, uint32_t bits, int count) {
    w->acc |= (uint64_t)bits << w->n;
    w->n += count;
    while (w->n >= 8) {
        if (w->p < w->end) *w->p++ = (uint8_t)w->acc; else w->overflow = 1;
        w->acc >>= 8;
        w->n -= 8;
    }
}

static inline uint32_t rev(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* fixed Huffman literal / length code of symbol s (RFC 1951 3.2.6), already bit-reversed */
static uint16_t lit_code[288];
static uint8_t lit_bits[288];
static volatile int huff_ready = 0;

static void huff_init(void) {
    for (int s = 0; s < 288; ++s) {
        uint32_t code;
        int bits;
        if (s < 144) { code = 0x30 + s; bits = 8; }
        else if (s < 256) { code = 0x190 + (s - 144); bits = 9; }
        else if (s < 280) { code = s - 256; bits = 7; }
        else { code = 0xC0 + (s - 280); bits = 8; }
        lit_code[s] = (uint16_t)rev(code, bits);
        lit_bits[s] = (uint8_t)bits;
    }
    __sync_synchronize();
    huff_ready = 1;
}

static inline void put_literal(bitw* w, uint8_t v) { bw_put(w, lit_code[v], lit_bits[v]); }

/* a run of `len` (3..258) copies of the previous byte: length symbol + extra bits, distance code 0 (= 1), five bits */
static inline void put_run(bitw* w, int len) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                      131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int k = 28;
    if (len < 258) {
        k = 0;
        while (k < 27 && base[k + 1] <= len) ++k;
    }
    bw_put(w, lit_code[257 + k], lit_bits[257 + k]);
    if (extra[k]) bw_put(w, (uint32_t)(len - base[k]), extra[k]);
    bw_put(w, 0, 5);
}

static inline void put_u32be(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

/* writes length + type + data + crc; `data` may already sit at p + 8 */
static uint8_t* chunk(uint8_t* p, const char type[4], const uint8_t* data, uint32_t len) {
    put_u32be(p, len);
    memcpy(p + 4, type, 4);
    if (len && data != p + 8) memmove(p + 8, data, len);
    put_u32be(p + 8 + len, crc32_update(0, p + 4, 4 + (size_t)len));
    return p + 12 + len;
}

long soc_png_encode_u8(const uint8_t* img, int h, int w, long row_stride, int binarize, const uint8_t* palette_rgb,
                       int n_colors, uint8_t* out, size_t cap) {
    if (!img || !out || h <= 0 || w <= 0 || row_stride < w) return -1;
    if (palette_rgb && (n_colors < 1 || n_colors > 256)) return -1;
    if (cap < soc_png_bound(h, w)) return -2;
    if (!huff_ready) huff_init();
    const size_t line = (size_t)w + 1, raw_n = (size_t)h * line;
    uint8_t* raw = (uint8_t*)malloc(raw_n + 8);       /* the filtered scanlines: what the zlib stream carries */
    if (!raw) return -3;
    /* filter type 2 ("Up"): byte - byte above (zero above the first row).  Rows equal to the one above become zeros. */
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = img + (size_t)y * (size_t)row_stride;
        const uint8_t* up = y ? src - row_stride : NULL;
        uint8_t* dst = raw + (size_t)y * line;
        dst[0] = 2;
        if (binarize) {
            if (up) for (int x = 0; x < w; ++x) dst[1 + x] = (uint8_t)((src[x] ? 255 : 0) - (up[x] ? 255 : 0));
            else for (int x = 0; x < w; ++x) dst[1 + x] = src[x] ? 255 : 0;
        } else {
            if (up) for (int x = 0; x < w; ++x) dst[1 + x] = (uint8_t)(src[x] - up[x]);
            else memcpy(dst + 1, src, (size_t)w);
        }
    }
    memset(raw + raw_n, 0xA5, 8);                     /* guard bytes for the 8-byte run scan (never equal to a run of zeros) */

    uint8_t* p = out;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    memcpy(p, sig, 8);
    p += 8;
    uint8_t ihdr[13];
    put_u32be(ihdr, (uint32_t)w);
    put_u32be(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8;
    ihdr[9] = palette_rgb ? 3 : 0;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    p = chunk(p, "IHDR", ihdr, 13);
    if (palette_rgb) p = chunk(p, "PLTE", palette_rgb, (uint32_t)(3 * n_colors));

    /* IDAT: zlib header, one final fixed-Huffman block, adler32 */
    uint8_t* idat = p + 8;
    bitw bw = {idat + 2, out + cap - 16, 0, 0, 0};
    idat[0] = 0x78;
    idat[1] = 0x01;
    bw_put(&bw, 1, 1);                                /* BFINAL */
    bw_put(&bw, 1, 2);                                /* BTYPE = 01 */
    size_t i = 0;
    adler ad = {1, 0};
    while (i < raw_n) {
        const uint8_t b = raw[i];
        put_literal(&bw, b);
        size_t j = i + 1;
        uint64_t pat;
        memset(&pat, b, 8);
        for (;;) {                                    /* bytes equal to b behind position i, eight at a time */
            uint64_t v;
            memcpy(&v, raw + j, 8);
            if (j + 8 <= raw_n && v == pat) j += 8; else break;
        }
        while (j < raw_n && raw[j] == b) ++j;
        size_t run = j - i - 1;
        adler_run(&ad, b, (uint64_t)(j - i));
        while (run >= 3) {
            const int len = run >= 258 ? 258 : (int)run;
            put_run(&bw, len);
            run -= (size_t)len;
        }
        while (run--) put_literal(&bw, b);
        i = j;
    }
    bw_put(&bw, lit_code[256], lit_bits[256]);        /* end of block */
    if (bw.n) bw_put(&bw, 0, 8 - bw.n);               /* flush to a byte boundary */
    free(raw);
    if (bw.overflow || bw.p + 4 > out + cap - 16) return -2;
    put_u32be(bw.p, (uint32_t)((ad.b << 16) | ad.a));
    const uint32_t idat_len = (uint32_t)(bw.p + 4 - idat);
    p = chunk(p, "IDAT", idat, idat_len);
    p = chunk(p, "IEND", NULL, 0);
    return (long)(p - out);
}
