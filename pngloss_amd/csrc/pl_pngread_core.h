/*
 * pl_pngread_core.h -- the pixel arithmetic of the PNG READ side: what libpng does for
 * rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400) between "inflated IDAT bytes" and "RGBA8 rows":
 *   - the inverse of the five scanline filters (PNG specification section 9; libpng png_read_filter_row)
 *   - the transformations the reference registers: png_set_expand for images without an alpha channel (palette -> RGB, gray of
 *     1/2/4 bits -> 8 bits, tRNS -> alpha; rwpng.c:239-242), filler alpha 255 (:242), png_set_strip_16 = the HIGH byte of every
 *     16-bit sample (:252-254; tRNS keys are compared on the full 16 bits first, libpng expands before it strips),
 *     png_set_gray_to_rgb (:256-258).  No gamma correction is applied to pixels by that reader (the gamma is only recorded).
 * Shared by the HIP kernel (pl_pngread.hip) and the CPU check of tests/c/pngread_host.cpp (test infrastructure).
 */
#ifndef PL_PNGREAD_CORE_H
#define PL_PNGREAD_CORE_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PR_HD __host__ __device__ __forceinline__
#else
#define PR_HD inline
#endif

/* everything the expansion needs to know about one image */
struct PrFormat {
    uint32_t width, height, rowbytes;
    uint8_t color_type, bit_depth, bppf, has_trns;   /* bppf: bytes per complete pixel, at least 1 (the filters' stride) */
    uint16_t key[3];                                 /* tRNS of gray (key[0]) / RGB images: the transparent sample values */
    uint32_t pal[256];                               /* palette images: RGBA8 of every index (alpha from tRNS, 255 beyond it) */
};

PR_HD int pr_channels(int color_type) { return color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 3 ? 1 : color_type == 4 ? 2 : 4; }
PR_HD bool pr_valid(int color_type, int depth)
{
    switch (color_type) {
    case 0: return depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16;
    case 3: return depth == 1 || depth == 2 || depth == 4 || depth == 8;
    case 2: case 4: case 6: return depth == 8 || depth == 16;
    default: return false;
    }
}

/* inverse filter of one byte: x = filtered byte, a = left, b = above, c = upper left (reconstructed bytes, 0 outside the image) */
PR_HD int pr_recon(int ft, int x, int a, int b, int c)
{
    /* selects, not a switch: the rows of a band (lanes of a wave) have different filter types */
    const int da = b - c, db = a - c, dc = da + db;                           /* p - a, p - b, p - c  with  p = a + b - c */
    const int pa = da < 0 ? -da : da, pb = db < 0 ? -db : db, pc = dc < 0 ? -dc : dc;
    const int pae = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    int pred = 0;
    pred = ft == 1 ? a : pred;
    pred = ft == 2 ? b : pred;
    pred = ft == 3 ? ((a + b) >> 1) : pred;
    pred = ft == 4 ? pae : pred;
    return (x + pred) & 255;
}

/* the same for the four bytes of a 4-byte pixel at once, without a branch: on the device the 64 rows of a band (= lanes of a wave) have
 * different filter types, and a switch would run every case one after the other for the whole wave */
PR_HD uint32_t pr_recon4(int ft, uint32_t x, uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t pae = 0u;
    for (int k = 0; k < 4; k++) {
        const int A = (int)((a >> (8 * k)) & 255u), B = (int)((b >> (8 * k)) & 255u), C = (int)((c >> (8 * k)) & 255u);
        const int da = B - C, db = A - C, dc = da + db;                       /* p - a, p - b, p - c  with  p = a + b - c */
        const int pa = da < 0 ? -da : da, pb = db < 0 ? -db : db, pc = dc < 0 ? -dc : dc;
        const int pred = (pa <= pb && pa <= pc) ? A : (pb <= pc ? B : C);
        pae |= (uint32_t)pred << (8 * k);
    }
    const uint32_t avg = (a & b) + (((a ^ b) & 0xfefefefeu) >> 1);           /* per byte floor((a + b) / 2) */
    uint32_t pred = 0u;
    pred = ft == 1 ? a : pred;
    pred = ft == 2 ? b : pred;
    pred = ft == 3 ? avg : pred;
    pred = ft == 4 ? pae : pred;
    return ((x & 0x7f7f7f7fu) + (pred & 0x7f7f7f7fu)) ^ ((x ^ pred) & 0x80808080u);   /* per byte (x + pred) mod 256 */
}

/* RGBA8 (r | g << 8 | b << 16 | a << 24) of pixel x of an unfiltered row */
PR_HD uint32_t pr_expand(const PrFormat &F, const uint8_t *row, uint32_t x)
{
    const int d = F.bit_depth;
    if (F.color_type == 3) {
        uint32_t idx;
        if (d == 8) idx = row[x];
        else { const uint32_t bit = x * (uint32_t)d; idx = (row[bit >> 3] >> (8 - d - (bit & 7))) & ((1u << d) - 1u); }
        return F.pal[idx];
    }
    if (F.color_type == 0) {
        uint32_t v, g;                                    /* v: sample at its own depth, g: 8-bit gray */
        if (d == 16) { v = ((uint32_t)row[2 * x] << 8) | row[2 * x + 1]; g = v >> 8; }
        else if (d == 8) { v = row[x]; g = v; }
        else { const uint32_t bit = x * (uint32_t)d; v = (row[bit >> 3] >> (8 - d - (bit & 7))) & ((1u << d) - 1u); g = v * (255u / ((1u << d) - 1u)); }
        const uint32_t a = (F.has_trns && v == F.key[0]) ? 0u : 255u;
        return g | (g << 8) | (g << 16) | (a << 24);
    }
    if (F.color_type == 4) {
        const uint32_t g = d == 16 ? row[4 * x] : row[2 * x], a = d == 16 ? row[4 * x + 2] : row[2 * x + 1];
        return g | (g << 8) | (g << 16) | (a << 24);
    }
    if (F.color_type == 2) {
        if (d == 16) {
            const uint8_t *p = row + 6 * (size_t)x;
            const bool t = F.has_trns && ((((uint32_t)p[0] << 8) | p[1]) == F.key[0]) && ((((uint32_t)p[2] << 8) | p[3]) == F.key[1]) && ((((uint32_t)p[4] << 8) | p[5]) == F.key[2]);
            return (uint32_t)p[0] | ((uint32_t)p[2] << 8) | ((uint32_t)p[4] << 16) | (t ? 0u : 0xff000000u);
        }
        const uint8_t *p = row + 3 * (size_t)x;
        const bool t = F.has_trns && p[0] == F.key[0] && p[1] == F.key[1] && p[2] == F.key[2];
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | (t ? 0u : 0xff000000u);
    }
    /* colour type 6 */
    if (d == 16) { const uint8_t *p = row + 8 * (size_t)x; return (uint32_t)p[0] | ((uint32_t)p[2] << 8) | ((uint32_t)p[4] << 16) | ((uint32_t)p[6] << 24); }
    const uint8_t *p = row + 4 * (size_t)x;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* fills F from the IHDR fields and the PLTE / tRNS payloads; false: not a format PNG allows */
inline bool pr_format(PrFormat &F, uint32_t width, uint32_t height, int color_type, int depth, const unsigned char *plte, uint32_t plte_entries,
                      const unsigned char *trns, uint32_t trns_bytes)
{
    if (!pr_valid(color_type, depth) || !width || !height) return false;
    F.width = width; F.height = height; F.color_type = (uint8_t)color_type; F.bit_depth = (uint8_t)depth;
    const uint32_t bits = (uint32_t)pr_channels(color_type) * (uint32_t)depth;
    F.rowbytes = (uint32_t)(((uint64_t)width * bits + 7) / 8);
    F.bppf = (uint8_t)(bits >= 8 ? bits / 8 : 1);
    F.has_trns = 0; F.key[0] = F.key[1] = F.key[2] = 0;
    for (int i = 0; i < 256; i++) F.pal[i] = 0xff000000u;          /* libpng: indices beyond PLTE read as black */
    if (color_type == 3) {
        if (!plte) return false;
        for (uint32_t i = 0; i < plte_entries && i < 256; i++) F.pal[i] = (uint32_t)plte[3 * i] | ((uint32_t)plte[3 * i + 1] << 8) | ((uint32_t)plte[3 * i + 2] << 16) | 0xff000000u;
        if (trns) for (uint32_t i = 0; i < trns_bytes && i < 256; i++) F.pal[i] = (F.pal[i] & 0x00ffffffu) | ((uint32_t)trns[i] << 24);
    } else if (trns && color_type == 0 && trns_bytes >= 2) {
        F.has_trns = 1; F.key[0] = (uint16_t)((((uint32_t)trns[0] << 8) | trns[1]) & (depth == 16 ? 0xffffu : ((1u << depth) - 1u)));
    } else if (trns && color_type == 2 && trns_bytes >= 6) {
        F.has_trns = 1;
        for (int k = 0; k < 3; k++) F.key[k] = (uint16_t)((((uint32_t)trns[2 * k] << 8) | trns[2 * k + 1]) & (depth == 16 ? 0xffffu : 0xffu));
    }
    return true;
}

#endif
