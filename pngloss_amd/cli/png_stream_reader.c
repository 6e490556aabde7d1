/*
 * png_stream_reader.c -- the host half of the device PNG reader (pngloss --gpu-read).
 *
 * What rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400) gets from libpng, split in two: here the container format
 * (signature, chunk walk with CRC check, IHDR / PLTE / tRNS / gAMA / sRGB) and the inflate of the concatenated IDAT data with zlib
 * -- a serial bit stream per file, so it stays on the host, one file per decode thread; on the device
 * (pngloss_hip_png_decode_batch_host) the inverse filters and the expansion to RGBA8.  Files this reader does not take -- Adam7
 * interlace, any other chunk (text, ICC profiles, physical size ...: what libpng does with them is libpng's business), anything
 * damaged -- go through libpng as before.
 */
#include "png_stream_reader.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool png_stream_read(const char *path, png_stream_source *out)
{
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    unsigned char *file = NULL, *idat = NULL;
    bool ok = false;
    do {
        if (fseek(f, 0, SEEK_END) != 0) break;
        const long len = ftell(f);
        if (len < 8 + 25 + 12 || fseek(f, 0, SEEK_SET) != 0) break;
        file = malloc((size_t)len);
        if (!file || fread(file, 1, (size_t)len, f) != (size_t)len) break;
        static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
        if (memcmp(file, sig, 8) != 0) break;
        idat = malloc((size_t)len);
        if (!idat) break;
        size_t idat_len = 0, o = 8;
        bool seen_ihdr = false, seen_iend = false, bad = false, seen_idat = false, idat_closed = false;
        while (o + 12 <= (size_t)len && !seen_iend && !bad) {
            const uint32_t n = be32(file + o);
            const unsigned char *tag = file + o + 4, *body = file + o + 8;
            if (n > 0x7fffffffu || o + 12 + (size_t)n > (size_t)len) { bad = true; break; }
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), tag, 4 + n) != be32(body + n)) { bad = true; break; }
            if (!seen_ihdr && memcmp(tag, "IHDR", 4) != 0) { bad = true; break; }
            if (memcmp(tag, "IHDR", 4) == 0) {
                if (seen_ihdr || n != 13) { bad = true; break; }
                seen_ihdr = true;
                out->width = be32(body); out->height = be32(body + 4);
                out->bit_depth = body[8]; out->color_type = body[9];
                if (body[10] != 0 || body[11] != 0 || body[12] != 0) { bad = true; break; }      /* interlace (Adam7): libpng's */
                if (!out->width || !out->height || out->width > 0x7fffffffu / 8 || out->height > 0x7fffffffu / 8) { bad = true; break; }
                /* libpng's own user limits (1,000,000 each way) and the reference's size check (rwpng.c:287: rowbytes > INT_MAX / height) stay
                 * libpng's and the reference's to report */
                if (out->width > 1000000u || out->height > 1000000u || (uint64_t)out->width * 4u > (uint64_t)0x7fffffff / out->height) { bad = true; break; }
            } else if (memcmp(tag, "PLTE", 4) == 0) {
                if (seen_idat || n % 3 || n > 768 || out->palette_entries) { bad = true; break; }
                memcpy(out->palette, body, n); out->palette_entries = n / 3;
            } else if (memcmp(tag, "tRNS", 4) == 0) {
                if (seen_idat || n > 256 || out->has_trns) { bad = true; break; }
                /* libpng discards as benign errors: a tRNS in front of the PLTE of a palette image, one with more entries than the palette,
                 * one of the wrong length for gray (2) or RGB (6) -- such files are read by libpng, which knows what it does with them */
                if (out->color_type == 3 && (!out->palette_entries || n > out->palette_entries)) { bad = true; break; }
                if ((out->color_type == 0 && n != 2) || (out->color_type == 2 && n != 6)) { bad = true; break; }
                memcpy(out->trns, body, n); out->trns_bytes = n; out->has_trns = true;
            } else if (memcmp(tag, "gAMA", 4) == 0) {
                if (seen_idat || n != 4 || out->has_gama) { bad = true; break; }
                /* (a gAMA behind the PLTE, or of value 0, is ignored by libpng: the reference then keeps its default tag) */
                if (out->palette_entries || be32(body) == 0) { bad = true; break; }
                out->has_gama = true; out->gamma = be32(body) / 100000.0;
            } else if (memcmp(tag, "sRGB", 4) == 0) {
                if (seen_idat || n != 1) { bad = true; break; }
                out->has_srgb = true;
            } else if (memcmp(tag, "IDAT", 4) == 0) {
                if (idat_closed) { bad = true; break; }
                seen_idat = true;
                memcpy(idat + idat_len, body, n); idat_len += n;
            } else if (memcmp(tag, "IEND", 4) == 0) {
                seen_iend = true;
            } else { bad = true; break; }                                                       /* any other chunk: libpng's */
            if (seen_idat && memcmp(tag, "IDAT", 4) != 0) idat_closed = true;
            o += 12 + (size_t)n;
        }
        if (bad || !seen_iend || !seen_idat) break;
        /* the formats PNG allows */
        const int ct = out->color_type, d = out->bit_depth;
        const int channels = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : ct == 6 ? 4 : 0;
        const bool fmt = (ct == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (ct == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                         ((ct == 2 || ct == 4 || ct == 6) && (d == 8 || d == 16));
        if (!fmt || (ct == 3 && !out->palette_entries)) break;
        if (out->has_trns && (ct == 4 || ct == 6)) break;                                       /* not allowed; leave the complaint to libpng */
        const size_t rowbytes = ((size_t)out->width * (size_t)(channels * d) + 7) / 8;
        const size_t want = (rowbytes + 1) * (size_t)out->height;
        out->scanlines = malloc(want ? want : 1);
        if (!out->scanlines) break;
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit(&z) != Z_OK) break;
        z.next_in = idat; z.avail_in = (uInt)idat_len;
        z.next_out = out->scanlines; z.avail_out = (uInt)want;
        const int zr = inflate(&z, Z_FINISH);
        const size_t got = want - z.avail_out;
        inflateEnd(&z);
        if (zr != Z_STREAM_END || got != want || idat_len > 0xffffffffu || want > 0xffffffffu) break;
        out->scanline_bytes = want;
        out->file_size = (size_t)len;
        ok = true;
    } while (0);
    fclose(f);
    free(file);
    free(idat);
    if (!ok) { free(out->scanlines); out->scanlines = NULL; }
    return ok;
}
