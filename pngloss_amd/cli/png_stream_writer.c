/*
 * png_stream_writer.c -- writes a PNG from scanlines that are ALREADY filtered (filter-type byte + filtered bytes per
 * row), i.e. from what the GPU emits, without libpng.
 *
 * First step of the PNG write side (/root/reference/src/rwpng.c:477-637 does this through
 * libpng's png_write_row): the per-row filtering moves to the device, the host only deflates and frames chunks.  The
 * container is produced exactly the way libpng 1.6 would produce it for the same image -- chunk order, gAMA/sRGB
 * tags, pass-through of ancillary chunks by location, zlib parameters (level 9, memLevel 9, Z_FILTERED, window sized
 * to the data), the CMF window fix-up of the first IDAT and the 8192-byte IDAT slicing -- so the files are
 * byte-identical to the reference tool's (tests/test_cli_host.py compares against libpng on every colour type).
 */
#include "png_stream_writer.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define IDAT_SLICE 8192u          /* libpng's default zbuffer_size */

typedef struct { FILE *fp; size_t total; pngloss_error status; } sink;

static void put(sink *s, const void *p, size_t n)
{
    if (s->status != SUCCESS || !n) return;
    if (!fwrite(p, n, 1, s->fp)) s->status = CANT_WRITE_ERROR;
    s->total += n;
}

static void be32(unsigned char *d, uint32_t v) { d[0] = (unsigned char)(v >> 24); d[1] = (unsigned char)(v >> 16); d[2] = (unsigned char)(v >> 8); d[3] = (unsigned char)v; }

static void put_chunk(sink *s, const char type[4], const unsigned char *data, size_t n)
{
    unsigned char head[8], tail[4];
    be32(head, (uint32_t)n);
    memcpy(head + 4, type, 4);
    uLong crc = crc32(0L, head + 4, 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    be32(tail, (uint32_t)crc);
    put(s, head, 8);
    put(s, data, n);
    put(s, tail, 4);
}

/* libpng's check_location(): of the position flags recorded while reading, the LAST position wins */
static int top_location(int loc)
{
    loc &= PNG_STREAM_HAVE_IHDR | PNG_STREAM_HAVE_PLTE | PNG_STREAM_AFTER_IDAT;
    while (loc != (loc & -loc)) loc &= ~(loc & -loc);
    return loc;
}

static void put_passthrough(sink *s, const struct rwpng_chunk *list, int where, size_t *meta)
{
    for (const struct rwpng_chunk *c = list; c; c = c->next) {
        if (top_location(c->location) != where) continue;
        if (!(c->name[3] & 0x20)) continue;              /* not safe-to-copy: libpng's default policy drops it */
        put_chunk(s, (const char *)c->name, c->data, c->size);
        if (meta) *meta += c->size + 12;
    }
}

/* zlib window for `bytes` of data the way png_deflate_claim() picks it (only ever shrinks for <= 16384 bytes) */
static int window_bits_for(size_t bytes)
{
    int bits = 15;
    if (bytes <= 16384) {
        unsigned half = 1u << (bits - 1);
        while (bytes + 262 <= half) { half >>= 1; --bits; }
    }
    return bits;
}

/* libpng's optimize_cmf(): shrink the window advertised in the zlib header to the smallest that covers the data */
static void fix_cmf(unsigned char *z, size_t data_size)
{
    if (data_size > 16384) return;
    unsigned z_cmf = z[0];
    if ((z_cmf & 0x0f) != 8 || (z_cmf & 0xf0) > 0x70) return;
    unsigned z_cinfo = z_cmf >> 4;
    unsigned half = 1u << (z_cinfo + 7);
    if (data_size > half) return;
    do { half >>= 1; --z_cinfo; } while (z_cinfo > 0 && data_size <= half);
    z_cmf = (z_cmf & 0x0f) | (z_cinfo << 4);
    z[0] = (unsigned char)z_cmf;
    unsigned tmp = z[1] & 0xe0;
    tmp += 0x1f - ((z_cmf << 8) + tmp) % 0x1f;
    z[1] = (unsigned char)tmp;
}

/* IDAT the way libpng writes it: one zlib stream over [filter byte, filtered row] x height, cut into 8192-byte chunks */
static pngloss_error put_idat_zlib(sink *sp, const png_stream_image *im, size_t rowbytes, size_t data_size)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 9, Z_DEFLATED, window_bits_for(data_size), 9, Z_FILTERED) != Z_OK) return LIBPNG_INIT_ERROR;
    unsigned char *buf = malloc(IDAT_SLICE);
    if (!buf) { deflateEnd(&z); return OUT_OF_MEMORY_ERROR; }
    z.next_out = buf;
    z.avail_out = IDAT_SLICE;
    bool first = true;
    pngloss_error rc = SUCCESS;
    for (uint32_t y = 0; y <= im->height && rc == SUCCESS; y++) {
        const bool last = y == im->height;
        /* each row goes in as two pieces (type byte, bytes); deflate's output does not depend on how input is sliced */
        for (int piece = 0; piece < (last ? 1 : 2) && rc == SUCCESS; piece++) {
            unsigned char type_byte;
            if (!last) {
                if (piece == 0) { type_byte = im->filter_ids[y]; z.next_in = &type_byte; z.avail_in = 1; }
                else { z.next_in = (Bytef *)(im->rows + (size_t)y * im->pitch); z.avail_in = (uInt)rowbytes; }
            }
            for (;;) {
                const int zr = deflate(&z, last ? Z_FINISH : Z_NO_FLUSH);
                if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) { rc = LIBPNG_INIT_ERROR; break; }
                if (z.avail_out == 0 || zr == Z_STREAM_END) {
                    const size_t n = IDAT_SLICE - z.avail_out;
                    if (n) {
                        if (first) { fix_cmf(buf, data_size); first = false; }
                        put_chunk(sp, "IDAT", buf, n);
                    }
                    z.next_out = buf;
                    z.avail_out = IDAT_SLICE;
                }
                if (zr == Z_STREAM_END) break;
                if (!last && z.avail_in == 0) break;
            }
        }
    }
    deflateEnd(&z);
    free(buf);
    return rc;
}

pngloss_error png_stream_write(FILE *out, const png_stream_image *im, size_t *bytes_written, size_t *metadata_bytes)
{
    static const unsigned char signature[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    const unsigned channels = im->color_type == 0 ? 1 : (im->color_type == 4 ? 2 : (im->color_type == 2 ? 3 : 4));
    const size_t rowbytes = (size_t)im->width * channels;
    const size_t data_size = (rowbytes + 1) * im->height;
    sink s = { out, 0, SUCCESS };
    size_t meta = 0;

    put(&s, signature, 8);
    unsigned char ihdr[13];
    be32(ihdr, im->width);
    be32(ihdr + 4, im->height);
    ihdr[8] = 8; ihdr[9] = (unsigned char)im->color_type; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    put_chunk(&s, "IHDR", ihdr, 13);
    if (im->tag_gamma) {
        unsigned char g[4];
        be32(g, (uint32_t)(im->gamma * 100000.0 + 0.5));
        put_chunk(&s, "gAMA", g, 4);
    }
    if (im->tag_srgb) {
        const unsigned char intent = 0;
        put_chunk(&s, "sRGB", &intent, 1);
    }
    put_passthrough(&s, im->chunks, PNG_STREAM_HAVE_IHDR, &meta);
    put_passthrough(&s, im->chunks, PNG_STREAM_HAVE_PLTE, &meta);

    if (im->zdata) {
        /* ---- IDAT from a zlib stream that was compressed on the GPU: framing only (a chunk holds < 2^31 bytes) ---- */
        const size_t slice = (size_t)1 << 30;
        for (size_t off = 0; off < im->zsize; off += slice)
            put_chunk(&s, "IDAT", im->zdata + off, im->zsize - off < slice ? im->zsize - off : slice);
    } else {
        const pngloss_error rc = put_idat_zlib(&s, im, rowbytes, data_size);
        if (rc != SUCCESS) return rc;
    }

    put_passthrough(&s, im->chunks, PNG_STREAM_AFTER_IDAT, &meta);
    put_chunk(&s, "IEND", NULL, 0);
    if (s.status == SUCCESS && im->maximum_file_size && s.total > im->maximum_file_size) return TOO_LARGE_FILE;
    if (bytes_written) *bytes_written = s.total;
    if (metadata_bytes) *metadata_bytes = meta;
    return SUCCESS;       /* like the reference, a short write is not reported (rwpng.c:631-636 only checks the size cap) */
}
