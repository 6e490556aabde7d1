/* png_stream_reader.h -- the host half of the device PNG reader (--gpu-read): chunk walk + zlib inflate, no libpng. */
#ifndef PNG_STREAM_READER_H
#define PNG_STREAM_READER_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint32_t width, height;
    uint8_t color_type, bit_depth;
    unsigned char *scanlines;      /* malloc'ed: height * (1 + rowbytes) inflated bytes */
    size_t scanline_bytes;
    unsigned char palette[768];
    uint32_t palette_entries;
    unsigned char trns[256];
    uint32_t trns_bytes;
    bool has_trns, has_srgb, has_gama;
    double gamma;                  /* gAMA / 100000 when has_gama */
    size_t file_size;
} png_stream_source;

/* Reads the file and inflates its image data.  true: `out` is filled (free out->scanlines).  false: the file is not one this
 * path takes -- interlaced, chunks beyond IHDR PLTE tRNS gAMA sRGB IDAT IEND (their handling is libpng's), damaged -- and the
 * caller reads it with libpng instead (which also produces the error messages for damaged files). */
bool png_stream_read(const char *path, png_stream_source *out);

#endif
