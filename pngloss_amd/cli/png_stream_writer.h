/* png_stream_writer.h -- PNG container + zlib around scanlines that were filtered elsewhere (on the GPU). */
#ifndef PNGLOSS_AMD_PNG_STREAM_WRITER_H
#define PNGLOSS_AMD_PNG_STREAM_WRITER_H

#include "png_bridge.h"

#ifdef __cplusplus
extern "C" {
#endif

/* chunk position flags as libpng records them while reading (png.h PNG_HAVE_IHDR / PNG_HAVE_PLTE / PNG_AFTER_IDAT) */
#define PNG_STREAM_HAVE_IHDR 0x01
#define PNG_STREAM_HAVE_PLTE 0x02
#define PNG_STREAM_AFTER_IDAT 0x08

typedef struct {
    uint32_t width, height;
    int color_type;                    /* 0 gray, 4 gray+alpha, 2 RGB, 6 RGBA; always 8 bits per sample         */
    const unsigned char *filter_ids;   /* [height] PNG filter type 0..4 of every scanline                       */
    const unsigned char *rows;         /* filtered scanline bytes, width*channels per row ...                   */
    size_t pitch;                      /* ... `pitch` bytes apart                                               */
    double gamma;                      /* written as gAMA when tag_gamma                                        */
    bool tag_gamma, tag_srgb;          /* what rwpng.c:508-516 of the reference derives from output_color       */
    const struct rwpng_chunk *chunks;  /* ancillary chunks to pass through (list order = write order)           */
    size_t maximum_file_size;          /* 0 = unlimited, else TOO_LARGE_FILE beyond it                          */
    const unsigned char *zdata;        /* optional: the finished zlib stream of the scanlines (GPU deflate); then    */
    size_t zsize;                      /* filter_ids/rows are not read and no zlib runs here                          */
} png_stream_image;

pngloss_error png_stream_write(FILE *out, const png_stream_image *image, size_t *bytes_written, size_t *metadata_bytes);

#ifdef __cplusplus
}
#endif
#endif
