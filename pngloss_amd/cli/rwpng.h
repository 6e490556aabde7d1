/*
 * rwpng.h -- PNG <-> RGBA8 bridge of the pngloss command line tool (host C, libpng).
 *
 * Same public surface as the reference's /root/reference/src/rwpng.h (types :23-75, prototypes :79-87) so that code
 * written against the reference keeps compiling: pngloss_error, png24_image, rwpng_read_image24(),
 * rwpng_write_image24(), rwpng_free_image24(), rwpng_version_info().  The implementation (rwpng.c) is new.
 */
#ifndef PNGLOSS_AMD_RWPNG_H
#define PNGLOSS_AMD_RWPNG_H

#include <setjmp.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exit / status codes, numerically identical to rwpng.h:23-38 */
typedef enum {
    SUCCESS = 0,
    MISSING_ARGUMENT = 1,
    READ_ERROR = 2,
    INVALID_ARGUMENT = 4,
    NOT_OVERWRITING_ERROR = 15,
    CANT_WRITE_ERROR = 16,
    OUT_OF_MEMORY_ERROR = 17,
    WRONG_ARCHITECTURE = 18,
    PNG_OUT_OF_MEMORY_ERROR = 24,
    LIBPNG_FATAL_ERROR = 25,
    WRONG_INPUT_COLOR_TYPE = 26,
    LIBPNG_INIT_ERROR = 35,
    TOO_LARGE_FILE = 98,
    TOO_LOW_QUALITY = 99,
} pngloss_error;

typedef struct rwpng_rgba { unsigned char r, g, b, a; } rwpng_rgba;

/* ancillary chunk carried from the input file to the output file */
struct rwpng_chunk {
    struct rwpng_chunk *next;
    unsigned char *data;
    size_t size;
    unsigned char name[5];
    unsigned char location;
};

typedef enum {
    RWPNG_NONE,
    RWPNG_SRGB,
    RWPNG_ICCP,
    RWPNG_ICCP_WARN_GRAY,
    RWPNG_GAMA_CHRM,
    RWPNG_GAMA_ONLY,
    RWPNG_COCOA,
} rwpng_color_transform;

typedef struct {
    jmp_buf jmpbuf;
    uint32_t width;
    uint32_t height;
    size_t file_size;
    size_t maximum_file_size;
    size_t metadata_size;
    double gamma;
    unsigned char **row_pointers;
    unsigned char *rgba_data;
    struct rwpng_chunk *chunks;
    rwpng_color_transform input_color;
    rwpng_color_transform output_color;
} png24_image;

void rwpng_version_info(FILE *fp);
/* any PNG -> RGBA8 rows (rwpng.c:179-400 of the reference); strip drops ancillary chunks */
pngloss_error rwpng_read_image24(FILE *infile, png24_image *image, bool strip, bool verbose);
/* RGBA8 rows -> PNG with colour type chosen from the pixels, explicit per-row filters (NULL: libpng's heuristic),
 * zlib level 9 (rwpng.c:477-637 of the reference) */
pngloss_error rwpng_write_image24(FILE *outfile, png24_image *image, unsigned char *row_filters);
void rwpng_free_image24(png24_image *image);

#ifdef __cplusplus
}
#endif
#endif
