/*
 * png_bridge.h -- PNG <-> RGBA8 bridge of the pngloss command line tool (host C over libpng).
 *
 * It keeps the NAMES of the reference's I/O surface (/root/reference/src/rwpng.h: codes :23-38, image record :62-75,
 * prototypes :79-87) so that code written against that surface keeps compiling and behaves the same -- pngloss_error,
 * png24_image, rwpng_read_image24(), rwpng_write_image24(), rwpng_free_image24(), rwpng_version_info() -- while the
 * implementation in png_bridge.c is new.
 */
#ifndef PNGLOSS_AMD_PNG_BRIDGE_H
#define PNGLOSS_AMD_PNG_BRIDGE_H

#include <setjmp.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Process exit / status codes.  The numbers are part of the tool's contract with its callers (the web front-end
 * branches on 98/99), so they are pinned one by one. */
#define PNGLOSS_STATUS_TABLE(X)                                                                                    \
    X(SUCCESS, 0)                 /* all good                                                                  */ \
    X(MISSING_ARGUMENT, 1)        /* no input given                                                            */ \
    X(READ_ERROR, 2)              /* input cannot be opened                                                    */ \
    X(INVALID_ARGUMENT, 4)        /* bad option or option value                                                */ \
    X(NOT_OVERWRITING_ERROR, 15)  /* output exists and --force was not given                                   */ \
    X(CANT_WRITE_ERROR, 16)       /* output cannot be created / renamed                                        */ \
    X(OUT_OF_MEMORY_ERROR, 17)    /* host or device allocation failed                                          */ \
    X(WRONG_ARCHITECTURE, 18)     /* kept for numbering compatibility (unused here)                            */ \
    X(PNG_OUT_OF_MEMORY_ERROR, 24)/* libpng could not allocate                                                 */ \
    X(LIBPNG_FATAL_ERROR, 25)     /* libpng reported a fatal decoding error                                    */ \
    X(WRONG_INPUT_COLOR_TYPE, 26) /* kept for numbering compatibility (unused here)                            */ \
    X(LIBPNG_INIT_ERROR, 35)      /* libpng writer could not be set up                                         */ \
    X(TOO_LARGE_FILE, 98)         /* --skip-if-larger refused the result                                       */ \
    X(TOO_LOW_QUALITY, 99)        /* kept for numbering compatibility (unused here)                            */

typedef enum {
#define X(name, value) name = value,
    PNGLOSS_STATUS_TABLE(X)
#undef X
} pngloss_error;

/* One ancillary chunk carried from the input file to the output file (singly linked, newest first). */
struct rwpng_chunk {
    struct rwpng_chunk *next; /* following chunk or NULL                         */
    unsigned char *data;      /* payload (malloc'ed) or NULL when size == 0      */
    size_t size;              /* payload bytes                                   */
    unsigned char name[5];    /* four-character type + NUL                       */
    unsigned char location;   /* libpng's PNG_HAVE_* position flags              */
};

/* How the colour tags of the input are carried to the output. */
typedef enum {
    RWPNG_NONE,           /* no usable gamma information                         */
    RWPNG_SRGB,           /* sRGB chunk present: keep tagging sRGB               */
    RWPNG_ICCP,           /* (colour-managed builds only)                        */
    RWPNG_ICCP_WARN_GRAY, /* (colour-managed builds only)                        */
    RWPNG_GAMA_CHRM,      /* (colour-managed builds only)                        */
    RWPNG_GAMA_ONLY,      /* plain gamma value, passed through implicitly        */
    RWPNG_COCOA,          /* (macOS builds only)                                 */
} rwpng_color_transform;

/* A decoded image: always 8-bit RGBA, one row pointer per scanline. */
typedef struct {
    jmp_buf jmpbuf;                      /* libpng error landing pad                              */
    uint32_t width, height;              /* pixels                                                */
    size_t file_size;                    /* bytes read (after decoding) / written (after encoding) */
    size_t maximum_file_size;            /* 0 = unlimited; otherwise encoding fails with TOO_LARGE_FILE beyond it */
    size_t metadata_size;                /* bytes of ancillary chunks written                     */
    double gamma;                        /* file gamma (0.45455 = sRGB-ish default)               */
    unsigned char **row_pointers;        /* height pointers into rgba_data                        */
    unsigned char *rgba_data;            /* width*height*4 bytes                                  */
    struct rwpng_chunk *chunks;          /* ancillary chunks to pass through                      */
    rwpng_color_transform input_color;   /* what the reader found                                 */
    rwpng_color_transform output_color;  /* what the writer should tag                            */
} png24_image;

/* one line describing the libpng in use */
void rwpng_version_info(FILE *fp);
/* any PNG -> RGBA8 rows; `strip` drops ancillary chunks; libpng warnings go to stderr only when `verbose` */
pngloss_error rwpng_read_image24(FILE *infile, png24_image *image, bool strip, bool verbose);
/* RGBA8 rows -> PNG: colour type chosen from the pixels, per-row filter flags (NULL = libpng's heuristic), zlib 9 */
pngloss_error rwpng_write_image24(FILE *outfile, png24_image *image, unsigned char *row_filters);
/* releases rows, pixels and chunk list (safe on a zeroed record) */
void rwpng_free_image24(png24_image *image);

#ifdef __cplusplus
}
#endif
#endif
