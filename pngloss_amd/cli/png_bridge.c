/*
 * png_bridge.c -- libpng glue for the pngloss CLI: decode any PNG to RGBA8, encode RGBA8 with caller-chosen row filters.
 *
 * Written for this project; it performs the same sequence of libpng operations as the reference's rwpng.c
 * (/root/reference/src/rwpng.c:179-400 read, :444-637 write) so that, given the same libpng/zlib, the files it writes
 * are byte-identical to the reference's -- that equality is what tests/test_gpu_cli.py checks.
 */
#include "png_bridge.h"

#include <limits.h>
#include <png.h>
#include <stdlib.h>
#include <string.h>

#define SRGB_GAMMA 0.45455

/* ------------------------------------------------------------------------------------------------ common */

static void on_libpng_error(png_structp png, png_const_charp msg)
{
    png24_image *img = png_get_error_ptr(png);
    fprintf(stderr, "  error: %s (libpng failed)\n", msg);
    fflush(stderr);
    if (!img) abort();
    longjmp(img->jmpbuf, 1);
}

static void warn_loud(png_structp png, png_const_charp msg) { (void)png; fprintf(stderr, "  libpng warning: %s\n", msg); }
static void warn_mute(png_structp png, png_const_charp msg) { (void)png; (void)msg; }

void rwpng_version_info(FILE *fp)
{
    fprintf(fp, "   Compiled with no support for color profiles. Using libpng %s.\n", png_get_header_ver(NULL));
}

static unsigned char **make_row_table(unsigned char *base, size_t rows, size_t rowbytes)
{
    unsigned char **t = malloc(rows * sizeof *t);
    if (t)
        for (size_t y = 0; y < rows; y++) t[y] = base + y * rowbytes;
    return t;
}

static void free_chunk_list(struct rwpng_chunk *c)
{
    while (c) {
        struct rwpng_chunk *next = c->next;
        free(c->data);
        free(c);
        c = next;
    }
}

void rwpng_free_image24(png24_image *image)
{
    free(image->row_pointers);
    image->row_pointers = NULL;
    free(image->rgba_data);
    image->rgba_data = NULL;
    free_chunk_list(image->chunks);
    image->chunks = NULL;
}

/* ------------------------------------------------------------------------------------------------ reading */

typedef struct { FILE *fp; size_t total; } counted_source;

static void pull_bytes(png_structp png, png_bytep dst, png_size_t want)
{
    counted_source *src = png_get_io_ptr(png);
    size_t got = fread(dst, 1, want, src->fp);
    if (!got) png_error(png, "Read error");
    src->total += got;
}

/* libpng hands every chunk it does not interpret to this hook; colour-management chunks go back to libpng, everything
 * else with a valid position is remembered (newest first) for the writer */
static int remember_chunk(png_structp png, png_unknown_chunkp in)
{
    static const char *const left_to_libpng[] = { "iCCP", "cHRM", "gAMA" };
    for (size_t i = 0; i < sizeof left_to_libpng / sizeof left_to_libpng[0]; i++)
        if (memcmp(left_to_libpng[i], in->name, 5) == 0) return 0;
    if (in->location == 0) return 1;

    struct rwpng_chunk **head = png_get_user_chunk_ptr(png);
    struct rwpng_chunk *c = malloc(sizeof *c);
    if (!c) return 1;
    memcpy(c->name, in->name, 5);
    c->size = in->size;
    c->location = in->location;
    c->data = NULL;
    if (in->size) {
        c->data = malloc(in->size);
        if (c->data) memcpy(c->data, in->data, in->size);
    }
    c->next = *head;
    *head = c;
    return 1;
}

pngloss_error rwpng_read_image24(FILE *infile, png24_image *image, bool strip, bool verbose)
{
    png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, image, on_libpng_error, verbose ? warn_loud : warn_mute);
    if (!png) return PNG_OUT_OF_MEMORY_ERROR;
    png_infop info = png_create_info_struct(png);
    if (!info) {
        png_destroy_read_struct(&png, NULL, NULL);
        return PNG_OUT_OF_MEMORY_ERROR;
    }
    if (setjmp(image->jmpbuf)) {
        png_destroy_read_struct(&png, &info, NULL);
        return LIBPNG_FATAL_ERROR;
    }

#if defined(PNG_SKIP_sRGB_CHECK_PROFILE) && defined(PNG_SET_OPTION_SUPPORTED)
    png_set_option(png, PNG_SKIP_sRGB_CHECK_PROFILE, PNG_OPTION_ON);
#endif
    if (!strip) {
#if defined(PNG_UNKNOWN_CHUNKS_SUPPORTED)
        png_set_keep_unknown_chunks(png, PNG_HANDLE_CHUNK_IF_SAFE, (png_const_bytep) "pHYs\0iTXt\0tEXt\0zTXt", 4);
#endif
        png_set_read_user_chunk_fn(png, &image->chunks, remember_chunk);
    }

    counted_source src = { infile, 0 };
    png_set_read_fn(png, &src, pull_bytes);
    png_read_info(png, info);

    int depth, ctype;
    png_get_IHDR(png, info, &image->width, &image->height, &depth, &ctype, NULL, NULL, NULL);

    /* whatever comes in leaves as 8-bit RGBA */
    if (!(ctype & PNG_COLOR_MASK_ALPHA)) {
        png_set_expand(png);                       /* palette -> RGB, tRNS -> alpha, 1/2/4-bit gray -> 8 */
        png_set_filler(png, 65535L, PNG_FILLER_AFTER);
    }
    if (depth == 16) png_set_strip_16(png);
    if (!(ctype & PNG_COLOR_MASK_COLOR)) png_set_gray_to_rgb(png);

    double gamma = SRGB_GAMMA;
    if (png_get_valid(png, info, PNG_INFO_sRGB)) {
        image->input_color = image->output_color = RWPNG_SRGB;
    } else {
        png_get_gAMA(png, info, &gamma);
        if (gamma > 0 && gamma <= 1.0) {
            image->input_color = image->output_color = RWPNG_GAMA_ONLY;
        } else {
            fprintf(stderr, "pngloss readpng:  ignored out-of-range gamma %f\n", gamma);
            image->input_color = image->output_color = RWPNG_NONE;
            gamma = SRGB_GAMMA;
        }
    }
    image->gamma = gamma;

    png_set_interlace_handling(png);
    png_read_update_info(png, info);

    const png_size_t rowbytes = png_get_rowbytes(png, info);
    if (rowbytes > (png_size_t)INT_MAX / image->height) {     /* keep everything addressable with 32 bits */
        png_destroy_read_struct(&png, &info, NULL);
        return PNG_OUT_OF_MEMORY_ERROR;
    }
    image->rgba_data = malloc(rowbytes * image->height);
    if (!image->rgba_data) {
        fprintf(stderr, "pngloss readpng:  unable to allocate image data\n");
        png_destroy_read_struct(&png, &info, NULL);
        return PNG_OUT_OF_MEMORY_ERROR;
    }
    unsigned char **rows = make_row_table(image->rgba_data, image->height, rowbytes);
    if (!rows) {
        png_destroy_read_struct(&png, &info, NULL);
        return PNG_OUT_OF_MEMORY_ERROR;
    }
    image->row_pointers = rows;       /* owned by the image from here on (also on the longjmp path) */
    png_read_image(png, rows);
    png_read_end(png, NULL);
    png_destroy_read_struct(&png, &info, NULL);

    image->file_size = src.total;
    return SUCCESS;
}

/* ------------------------------------------------------------------------------------------------ writing */

typedef struct { FILE *fp; size_t total; pngloss_error status; } counted_sink;

static void push_bytes(png_structp png, png_bytep data, png_size_t n)
{
    counted_sink *dst = png_get_io_ptr(png);
    if (dst->status != SUCCESS) return;
    if (!fwrite(data, n, 1, dst->fp)) dst->status = CANT_WRITE_ERROR;
    dst->total += n;
}

static void flush_nothing(png_structp png) { (void)png; }

/* what rwpng.c:558-573 of the reference detects: can the pixels be stored as gray and/or without alpha? */
static void classify_pixels(const png24_image *image, bool *gray, bool *opaque)
{
    bool g = true, o = true;
    for (uint32_t y = 0; y < image->height && (g || o); y++) {
        const unsigned char *p = image->row_pointers[y];
        for (uint32_t x = 0; x < image->width; x++, p += 4) {
            g = g && p[0] == p[1] && p[1] == p[2];
            o = o && p[3] == 255;
        }
    }
    *gray = g;
    *opaque = o;
}

pngloss_error rwpng_write_image24(FILE *outfile, png24_image *image, unsigned char *row_filters)
{
    png_structp png = png_create_write_struct(PNG_LIBPNG_VER_STRING, image, on_libpng_error, NULL);
    if (!png) return LIBPNG_INIT_ERROR;
    png_infop info = png_create_info_struct(png);
    if (!info) {
        png_destroy_write_struct(&png, NULL);
        return LIBPNG_INIT_ERROR;
    }
    unsigned char *volatile gray_rows = NULL;
    unsigned char **volatile table = NULL;
    if (setjmp(image->jmpbuf)) {
        png_destroy_write_struct(&png, &info);
        free(gray_rows);
        free(table);
        return LIBPNG_INIT_ERROR;
    }
    png_set_compression_level(png, 9);
    png_set_compression_mem_level(png, 9);

    png_init_io(png, outfile);
    counted_sink sink = { outfile, 0, SUCCESS };
    png_set_write_fn(png, &sink, push_bytes, flush_nothing);

    /* colour tags */
    if (image->output_color != RWPNG_GAMA_ONLY && image->output_color != RWPNG_NONE) png_set_gAMA(png, info, image->gamma);
    if (image->output_color == RWPNG_SRGB) png_set_sRGB(png, info, 0);

    /* ancillary chunks remembered by the reader */
    image->metadata_size = 0;
    for (struct rwpng_chunk *c = image->chunks; c; c = c->next) {
        png_unknown_chunk u;
        memset(&u, 0, sizeof u);
        memcpy(u.name, c->name, 5);
        u.data = c->data;
        u.size = c->size;
        u.location = c->location;
        png_set_unknown_chunks(png, info, &u, 1);
        image->metadata_size += c->size + 12;
    }

    bool gray, opaque;
    classify_pixels(image, &gray, &opaque);
    const uint32_t W = image->width, H = image->height;
    if (gray) {
        /* libpng wants gray+alpha pairs; the gray value is the green channel */
        gray_rows = malloc((size_t)W * 2 * H);
        if (gray_rows) {
            for (uint32_t y = 0; y < H; y++) {
                const unsigned char *s = image->row_pointers[y];
                unsigned char *d = gray_rows + (size_t)y * W * 2;
                for (uint32_t x = 0; x < W; x++) { d[2 * x] = s[4 * x + 1]; d[2 * x + 1] = s[4 * x + 3]; }
            }
        } else {
            gray = false;
        }
    }
    const int ctype = gray ? (opaque ? PNG_COLOR_TYPE_GRAY : PNG_COLOR_TYPE_GRAY_ALPHA)
                           : (opaque ? PNG_COLOR_TYPE_RGB : PNG_COLOR_TYPE_RGB_ALPHA);
    png_set_IHDR(png, info, W, H, 8, ctype, 0, PNG_COMPRESSION_TYPE_DEFAULT, PNG_FILTER_TYPE_DEFAULT);

    table = gray ? make_row_table(gray_rows, H, (size_t)W * 2) : make_row_table(image->rgba_data, H, (size_t)W * 4);
    if (!table) {
        png_destroy_write_struct(&png, &info);
        free(gray_rows);
        return OUT_OF_MEMORY_ERROR;
    }

    png_write_info(png, info);
    if (opaque) png_set_filler(png, 0, PNG_FILLER_AFTER);       /* drop the alpha byte on the way out */
    png_set_packing(png);
    png_set_filter(png, PNG_FILTER_TYPE_BASE, PNG_ALL_FILTERS); /* PNG: the first row is always adaptive */
    if (row_filters) {
        png_write_row(png, table[0]);
        for (uint32_t y = 1; y < H; y++) {
            png_set_filter(png, PNG_FILTER_TYPE_BASE, row_filters[y]);
            png_write_row(png, table[y]);
        }
    } else {
        png_write_image(png, table);
    }
    png_write_end(png, NULL);
    png_destroy_write_struct(&png, &info);
    free(table);
    free(gray_rows);

    if (sink.status == SUCCESS && image->maximum_file_size && sink.total > image->maximum_file_size) return TOO_LARGE_FILE;
    image->file_size = sink.total;
    return SUCCESS;
}
