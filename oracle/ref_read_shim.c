/*
 * ref_read_shim.c -- TEST INFRASTRUCTURE: a few lines of our own around the REAL reference reader, so that Python can ask
 * "what RGBA8 does rwpng_read_image24 (/root/reference/src/rwpng.c:422, :179-400) make of this PNG file?".  oracle/Makefile
 * compiles it together with the reference's rwpng.c where that lies (nothing of the reference is copied); the result,
 * oracle/_ref/librwpng_ref.so, only exists in the build container and is used by tests/golden/make_png_read_golden.py to
 * produce the committed fixtures of the read side.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rwpng.h"

/* returns the reference's error code; on success *rgba is malloc'ed width*height*4 bytes (caller frees with ref_read_free) */
int ref_read_rgba(const char *path, unsigned char **rgba, unsigned *width, unsigned *height)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    png24_image img;
    memset(&img, 0, sizeof img);
    const pngloss_error rc = rwpng_read_image24(fp, &img, true, false);
    fclose(fp);
    if (rc != SUCCESS) return (int)rc;
    *width = img.width; *height = img.height;
    *rgba = malloc((size_t)img.width * img.height * 4);
    for (unsigned y = 0; y < img.height; y++) memcpy(*rgba + (size_t)y * img.width * 4, img.row_pointers[y], (size_t)img.width * 4);
    rwpng_free_image24(&img);
    return 0;
}

void ref_read_free(unsigned char *p) { free(p); }
