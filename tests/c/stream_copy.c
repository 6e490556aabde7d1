/* tests/c/stream_copy.c -- test driver for png_stream_writer.c: decode a PNG (png_bridge), filter its scanlines HERE on
 * the CPU (test code standing in for the GPU emit kernel), write the file with png_stream_write, and -- for comparison
 * -- with libpng through rwpng_write_image24 using the same per-row filters.  Both files must be byte-identical.
 * usage: stream_copy in.png out_stream.png out_libpng.png policy strip      (policy as in rwpng_copy.c) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "png_bridge.h"
#include "png_stream_writer.h"

static int paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
static int predict(int f, int left, int above, int diag) { return f == 1 ? left : f == 2 ? above : f == 3 ? (left + above) >> 1 : f == 4 ? paeth(left, above, diag) : 0; }

int main(int argc, char **argv)
{
    if (argc < 6) return 1;
    const int policy = atoi(argv[4]);
    const bool strip = atoi(argv[5]) != 0;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 2;
    png24_image img;
    memset(&img, 0, sizeof img);
    pngloss_error rc = rwpng_read_image24(in, &img, strip, false);
    fclose(in);
    if (rc) return (int)rc;
    const uint32_t W = img.width, H = img.height;
    /* output colour type the way the writer side detects it */
    bool gray = true, opaque = true;
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            const unsigned char *p = img.row_pointers[y] + 4 * x;
            gray = gray && p[0] == p[1] && p[1] == p[2];
            opaque = opaque && p[3] == 255;
        }
    const int ch = gray ? (opaque ? 1 : 2) : (opaque ? 3 : 4);
    const int ctype = gray ? (opaque ? 0 : 4) : (opaque ? 2 : 6);
    const size_t rb = (size_t)W * ch;
    unsigned char *raw = calloc(H ? H : 1, rb ? rb : 1), *filt = calloc(H ? H : 1, rb ? rb : 1), *ids = calloc(H ? H : 1, 1), *flags = NULL;
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            const unsigned char *p = img.row_pointers[y] + 4 * x;
            unsigned char *d = raw + y * rb + (size_t)x * ch;
            if (ch == 1) d[0] = p[1]; else if (ch == 2) { d[0] = p[1]; d[1] = p[3]; } else if (ch == 3) { d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; } else memcpy(d, p, 4);
        }
    static const unsigned char flag[5] = { 0x08, 0x10, 0x20, 0x40, 0x80 };
    if (policy >= 0) flags = malloc(H ? H : 1);
    for (uint32_t y = 0; y < H; y++) {
        const unsigned char *row = raw + y * rb, *up = y ? row - rb : NULL;
        int f;
        /* libpng (png_write_start_row / png_set_filter) drops the filters that have no neighbour to predict from: sub, average
         * and paeth in 1-pixel-wide images, up, average and paeth in 1-pixel-high ones; a row asked to use one gets "none" */
        const int allowed = (W == 1 ? 0x05 : 0x1f) & (H == 1 ? 0x03 : 0x1f);      /* bit g = filter g may be used */
        if (policy < 0 || y == 0) {          /* libpng's heuristic: least sum of |signed residual|, first minimum wins */
            unsigned long best = ~0ul; f = 0;
            for (int g = 0; g < 5; g++) {
                if (!((allowed >> g) & 1)) continue;
                unsigned long sum = 0;
                for (size_t i = 0; i < rb; i++) {
                    int v = (row[i] - predict(g, i >= (size_t)ch ? row[i - ch] : 0, up ? up[i] : 0, (up && i >= (size_t)ch) ? up[i - ch] : 0)) & 255;
                    sum += v < 128 ? v : 256 - v;
                }
                if (sum < best) { best = sum; f = g; }
            }
        } else f = policy == 5 ? (int)(y % 5) : policy;
        if (!((allowed >> f) & 1)) f = 0;
        ids[y] = (unsigned char)f;
        if (flags) flags[y] = flag[policy == 5 ? y % 5 : policy];
        for (size_t i = 0; i < rb; i++)
            filt[y * rb + i] = (unsigned char)(row[i] - predict(f, i >= (size_t)ch ? row[i - ch] : 0, up ? up[i] : 0, (up && i >= (size_t)ch) ? up[i - ch] : 0));
    }
    png_stream_image si = { W, H, ctype, ids, filt, rb, img.gamma,
                            img.output_color != RWPNG_GAMA_ONLY && img.output_color != RWPNG_NONE, img.output_color == RWPNG_SRGB, img.chunks, 0 };
    FILE *o1 = fopen(argv[2], "wb");
    size_t n1 = 0, m1 = 0;
    rc = png_stream_write(o1, &si, &n1, &m1);
    fclose(o1);
    if (rc) return 40 + (int)rc;
    FILE *o2 = fopen(argv[3], "wb");
    rc = rwpng_write_image24(o2, &img, flags);
    fclose(o2);
    printf("%zu %zu %zu %zu\n", n1, img.file_size, m1, img.metadata_size);
    return (int)rc;
}
