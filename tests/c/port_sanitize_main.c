/*
 * port_sanitize_main.c -- TEST INFRASTRUCTURE: the CPU restatement (oracle/pngloss_port.c), every chain variant, under
 * -fsanitize=address,undefined (the checker every GPU test trusts must itself be clean).  Built and run by
 * tests/test_oracle.py::test_restatement_is_clean_under_asan_and_ubsan; prints a digest per case so the test can also compare
 * the sanitized build with the ordinary one.
 */
#include "../../oracle/pngloss_port.c"

extern void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame);

static uint64_t fnv(const unsigned char *p, size_t n) { uint64_t h = 0xcbf29ce484222325ull; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; } return h; }

int main(void)
{
    static const struct { uint32_t w, h; int mode; unsigned s; long b; int filters; } cases[] = {
        { 64, 48, 0, 19, 2, 1 }, { 64, 48, 1, 19, 2, 1 }, { 64, 48, 2, 19, 2, 1 }, { 64, 48, 3, 19, 2, 1 }, { 64, 48, 4, 19, 2, 1 }, { 64, 48, 5, 19, 2, 1 },
        { 130, 9, 0, 0, 2, 1 }, { 130, 9, 1, 255, 1, 1 }, { 97, 13, 5, 85, 8, 0 }, { 1, 1, 1, 19, 2, 1 }, { 2, 3, 1, 19, 2, 1 }, { 5, 1, 1, 19, 2, 1 }, { 1, 7, 1, 19, 2, 1 },
        { 200, 20, 0, 40, 32767, 1 }, { 33, 33, 3, 7, 3, 0 },
    };
    int bad = 0;
    for (size_t i = 0; i < sizeof cases / sizeof cases[0]; i++) {
        const uint32_t w = cases[i].w, h = cases[i].h;
        uint64_t d0 = 0, f0 = 0;
        for (int variant = 0; variant < 3; variant++) {
            unsigned char *img = malloc((size_t)w * h * 4), *filt = malloc(h);
            unsigned char **rows = malloc(h * sizeof *rows);
            pngloss_synth_rgba(img, w, h, cases[i].mode, i);
            for (uint32_t y = 0; y < h; y++) rows[y] = img + (size_t)y * w * 4;
            memset(filt, 0, h);
            port_set_chain_variant(variant);
            if (port_optimize_with_rows(rows, w, h, cases[i].filters ? filt : NULL, false, (uint_fast8_t)cases[i].s, cases[i].b) != 0) bad++;
            const uint64_t d = fnv(img, (size_t)w * h * 4), f = fnv(filt, h);
            if (variant == 0) { d0 = d; f0 = f; } else if (d != d0 || f != f0) { bad++; fprintf(stderr, "variant %d differs in case %zu\n", variant, i); }
            free(rows); free(filt); free(img);
        }
        printf("%zu %016llx %016llx\n", i, (unsigned long long)d0, (unsigned long long)f0);
    }
    return bad ? 1 : 0;
}
