/*
 * pngloss_synth.c -- deterministic synthetic RGBA8 frames + FNV-1a-64 digests.
 *
 * Host-side utility (plain C, no GPU).  The generator and digest are the ones SURVEY.md Appendix B pins the
 * reference's golden digests on, so that bench.py / tests can regenerate the exact BASELINE.json inputs
 * (4096x4096 "photo" frame 0, 1920x1080 frames 0..255, 8192x8192 ...) on any box without shipping image data.
 *
 * The reference itself has no generator (its suite is 11 PNG files, /root/reference/suite/); this file is new.
 */
#include "pngloss_synth.h"

static inline uint32_t draw(uint64_t *s)
{
    uint64_t v = *s;
    v ^= v << 13;
    v ^= v >> 7;
    v ^= v << 17;
    *s = v;
    return (uint32_t)(v >> 32);
}

static inline unsigned char sat255(uint32_t v) { return (unsigned char)(v > 255u ? 255u : v); }

/* mode 0 photo-like ramps + 3-bit noise, alpha 255-((x^y)&31)   -> 4 B/px class
 * mode 1 uniform noise in all four channels                       -> 4 B/px class
 * mode 2 mode 0 with A=255                                        -> 3 B/px class (RGB)
 * mode 3 mode 0 with R=B=G                                        -> 2 B/px class (gray+alpha)
 * mode 4 mode 0 with R=B=G, A=255                                 -> 1 B/px class (gray)
 * mode 5 mode 0 with A=0 on an 8x8 checkerboard                   -> 4 B/px, fully transparent pixels */
void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame)
{
    uint64_t s = 0x9E3779B97F4A7C15ull + frame * 0xD1B54A32D192ED03ull;
    const uint64_t W = width, H = height;
    for (uint64_t y = 0; y < H; y++) {
        for (uint64_t x = 0; x < W; x++) {
            unsigned char *p = rgba + (y * W + x) * 4;
            uint32_t r = draw(&s);
            if (mode == 1) {
                p[0] = (unsigned char)(r & 255u);
                p[1] = (unsigned char)((r >> 8) & 255u);
                p[2] = (unsigned char)((r >> 16) & 255u);
                p[3] = (unsigned char)(r >> 24);
                continue;
            }
            p[0] = sat255((uint32_t)(x * 255u / W) + (r & 7u));
            p[1] = sat255((uint32_t)(y * 255u / H) + ((r >> 3) & 7u));
            p[2] = sat255((uint32_t)((x + y) * 255u / (W + H)) + ((r >> 6) & 7u));
            p[3] = (unsigned char)(255u - (uint32_t)((x ^ y) & 31u));
            switch (mode) {
            case 2: p[3] = 255; break;
            case 3: p[0] = p[1]; p[2] = p[1]; break;
            case 4: p[0] = p[1]; p[2] = p[1]; p[3] = 255; break;
            case 5: if (((x / 8) + (y / 8)) & 1u) p[3] = 0; break;
            default: break;
            }
        }
    }
}

uint64_t pngloss_fnv1a64_seed(const unsigned char *data, size_t n, uint64_t h)
{
    for (size_t i = 0; i < n; i++) {
        h ^= data[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

uint64_t pngloss_fnv1a64(const unsigned char *data, size_t n)
{
    return pngloss_fnv1a64_seed(data, n, 0xcbf29ce484222325ull);
}
