/*
 * pngread_host.cpp -- TEST INFRASTRUCTURE: the pixel arithmetic of the device PNG reader (pngloss_amd/csrc/pl_pngread_core.h, shared
 * with the HIP kernel) run on the CPU with a plain row-by-row unfilter loop, so that the CPU suite can check it against the
 * fixtures the real reference reader produced (tests/golden/png_read_cases.npz).  Never shipped.
 */
#include "../../pngloss_amd/csrc/pl_pngread_core.h"

#include <vector>

extern "C" int pngread_host_decode(const unsigned char *scanlines, uint32_t width, uint32_t height, int color_type, int depth,
                                   const unsigned char *plte, uint32_t plte_entries, const unsigned char *trns, uint32_t trns_bytes, unsigned char *rgba)
{
    PrFormat F;
    if (!pr_format(F, width, height, color_type, depth, plte, plte_entries, trns, trns_bytes)) return 4;
    std::vector<uint8_t> prev(F.rowbytes, 0), cur(F.rowbytes, 0);
    for (uint32_t y = 0; y < height; y++) {
        const unsigned char *src = scanlines + (size_t)y * (F.rowbytes + 1);
        const int ft = src[0];
        if (ft > 4) return 25;
        for (uint32_t i = 0; i < F.rowbytes; i++) {
            const int a = i >= F.bppf ? cur[i - F.bppf] : 0, b = prev[i], c = i >= F.bppf ? prev[i - F.bppf] : 0;
            cur[i] = (uint8_t)pr_recon(ft, src[1 + i], a, b, c);
        }
        for (uint32_t x = 0; x < width; x++) {
            const uint32_t v = pr_expand(F, cur.data(), x);
            unsigned char *d = rgba + ((size_t)y * width + x) * 4;
            d[0] = (unsigned char)v; d[1] = (unsigned char)(v >> 8); d[2] = (unsigned char)(v >> 16); d[3] = (unsigned char)(v >> 24);
        }
        prev.swap(cur);
    }
    return 0;
}

/* pr_recon4 (four bytes at once, branch-free) against pr_recon byte by byte: number of disagreements over n pseudo-random inputs */
extern "C" int pngread_host_recon4_check(uint32_t n, uint32_t seed)
{
    int bad = 0;
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t x = next(), a = (i & 7) == 0 ? next() & 0x01ff01ffu : next(), b = (i & 5) == 0 ? a : next(), c = (i & 3) == 0 ? b : next();
        for (int ft = 0; ft < 5; ft++) {
            uint32_t want = 0;
            for (int k = 0; k < 4; k++) want |= (uint32_t)pr_recon(ft, (x >> (8 * k)) & 255, (a >> (8 * k)) & 255, (b >> (8 * k)) & 255, (c >> (8 * k)) & 255) << (8 * k);
            bad += pr_recon4(ft, x, a, b, c) != want;
        }
    }
    return bad;
}
