/*
 * deflate_host.cpp -- test driver: runs pngloss_amd/csrc/pl_deflate_core.h (the code the GPU kernels execute) serially
 * on the CPU so that the bitstream logic can be checked against zlib's inflate where there is no GPU.
 *
 *   size_t dfl_host_zlib(in, n, out, cap, max_chain, min_len, block_bytes, stats[4])   -> zlib stream, 0 on overflow
 *   built with -DDFL_MAIN: deflate_host FILE [max_chain min_len block_bytes]  prints sizes next to zlib level 9
 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include <zlib.h>

#include <pthread.h>
#include <thread>

#define DFL_COOP_MAX_BLOCK (1u << 22)          /* the tests also use blocks larger than the product's 256 KiB */
#include "../../pngloss_amd/csrc/pl_deflate_coop.h"

static uint32_t g_levels[12] = { 0 };
static int g_optimal = 1;
static int g_team = 0;              /* 0: one-thread dfl_encode_block; N >= 1: dfl_encode_block_coop with a team of N threads */
extern "C" void dfl_host_set_team(int n) { g_team = n; }

static void barrier_wait(void *b) { pthread_barrier_wait(static_cast<pthread_barrier_t *>(b)); }

static dfl_block_result encode_with_team(int nthreads, const uint8_t *in, const uint32_t *match, const uint32_t *near, const dfl_block_desc *d,
                                         const dfl_params *prm, uint32_t *tok, uint32_t *choice, uint8_t *out)
{
    static dfl_coop shared;                      /* the team's "LDS" */
    dfl_block_result res{};
    if (nthreads == 1) {
        dfl_team t = { 0, 1, nullptr, nullptr };
        return dfl_encode_block_coop(&t, in, match, near, d, prm, tok, choice, out, &shared);
    }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)nthreads);
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; i++)
        th.emplace_back([&, i] {
            dfl_team t = { (uint32_t)i, (uint32_t)nthreads, barrier_wait, &bar };
            const dfl_block_result r = dfl_encode_block_coop(&t, in, match, near, d, prm, tok, choice, out, &shared);
            if (i == 0) res = r;
        });
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&bar);
    return res;
}
extern "C" void dfl_host_set_optimal(int on) { g_optimal = on; }
extern "C" void dfl_host_set_levels(const uint32_t *lv, int n) { for (int i = 0; i < 11; i++) g_levels[i] = i < n ? lv[i] : 0; }

extern "C" size_t dfl_host_zlib(const uint8_t *in, uint32_t n, uint8_t *out, size_t cap, uint32_t max_chain,
                                uint32_t min_len, uint32_t block_bytes, uint32_t *stats)
{
    dfl_params prm = { max_chain, min_len, block_bytes };
    std::vector<uint32_t> key(n), skey(n);
    std::vector<uint32_t> sorted(n), rank(n), gstart(n), match(n, 0u), tok(block_bytes ? block_bytes : 1);
    static const uint32_t product_levels[] = DFL_DEFAULT_LEVELS;
    uint32_t default_levels[9] = { 0 };                              /* zero-terminated copy */
    for (size_t k = 0; k < sizeof product_levels / sizeof product_levels[0] && k < 8; k++) default_levels[k] = product_levels[k];
    const uint32_t *levels = g_levels[0] ? g_levels : default_levels;
    for (int lv = 0; levels[lv]; lv++) {
        for (uint32_t p = 0; p < n; p++) key[p] = dfl_sort_key(in, p, n, levels[lv]);
        std::iota(sorted.begin(), sorted.end(), 0u);
        std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        for (uint32_t i = 0; i < n; i++) {
            rank[sorted[i]] = i;
            skey[i] = key[sorted[i]];
            gstart[i] = (i && skey[i] == skey[i - 1]) ? gstart[i - 1] : i;
        }
        for (uint32_t p = 0; p < n; p++)
            match[p] = dfl_search_level(in, 0, n, p, sorted.data(), rank[p], gstart[rank[p]], dfl_level_chain(max_chain, levels[lv]), levels[lv], lv ? levels[lv - 1] : 0u, match[p]);
    }

    std::vector<uint32_t> near(n);
    for (uint32_t p = 0; p < n; p++) near[p] = dfl_near_match(in, 0, n, p);

    size_t pos = 0;
    if (cap < 8) return 0;
    out[pos++] = 0x78; out[pos++] = 0xda;
    uint32_t adler = 1;
    dfl_work work;
    std::vector<uint8_t> buf(dfl_block_bound(block_bytes) + 16);
    std::vector<uint32_t> choice(n + 1);
    if (stats) std::memset(stats, 0, 4 * sizeof(uint32_t));
    for (uint32_t b0 = 0; b0 < n; b0 += block_bytes) {
        dfl_block_desc d = { b0, std::min(n, b0 + block_bytes), 0, n, 0, 0, (uint32_t)buf.size(), b0 + block_bytes >= n ? 1u : 0u };
        std::memset(buf.data(), 0, buf.size());
        dfl_block_result r;
        if (g_team > 0) {
            r = encode_with_team(g_team, in, match.data(), near.data(), &d, &prm, tok.data(), g_optimal ? choice.data() : nullptr, buf.data());
        } else {
            r = dfl_encode_block(in, match.data(), near.data(), &d, &prm, tok.data(), g_optimal ? choice.data() : nullptr, buf.data(), &work);
            dfl_adler_partial(in, d.begin, d.end, 0, 1, &r.adler_a, &r.adler_b);
        }
        if (pos + r.bytes + 6 > cap) return 0;
        std::memcpy(out + pos, buf.data(), r.bytes);
        pos += r.bytes;
        adler = dfl_adler_fold(adler, r.adler_a, r.adler_b, d.end - d.begin);
        if (stats) { stats[r.kind]++; stats[3] += r.tokens; }
    }
    if (n == 0) { out[pos++] = 0x03; out[pos++] = 0x00; }      /* no block at all: an empty final one (the product never deflates an empty image) */
    out[pos++] = (uint8_t)(adler >> 24); out[pos++] = (uint8_t)(adler >> 16); out[pos++] = (uint8_t)(adler >> 8); out[pos++] = (uint8_t)adler;
    return pos;
}

#ifdef DFL_MAIN
int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s FILE [max_chain min_len block_bytes]\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::vector<uint8_t> in;
    uint8_t tmp[65536];
    size_t got;
    while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) in.insert(in.end(), tmp, tmp + got);
    std::fclose(f);
    const uint32_t max_chain = argc > 2 ? std::atoi(argv[2]) : 256, min_len = argc > 3 ? std::atoi(argv[3]) : 3,
                   block = argc > 4 ? std::atoi(argv[4]) : 262144;
    if (std::getenv("DFL_LAZY_ONLY")) g_optimal = 0;
    if (std::getenv("DFL_TEAM")) g_team = std::atoi(std::getenv("DFL_TEAM"));
    for (int i = 5; i < argc && i < 16; i++) g_levels[i - 5] = std::atoi(argv[i]);
    std::vector<uint8_t> out(in.size() + in.size() / 8 + 1024);
    uint32_t stats[4];
    auto t0 = std::chrono::steady_clock::now();
    const size_t z = dfl_host_zlib(in.data(), (uint32_t)in.size(), out.data(), out.size(), max_chain, min_len, block, stats);
    auto t1 = std::chrono::steady_clock::now();
    std::vector<uint8_t> back(in.size() + 16);
    uLongf blen = back.size();
    const int rc = uncompress(back.data(), &blen, out.data(), z);
    const bool ok = rc == Z_OK && blen == in.size() && !std::memcmp(back.data(), in.data(), in.size());
    /* zlib exactly as libpng drives it for the reference */
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 9, Z_DEFLATED, 15, 9, Z_FILTERED);
    std::vector<uint8_t> ref(deflateBound(&zs, in.size()));
    zs.next_in = in.data(); zs.avail_in = (uInt)in.size(); zs.next_out = ref.data(); zs.avail_out = (uInt)ref.size();
    auto t2 = std::chrono::steady_clock::now();
    deflate(&zs, Z_FINISH);
    auto t3 = std::chrono::steady_clock::now();
    const size_t zref = zs.total_out;
    deflateEnd(&zs);
    std::printf("%s: in %zu  ours %zu (%s, stored/fixed/dyn %u/%u/%u, %u tokens, %.2fs)  zlib9f %zu (%.2fs)  ratio ours/zlib %.4f\n",
                argv[1], in.size(), z, ok ? "roundtrip ok" : "ROUNDTRIP FAILED", stats[0], stats[1], stats[2], stats[3],
                std::chrono::duration<double>(t1 - t0).count(), zref, std::chrono::duration<double>(t3 - t2).count(),
                zref ? (double)z / zref : 0.0);
    return ok ? 0 : 1;
}
#endif
