/*
 * pl_deflate.hip -- zlib streams of the emitted scanlines, made on the GPU (the deflate half of
 * /root/reference/src/rwpng.c:477-637, which libpng/zlib level 9 does on one CPU core and which is ~97 % of the tool's
 * time once the row engine runs on the device).
 *
 * The algorithm lives in pl_deflate_core.h (also compiled for the CPU by tests/c/deflate_host.cpp, which checks it
 * against zlib's inflate).  Here: the kernels that run it, and the host sequencing for a batch of images.
 *
 *   dfl_pack      scanlines (filter id + filtered bytes per row) of every image  ->  one contiguous byte stream
 *   per level:    dfl_keys -> radix sort of (key, position) -> dfl_flags + max-scan -> dfl_match   (all position-parallel)
 *   dfl_encode    one workgroup per deflate block (pl_deflate_coop.h): parse, Huffman codes, bits; byte-aligned outputs
 *   dfl_gather    compacts the block outputs into one buffer per image
 *
 * Positions are 32-bit: a call handles at most DFL_MAX_STREAM bytes of scanlines at a time (the caller's images are
 * processed in groups).  Everything is HBM-resident; the only host round trip is the table of block sizes.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pl_deflate.h"
#include "pl_deflate_coop.h"

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kLevels[] = DFL_DEFAULT_LEVELS;            /* key lengths, longest first (see dfl_search_level) */
constexpr int kNumLevels = sizeof(kLevels) / sizeof(kLevels[0]);

struct DflImageDev {
    const uint8_t *ids;      /* filter type per row */
    const uint8_t *rows;     /* filtered bytes, `pitch` apart */
    uint32_t pitch, rowbytes;
};

__global__ __launch_bounds__(kThreads) void dfl_pack(const dfl_block_desc *desc, const DflImageDev *img, uint8_t *s)
{
    const dfl_block_desc d = desc[blockIdx.y];
    const uint32_t p = d.begin + blockIdx.x * kThreads + threadIdx.x;
    if (p >= d.end) return;
    const DflImageDev im = img[d.image];
    const uint32_t stride = im.rowbytes + 1u, o = p - d.img_begin;
    const uint32_t y = o / stride, x = o - y * stride;
    s[p] = x ? im.rows[(size_t)y * im.pitch + (x - 1u)] : im.ids[y];
}

/* second record per position: the longest match at distance 1..8 (dfl_near_match) */
__global__ __launch_bounds__(kThreads) void dfl_near(const dfl_block_desc *desc, const uint8_t *s, uint32_t *near)
{
    const dfl_block_desc d = desc[blockIdx.y];
    const uint32_t p = d.begin + blockIdx.x * kThreads + threadIdx.x;
    if (p >= d.end) return;
    near[p] = dfl_near_match(s, d.img_begin, d.img_end, p);
}

__global__ __launch_bounds__(kThreads) void dfl_keys(const dfl_block_desc *desc, const uint8_t *s, uint32_t nbytes,
                                                     uint32_t *key, uint32_t *val)
{
    const dfl_block_desc d = desc[blockIdx.y];
    const uint32_t p = d.begin + blockIdx.x * kThreads + threadIdx.x;
    if (p >= d.end) return;
    key[p] = dfl_sort_key(s, p, d.img_end, nbytes);
    val[p] = p;
}

/* flag[i] = i where a new key group starts in the sorted order (else 0): an inclusive max-scan of flag[] then gives
 * every entry the index of the first entry of its group */
__global__ __launch_bounds__(kThreads) void dfl_flags(const uint32_t *skey, uint32_t n, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) flag[i] = (i && skey[i] != skey[i - 1]) ? i : 0u;
}

/* One search level, one thread per entry of the SORTED order: neighbouring threads are positions with the same
 * context (similar chain lengths, shared candidate lists), no inverse permutation is needed, and the positions that the
 * exact skip rule exempts from this level (most of them, in compressible data) leave after one load. */
__global__ __launch_bounds__(kThreads) void dfl_match(const uint32_t *sorted, const uint32_t *group_start, uint32_t n,
                                                      const uint8_t *s, const uint32_t *img_begin, uint32_t nimg,
                                                      uint32_t max_chain, uint32_t key_bytes, uint32_t longer_key_bytes, uint32_t *match)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = sorted[i];
    const uint32_t best = longer_key_bytes ? match[p] : 0u;
    if (best && DFL_TOK_LEN(best) >= longer_key_bytes) return;
    uint32_t lo = 0, hi = nimg;                 /* image of p: img_begin[lo] <= p < img_begin[lo + 1] */
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (img_begin[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t m = dfl_search_level(s, img_begin[lo], img_begin[lo + 1], p, sorted, i, group_start[i], max_chain,
                                        key_bytes, longer_key_bytes, best);
    if (m != best || !longer_key_bytes) match[p] = m;
}

/* one 256-thread workgroup per deflate block: pl_deflate_coop.h; `arena` must be zero (the bits are OR-ed in) */
__global__ __launch_bounds__(kThreads) void dfl_encode(const dfl_block_desc *desc, const uint8_t *s, const uint32_t *match, const uint32_t *near,
                                                       dfl_params prm, uint32_t *tok, uint32_t *choice, uint8_t *arena,
                                                       dfl_block_result *result)
{
    __shared__ dfl_coop shared;
    const dfl_block_desc d = desc[blockIdx.x];
    dfl_team team = { threadIdx.x, kThreads, nullptr, nullptr };
    const dfl_block_result res = dfl_encode_block_coop(&team, s, match, near, &d, &prm, tok + d.begin, choice, arena + d.out_offset, &shared);
    if (threadIdx.x == 0) result[blockIdx.x] = res;
}

__global__ __launch_bounds__(kThreads) void dfl_gather(const dfl_block_desc *desc, const dfl_block_result *result,
                                                       const uint32_t *dest, const uint8_t *arena, uint8_t *compact)
{
    const dfl_block_desc d = desc[blockIdx.x];
    const uint32_t n = result[blockIdx.x].bytes;
    const uint8_t *src = arena + d.out_offset;
    uint8_t *dst = compact + dest[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) dst[i] = src[i];
}

#define DFL_CHECK(expr)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "pngloss_hip: %s failed: %s\n", #expr, hipGetErrorString(e_));                        \
            rc = e_;                                                                                                   \
            goto done;                                                                                                 \
        }                                                                                                              \
    } while (0)

template <class T> hipError_t dev_alloc(T **p, size_t count) { return hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)); }

/* one group of images whose streams fit 32-bit positions */
hipError_t deflate_group(pl_deflate_image *imgs, size_t n, const dfl_params &prm, hipStream_t stream)
{
    hipError_t rc = hipSuccess;
    std::vector<dfl_block_desc> desc;
    std::vector<DflImageDev> dev_img(n);
    std::vector<uint32_t> first_block(n + 1, 0);
    uint32_t total = 0, arena_bytes = 0, max_block = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t len = (imgs[i].rowbytes + 1u) * imgs[i].height;
        dev_img[i] = DflImageDev{ imgs[i].d_filter_types, imgs[i].d_scanlines, imgs[i].pitch, imgs[i].rowbytes };
        first_block[i] = (uint32_t)desc.size();
        for (uint32_t b0 = 0; b0 < len; b0 += prm.block_bytes) {
            const uint32_t bl = std::min(prm.block_bytes, len - b0);
            desc.push_back(dfl_block_desc{ total + b0, total + b0 + bl, total, total + len, (uint32_t)i, arena_bytes, dfl_block_bound(bl),
                                           b0 + bl == len ? 1u : 0u });
            arena_bytes += dfl_block_bound(bl);
            max_block = std::max(max_block, bl);
        }
        total += len;
    }
    first_block[n] = (uint32_t)desc.size();
    const uint32_t nblocks = (uint32_t)desc.size();
    for (size_t i = 0; i < n; i++) imgs[i].out_size = 0;
    if (!nblocks) return hipSuccess;

    uint8_t *d_s = nullptr, *d_arena = nullptr, *d_compact = nullptr, *d_temp = nullptr;
    uint32_t *d_key[2] = { nullptr, nullptr };
    uint32_t *d_val[2] = { nullptr, nullptr }, *d_choice = nullptr, *d_match = nullptr, *d_near = nullptr, *d_tok = nullptr, *d_dest = nullptr;
    dfl_block_desc *d_desc = nullptr;
    dfl_block_result *d_result = nullptr;
    DflImageDev *d_img = nullptr;
    size_t temp_bytes = 0, scan_bytes = 0;
    uint32_t *d_flag = nullptr, *d_gstart = nullptr;
    std::vector<dfl_block_result> result(nblocks);
    std::vector<uint32_t> dest(nblocks);
    std::vector<uint32_t> img_off(n + 1, 0), img_begin;
    uint32_t *d_img_begin = nullptr;
    const dim3 pos_grid((max_block + kThreads - 1) / kThreads, nblocks);

    const bool debug = std::getenv("PNGLOSS_HIP_DEBUG") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double ms_alloc = 0, ms_kernels = 0, ms_gather = 0, ms_copy = 0;
    DFL_CHECK(dev_alloc(&d_s, (size_t)total + 512));              /* slack: the key/compare loads may run past the end */
    DFL_CHECK(dev_alloc(&d_key[0], total));
    DFL_CHECK(dev_alloc(&d_key[1], total));
    DFL_CHECK(dev_alloc(&d_val[0], total));
    DFL_CHECK(dev_alloc(&d_val[1], total));
    DFL_CHECK(dev_alloc(&d_choice, total));               /* token choices of the optimal parse */
    DFL_CHECK(dev_alloc(&d_match, total));
    DFL_CHECK(dev_alloc(&d_near, total));
    DFL_CHECK(dev_alloc(&d_tok, total));
    DFL_CHECK(dev_alloc(&d_arena, arena_bytes));
    DFL_CHECK(dev_alloc(&d_desc, nblocks));
    DFL_CHECK(dev_alloc(&d_result, nblocks));
    DFL_CHECK(dev_alloc(&d_dest, nblocks));
    DFL_CHECK(dev_alloc(&d_img, n));
    DFL_CHECK(dev_alloc(&d_img_begin, n + 1));
    DFL_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], total, 0, DFL_KEY_BITS, stream));
    d_flag = d_val[0];                                             /* both are free between the sort and the next level's keys */
    d_gstart = d_key[0];
    DFL_CHECK(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, d_flag, d_gstart, hipcub::Max(), total, stream));
    temp_bytes = std::max(temp_bytes, scan_bytes);
    DFL_CHECK(dev_alloc(&d_temp, temp_bytes));
    ms_alloc = ms_since(t_begin);
    DFL_CHECK(hipMemsetAsync(d_s + total, 0, 512, stream));
    DFL_CHECK(hipMemcpyAsync(d_desc, desc.data(), sizeof(dfl_block_desc) * nblocks, hipMemcpyHostToDevice, stream));
    DFL_CHECK(hipMemcpyAsync(d_img, dev_img.data(), sizeof(DflImageDev) * n, hipMemcpyHostToDevice, stream));
    {   /* start of every image in the stream (+ the end), for the sorted-order kernels */
        uint32_t at = 0;
        for (size_t i = 0; i < n; i++) { img_begin.push_back(at); at += (imgs[i].rowbytes + 1u) * imgs[i].height; }
        img_begin.push_back(at);
        DFL_CHECK(hipMemcpyAsync(d_img_begin, img_begin.data(), sizeof(uint32_t) * (n + 1), hipMemcpyHostToDevice, stream));
    }

    dfl_pack<<<pos_grid, kThreads, 0, stream>>>(d_desc, d_img, d_s);
    dfl_near<<<pos_grid, kThreads, 0, stream>>>(d_desc, d_s, d_near);
    for (int lv = 0; lv < kNumLevels; lv++) {
        dfl_keys<<<pos_grid, kThreads, 0, stream>>>(d_desc, d_s, kLevels[lv], d_key[0], d_val[0]);
        DFL_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], total, 0, DFL_KEY_BITS, stream));
        dfl_flags<<<(total + kThreads - 1) / kThreads, kThreads, 0, stream>>>(d_key[1], total, d_flag);
        DFL_CHECK(hipcub::DeviceScan::InclusiveScan(d_temp, temp_bytes, d_flag, d_gstart, hipcub::Max(), total, stream));
        dfl_match<<<(total + kThreads - 1) / kThreads, kThreads, 0, stream>>>(d_val[1], d_gstart, total, d_s, d_img_begin, (uint32_t)n,
                                                                                  dfl_level_chain(prm.max_chain, kLevels[lv]), kLevels[lv], lv ? kLevels[lv - 1] : 0u, d_match);
    }
    DFL_CHECK(hipMemsetAsync(d_arena, 0, arena_bytes, stream));
    dfl_encode<<<nblocks, kThreads, 0, stream>>>(d_desc, d_s, d_match, d_near, prm, d_tok, d_choice, d_arena, d_result);
    DFL_CHECK(hipGetLastError());
    DFL_CHECK(hipMemcpyAsync(result.data(), d_result, sizeof(dfl_block_result) * nblocks, hipMemcpyDeviceToHost, stream));
    DFL_CHECK(hipStreamSynchronize(stream));
    ms_kernels = ms_since(t_begin) - ms_alloc;

    {   /* compact layout: image i's blocks back to back at img_off[i] */
        uint32_t cursor = 0;
        for (size_t i = 0; i < n; i++) {
            img_off[i] = cursor;
            for (uint32_t b = first_block[i]; b < first_block[i + 1]; b++) { dest[b] = cursor; cursor += result[b].bytes; }
        }
        img_off[n] = cursor;
        DFL_CHECK(dev_alloc(&d_compact, (size_t)cursor + 16));
        DFL_CHECK(hipMemcpyAsync(d_dest, dest.data(), sizeof(uint32_t) * nblocks, hipMemcpyHostToDevice, stream));
        dfl_gather<<<nblocks, kThreads, 0, stream>>>(d_desc, d_result, d_dest, d_arena, d_compact);
        DFL_CHECK(hipGetLastError());
        DFL_CHECK(hipStreamSynchronize(stream));
        ms_gather = ms_since(t_begin) - ms_alloc - ms_kernels;
    }
    for (size_t i = 0; i < n; i++) {
        const uint32_t body = img_off[i + 1] - img_off[i];
        if (first_block[i] == first_block[i + 1]) continue;               /* empty image: no stream */
        const size_t need = (size_t)DFL_ZLIB_HEAD_BYTES + body + DFL_ZLIB_TAIL_BYTES;
        if (!imgs[i].out || imgs[i].out_capacity < need) { rc = hipErrorInvalidValue; goto done; }
        unsigned char *o = imgs[i].out;
        o[0] = 0x78; o[1] = 0xda;                                          /* deflate, 32 KiB window, "maximum compression" */
        DFL_CHECK(hipMemcpy(o + 2, d_compact + img_off[i], body, hipMemcpyDeviceToHost));
        uint32_t adler = 1;
        uint32_t kinds[3] = { 0, 0, 0 };
        for (uint32_t b = first_block[i]; b < first_block[i + 1]; b++) {
            adler = dfl_adler_fold(adler, result[b].adler_a, result[b].adler_b, desc[b].end - desc[b].begin);
            kinds[result[b].kind < 3 ? result[b].kind : 0]++;
        }
        unsigned char *t = o + 2 + body;                                   /* the last block carried BFINAL */
        t[0] = (unsigned char)(adler >> 24); t[1] = (unsigned char)(adler >> 16); t[2] = (unsigned char)(adler >> 8); t[3] = (unsigned char)adler;
        imgs[i].out_size = need;
        imgs[i].blocks_stored = kinds[0]; imgs[i].blocks_fixed = kinds[1]; imgs[i].blocks_dynamic = kinds[2];
    }
    ms_copy = ms_since(t_begin) - ms_alloc - ms_kernels - ms_gather;
done:
    (void)hipFree(d_s); (void)hipFree(d_key[0]); (void)hipFree(d_key[1]); (void)hipFree(d_val[0]); (void)hipFree(d_val[1]);
    (void)hipFree(d_choice); (void)hipFree(d_match); (void)hipFree(d_near); (void)hipFree(d_tok); (void)hipFree(d_arena); (void)hipFree(d_desc);
    (void)hipFree(d_result); (void)hipFree(d_dest); (void)hipFree(d_img); (void)hipFree(d_temp); (void)hipFree(d_compact);
    (void)hipFree(d_img_begin);
    if (debug)
        std::fprintf(stderr, "pngloss_hip deflate: %zu images, %u bytes, %u blocks: alloc %.2f ms, kernels %.2f ms, gather %.2f ms, "
                             "download %.2f ms, free %.2f ms\n", n, total, nblocks, ms_alloc, ms_kernels, ms_gather, ms_copy,
                     ms_since(t_begin) - ms_alloc - ms_kernels - ms_gather - ms_copy);
    return rc;
}

} // namespace

size_t pl_deflate_bound(uint32_t width, uint32_t height)
{
    const size_t len = ((size_t)width * 4 + 1) * height;
    const size_t blocks = len / PL_DEFLATE_BLOCK_BYTES + 1;
    return DFL_ZLIB_HEAD_BYTES + DFL_ZLIB_TAIL_BYTES + len + blocks * (5 * (PL_DEFLATE_BLOCK_BYTES / 65535 + 1) + 80);
}

hipError_t pl_deflate_images(pl_deflate_image *imgs, size_t n, hipStream_t stream)
{
    dfl_params prm = { PL_DEFLATE_MAX_CHAIN, DFL_KEY_BYTES, PL_DEFLATE_BLOCK_BYTES };
    if (const char *e = std::getenv("PNGLOSS_HIP_DEFLATE_CHAIN")) prm.max_chain = (uint32_t)std::max(1, std::atoi(e));
    /* images are processed in groups of at most `group_bytes` of scanlines (test hook: a small value forces many groups) */
    uint64_t group_bytes = PL_DEFLATE_MAX_STREAM;
    if (const char *e = std::getenv("PNGLOSS_HIP_DEFLATE_GROUP_BYTES")) group_bytes = std::min<uint64_t>(PL_DEFLATE_MAX_STREAM, std::max(1ll, std::atoll(e)));
    size_t i = 0;
    while (i < n) {
        uint64_t bytes = 0;
        size_t j = i;
        while (j < n) {
            const uint64_t len = ((uint64_t)imgs[j].rowbytes + 1u) * imgs[j].height;
            if (len > PL_DEFLATE_MAX_STREAM) return hipErrorInvalidValue;   /* one image beyond 32-bit positions */
            if (j > i && bytes + len > group_bytes) break;
            bytes += len;
            ++j;
        }
        const hipError_t rc = deflate_group(imgs + i, j - i, prm, stream);
        if (rc != hipSuccess) return rc;
        i = j;
    }
    return hipSuccess;
}
