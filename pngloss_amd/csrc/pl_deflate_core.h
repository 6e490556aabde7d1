/*
 * pl_deflate_core.h -- the DEFLATE encoder of the PNG write side (what
 * /root/reference/src/rwpng.c:477-637 gets from libpng + zlib level 9), restructured for a GPU:
 *
 *   1. candidates   every position is keyed by its first six bytes; a stable sort of the positions by that key puts
 *                   all earlier occurrences of a position's key right in front of it, nearest first -- the "hash
 *                   chain" of an LZ77 matcher, but built by a sort and walked through contiguous memory.
 *   2. matches      one thread per POSITION finds its longest match in the 32 KiB window (dfl_longest_match).
 *                   This is where a CPU deflate spends its time, and it is position-parallel.
 *   3. blocks       the stream is cut into deflate blocks of DFL_BLOCK input bytes; one wave per block parses
 *                   (lazy matching over the precomputed matches), builds the two length-limited Huffman codes,
 *                   picks stored / fixed / dynamic by exact size, and writes the bits.  Every block ends on a byte
 *                   boundary (empty stored block, the classic sync marker; the image's last block carries BFINAL
 *                   instead), so blocks are independent and their outputs are simply concatenated.
 *
 * The functions here are __host__ __device__: pl_deflate.hip runs them on the GPU; tests/c/deflate_host.cpp runs the
 * very same code serially on the CPU and checks it against zlib's inflate, which is how the bitstream logic is
 * validated in a container without a GPU.  Not a CPU fallback: the library only ever calls the kernels.
 */
#ifndef PL_DEFLATE_CORE_H
#define PL_DEFLATE_CORE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __HIPCC__
#define DFL_HD __host__ __device__ inline
#else
#define DFL_HD static inline
#endif

#define DFL_WINDOW     32768u
#define DFL_MIN_MATCH  3u
#define DFL_MAX_MATCH  258u
#define DFL_KEY_BYTES  6u              /* positions are grouped by their first six bytes: matches shorter than that
                                          are not worth coding in filtered image data (zlib's Z_FILTERED rule) */
#define DFL_TAIL_KEY   0xffffffffu     /* sort key of positions that have no complete key inside their image */
#define DFL_KEY_BITS   32
/* encoder settings of the product (pl_deflate.hip) -- the CPU test driver uses the same ones, so that its output is
 * byte-identical to the GPU's */
#define DFL_DEFAULT_LEVELS      { 128u, 64u, 32u, 20u, 12u, 8u, 6u }   /* key lengths of the search levels, longest first */
#define DFL_DEFAULT_MAX_CHAIN   64u                        /* candidates examined per position and level */
#define DFL_DEFAULT_BLOCK_BYTES 262144u                    /* input bytes per deflate block */
#define DFL_NUM_LL     288
#define DFL_NUM_D      32              /* 30 used + 2 so that the tables have a round size */
#define DFL_NUM_CL     19

/* a token or a match record: 0x80000000 | (len-3) << 15 | (dist-1)  -- or a literal byte (0..255) */
#define DFL_IS_MATCH(t)   ((t) >> 31)
#define DFL_TOK_LEN(t)    ((((t) >> 15) & 0xffu) + 3u)
#define DFL_TOK_DIST(t)   (((t) & 0x7fffu) + 1u)
#define DFL_MAKE_MATCH(len, dist) (0x80000000u | (((len) - 3u) << 15) | ((dist) - 1u))

typedef struct {
    uint32_t max_chain;      /* candidates examined per position */
    uint32_t min_len;        /* shortest match the parser may use (zlib's Z_FILTERED drops <= 5) */
    uint32_t block_bytes;    /* input bytes per deflate block */
} dfl_params;

/* one deflate block = [begin, end) of the concatenated scanline stream, inside image [img_begin, img_end) */
typedef struct {
    uint32_t begin, end, img_begin, img_end;
    uint32_t image;          /* index of the image in the batch */
    uint32_t out_offset;     /* where this block's bytes go in the block-output arena */
    uint32_t out_capacity;
    uint32_t last;           /* 1: the image's last block -- it carries BFINAL and no sync marker follows it */
} dfl_block_desc;

typedef struct {
    uint32_t bytes;          /* bytes produced (always a whole number: blocks end byte-aligned) */
    uint32_t kind;           /* 0 stored, 1 fixed, 2 dynamic */
    uint32_t tokens;
    uint32_t adler_a;        /* sum of the block's input bytes */
    uint64_t adler_b;        /* sum of (L - i) * byte[i] */
} dfl_block_result;

/* scratch of one block; lives in LDS on the device */
typedef struct {
    uint32_t freq_ll[DFL_NUM_LL], freq_d[DFL_NUM_D], freq_cl[DFL_NUM_CL];
    uint16_t code_ll[DFL_NUM_LL], code_d[DFL_NUM_D], code_cl[DFL_NUM_CL];
    uint8_t  len_ll[DFL_NUM_LL], len_d[DFL_NUM_D], len_cl[DFL_NUM_CL];
    uint16_t order[DFL_NUM_LL];            /* symbols sorted by frequency */
    uint32_t weight[2 * DFL_NUM_LL];
    uint16_t parent[2 * DFL_NUM_LL];
    uint8_t  depth[2 * DFL_NUM_LL];
    uint8_t  cl_sym[DFL_NUM_LL + DFL_NUM_D];    /* run-length coded code lengths: symbol 0..18 ... */
    uint8_t  cl_arg[DFL_NUM_LL + DFL_NUM_D];    /* ... and its repeat argument */
    uint16_t lenprice[DFL_MAX_MATCH + 2];       /* price in bits of a match of each length (optimal parse) */
} dfl_work;

/* candidates examined at the level with `key_bytes`-byte keys: the full chain for the long keys, a quarter of it for
 * 12..31 bytes, an eighth for the shortest.  Short matches far away are rarely worth their distance code once the
 * parse is priced, so searching hard for them only costs time (size +0.05 %, match kernels ~2x faster).  The chain is
 * non-increasing towards the short keys, which keeps the exact skip rule of dfl_search_level valid. */
DFL_HD uint32_t dfl_level_chain(uint32_t max_chain, uint32_t key_bytes)
{
    const uint32_t c = key_bytes < 12u ? max_chain / 8u : (key_bytes < 32u ? max_chain / 4u : max_chain);
    return c ? c : 1u;
}

DFL_HD uint32_t dfl_load32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

DFL_HD uint64_t dfl_key_at(const uint8_t *p)
{
    uint16_t hi;
    __builtin_memcpy(&hi, p + 4, 2);
    return (uint64_t)dfl_load32(p) | ((uint64_t)hi << 32);
}

/* Sort key of position p at a search level that groups positions by their first `nbytes` bytes: a 32-bit hash of
 * them.  A collision merges two groups, which only costs wasted comparisons (every candidate is verified byte by
 * byte); positions too close to the end of their image for a full key share one catch-all group. */
DFL_HD uint32_t dfl_sort_key(const uint8_t *s, uint32_t p, uint32_t img_end, uint32_t nbytes)
{
    if (p + nbytes > img_end) return DFL_TAIL_KEY;
    if (nbytes == DFL_KEY_BYTES) {
        const uint64_t k = dfl_key_at(s + p) * 0x9e3779b97f4a7c15ull;
        return (uint32_t)(k >> 32);
    }
    uint64_t h = 0x9e3779b97f4a7c15ull;
    uint32_t i = 0;
    for (; i + 4 <= nbytes; i += 4) {
        h = (h ^ dfl_load32(s + p + i)) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    for (; i < nbytes; i++) {
        h = (h ^ s[p + i]) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    return (uint32_t)(h >> 32);
}

/* ---------------------------------------------------------------------------------------------------------------
 * 2. longest match of position p.  One search level = positions stably sorted by dfl_sort_key: `sorted` holds the
 * positions, `r` is p's index in it and `group_start` the index of the first entry with p's key.  Walking down from
 * r-1 visits earlier positions that start with the same bytes, nearest first; the walk ends at the start of the group,
 * outside the 32 KiB window or outside the image.  Of equally long matches the nearest is kept (cheapest distance code).
 * Levels with longer keys find the long matches that hide deep in the six-byte groups of near-constant image data;
 * `best` (a match record or 0) is carried from level to level, `longer_key_bytes` is the key length of the level
 * before this one (0 for the first).
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD uint32_t dfl_search_level(const uint8_t *s, uint32_t img_begin, uint32_t img_end, uint32_t p,
                                 const uint32_t *sorted, uint32_t r, uint32_t group_start, uint32_t max_chain,
                                 uint32_t key_bytes, uint32_t longer_key_bytes, uint32_t best)
{
    const uint32_t room = img_end - p;
    if (room < DFL_KEY_BYTES) return best;
    const uint32_t max_len = room < DFL_MAX_MATCH ? room : DFL_MAX_MATCH;
    const uint8_t *b = s + p;
    uint32_t best_len = best ? DFL_TOK_LEN(best) : DFL_KEY_BYTES - 1, best_dist = best ? DFL_TOK_DIST(best) : 0;
    if (best_len >= max_len) return best;
    /* A candidate that beats a match of >= longer_key_bytes shares that many bytes with p, so it is in p's group of
     * the previous (longer-key) level; being among the nearest max_chain here it was among the nearest there and has
     * been examined already: this level cannot improve the match. */
    if (longer_key_bytes && best_len >= longer_key_bytes) return best;
    for (uint32_t chain = max_chain; chain && r > group_start; --chain) {
        const uint32_t q = sorted[--r];
        if (q < img_begin || p - q > DFL_WINDOW) break;
        const uint8_t *a = s + q;
        if (a[best_len] != b[best_len]) continue;                                /* cannot beat the best: skip */
        /* members of the group share their first key_bytes bytes (up to a hash collision), so the comparison starts
         * behind them; only a candidate that would win gets its prefix verified, and a collision is then measured
         * from the start -- the result is the one a comparison from byte 0 gives, with most of the bytes skipped */
        const uint32_t start = key_bytes < max_len ? key_bytes : 0u;
        uint32_t len = start;
        while (len + 4 <= max_len && dfl_load32(a + len) == dfl_load32(b + len)) len += 4;
        while (len < max_len && a[len] == b[len]) ++len;
        if (len > best_len && start) {
            uint32_t pre = 0;
            while (pre + 4 <= start && dfl_load32(a + pre) == dfl_load32(b + pre)) pre += 4;
            while (pre < start && a[pre] == b[pre]) ++pre;
            if (pre < start) len = pre;
        }
        if (len > best_len) {
            best_len = len;
            best_dist = p - q;
            if (len >= max_len) break;
        }
    }
    return best_dist ? DFL_MAKE_MATCH(best_len, best_dist) : 0u;
}

/* The longest match among the DFL_NEAR_DIST nearest positions (distance 1..8: the previous bytes of the same channel,
 * the previous pixel).  These have the cheapest distance codes, no extra bits up to distance 4, so the optimal parse
 * often prefers one of them to the longest match, which may lie 30 KB away; they are kept as a second record per
 * position (0.6 % of size on the reference's suite).  Of equally long ones the nearest wins. */
#define DFL_NEAR_DIST 8u

DFL_HD uint32_t dfl_near_match(const uint8_t *s, uint32_t img_begin, uint32_t img_end, uint32_t p)
{
    const uint32_t room = img_end - p;
    if (room < DFL_KEY_BYTES) return 0;
    const uint32_t max_len = room < DFL_MAX_MATCH ? room : DFL_MAX_MATCH;
    const uint8_t *b = s + p;
    uint32_t best_len = DFL_KEY_BYTES - 1, best_dist = 0;
    for (uint32_t dist = 1; dist <= DFL_NEAR_DIST && dist <= p - img_begin; dist++) {
        const uint8_t *a = b - dist;
        if (a[0] != b[0] || a[best_len] != b[best_len]) continue;
        uint32_t len = 0;
        while (len + 4 <= max_len && dfl_load32(a + len) == dfl_load32(b + len)) len += 4;
        while (len < max_len && a[len] == b[len]) ++len;
        if (len > best_len) {
            best_len = len;
            best_dist = dist;
            if (len >= max_len) break;
        }
    }
    return best_dist ? DFL_MAKE_MATCH(best_len, best_dist) : 0u;
}

/* ---------------------------------------------------------------------------------------------------------------
 * symbol mapping (RFC 1951 section 3.2.5), computed instead of tabulated
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD uint32_t dfl_log2(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

DFL_HD void dfl_len_symbol(uint32_t len, uint32_t *sym, uint32_t *ebits, uint32_t *extra)
{
    const uint32_t l = len - 3u;
    if (l < 8u) { *sym = 257u + l; *ebits = 0; *extra = 0; return; }
    if (len == 258u) { *sym = 285u; *ebits = 0; *extra = 0; return; }
    const uint32_t n = dfl_log2(l), eb = n - 2u;
    *sym = 257u + 4u * (n - 1u) + ((l >> eb) & 3u);
    *ebits = eb;
    *extra = l & ((1u << eb) - 1u);
}

DFL_HD void dfl_dist_symbol(uint32_t dist, uint32_t *sym, uint32_t *ebits, uint32_t *extra)
{
    const uint32_t d = dist - 1u;
    if (d < 4u) { *sym = d; *ebits = 0; *extra = 0; return; }
    const uint32_t n = dfl_log2(d), eb = n - 1u;
    *sym = 2u * n + ((d >> eb) & 1u);
    *ebits = eb;
    *extra = d & ((1u << eb) - 1u);
}

/* ---------------------------------------------------------------------------------------------------------------
 * 3a. parse: lazy matching over the precomputed longest matches (the decision rule of zlib's deflate_slow: a match
 * is emitted unless the next position has a strictly longer one).  Matches are clipped at the block end so that
 * the next block starts on a token boundary.  Tokens go to tok[]; returns their number and fills the histograms.
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD uint32_t dfl_clip(uint32_t m, uint32_t p, uint32_t end, uint32_t min_len)
{
    if (!m) return 0;
    uint32_t len = DFL_TOK_LEN(m);
    if (len > end - p) len = end - p;
    if (len < min_len) return 0;
    return DFL_MAKE_MATCH(len, DFL_TOK_DIST(m));
}

DFL_HD uint32_t dfl_parse_block(const uint8_t *s, const uint32_t *match, uint32_t begin, uint32_t end,
                                uint32_t min_len, uint32_t *tok, dfl_work *w)
{
    uint32_t n = 0, p = begin;
    uint32_t cur = p < end ? dfl_clip(match[p], p, end, min_len) : 0;
    while (p < end) {
        if (cur) {
            const uint32_t nxt = p + 1 < end ? dfl_clip(match[p + 1], p + 1, end, min_len) : 0;
            if (nxt && DFL_TOK_LEN(nxt) > DFL_TOK_LEN(cur)) {          /* defer: literal now, reconsider at p+1 */
                w->freq_ll[s[p]]++;
                tok[n++] = s[p];
                ++p;
                cur = nxt;
                continue;
            }
            uint32_t sym, eb, ex;
            dfl_len_symbol(DFL_TOK_LEN(cur), &sym, &eb, &ex);
            w->freq_ll[sym]++;
            dfl_dist_symbol(DFL_TOK_DIST(cur), &sym, &eb, &ex);
            w->freq_d[sym]++;
            tok[n++] = cur;
            p += DFL_TOK_LEN(cur);
        } else {
            w->freq_ll[s[p]]++;
            tok[n++] = s[p];
            ++p;
        }
        cur = p < end ? dfl_clip(match[p], p, end, min_len) : 0;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------------
 * 3a'. optimal parse.  Filtered image data has very cheap literals (a handful of small values) and a few very long
 * matches, so choosing by length (lazy matching) wastes bits: a short match far away can cost more than the literals it
 * replaces, and cutting a match short can let a much better one start.  With the code lengths of the previous parse as
 * prices, the cheapest tokenisation is a shortest path: cost[i] = min(price(literal) + cost[i+1],
 * price(len l, dist) + cost[i+l] for l up to the longest match at i -- or up to the longest NEAR match at i, the second
 * record of a position), evaluated backwards.
 *
 * To make that parallel the block is cut into chunks of DFL_DP_CHUNK positions and every chunk runs its own backward
 * pass, started DFL_DP_OVERLAP positions beyond its end from a neutral terminal condition (costs falling gently with
 * distance, at the per-byte price of a maximal match, so that no phase is preferred inside long runs).  The choices
 * depend only on cost differences and those forget the terminal condition within a few tokens; what a chunk keeps are
 * the choices for its own positions.  The chunking is part of the algorithm's definition (it does not depend on how
 * many threads work on the block), so every implementation produces the same tokens.
 * choice[p] = the token to use IF a token starts at p; the tokens of the block are the orbit of `begin`.
 * ------------------------------------------------------------------------------------------------------------- */
#define DFL_UNSEEN_PRICE  12u          /* bits charged for a symbol the previous parse never used */
#define DFL_DP_CHUNK      1024u
#define DFL_DP_OVERLAP    512u
#define DFL_DP_NEAR       4u           /* lengths just below the longest that are always tried */
#define DFL_DP_TOPS       5u           /* then the longest length of this many length codes, from the match's own downwards */
#define DFL_DP_ITERATIONS 3            /* parse by length, then this many optimal parses, each priced by the one before */
#define DFL_DP_RING       264u         /* >= 258 + 1, the furthest cost a position looks at */

/* price in bits of every match length under the current lit/len code */
DFL_HD void dfl_length_prices(dfl_work *w, uint32_t first, uint32_t step)
{
    for (uint32_t l = DFL_MIN_MATCH + first; l <= DFL_MAX_MATCH; l += step) {
        uint32_t sym, eb, ex;
        dfl_len_symbol(l, &sym, &eb, &ex);
        w->lenprice[l] = (uint16_t)((w->len_ll[sym] ? w->len_ll[sym] : DFL_UNSEEN_PRICE) + eb);
    }
}

/* top of the length code below the one that expresses l (RFC 1951 3.2.5) */
DFL_HD uint32_t dfl_prev_code_top(uint32_t l)
{
    const uint32_t v = l - 3u;
    uint32_t base;
    if (v < 8u) base = l;
    else if (l == 258u) base = 258u;
    else { const uint32_t eb = dfl_log2(v) - 2u; base = ((v >> eb) << eb) + 3u; }
    return base - 1u;
}

/* `ring` holds the cost window: cost[i] lives in ring[(i % DFL_DP_RING) * stride] as a 16-bit number (a run is at most
 * DFL_DP_CHUNK + DFL_DP_OVERLAP + 258 positions of at most 15 bits each above a terminal level of 1024).  On the
 * device the rings of a workgroup's threads are interleaved in LDS (stride = number of threads). */
DFL_HD void dfl_dp_chunk(const uint8_t *s, const uint32_t *match, const uint32_t *near, uint32_t begin, uint32_t end,
                         uint32_t min_len, uint32_t chunk, const dfl_work *w, uint32_t *choice, uint16_t *ring, uint32_t stride)
{
    const uint32_t L = end - begin, c0 = chunk * DFL_DP_CHUNK;
    if (c0 >= L) return;
    const uint32_t c1 = c0 + DFL_DP_CHUNK < L ? c0 + DFL_DP_CHUNK : L;
    const uint32_t top = c1 + DFL_DP_OVERLAP < L ? c1 + DFL_DP_OVERLAP : L;
    const uint32_t limit = top == L ? L : (top + DFL_MAX_MATCH < L ? top + DFL_MAX_MATCH : L);
    const uint32_t slope_q8 = ((uint32_t)w->lenprice[DFL_MAX_MATCH] << 8) / DFL_MAX_MATCH;
    for (uint32_t k = 0; top + k <= limit; k++)
        ring[((top + k) % DFL_DP_RING) * stride] = (uint16_t)(top == L ? 0u : 1024u - ((k * slope_q8) >> 8));
    uint32_t next_cost = ring[(top % DFL_DP_RING) * stride];       /* cost[i + 1], kept in a register */
    uint32_t slot = top % DFL_DP_RING;                             /* ring slot of position i, kept incrementally */
    /* the inputs of a position do not depend on the costs: they are fetched two positions ahead */
    const bool two = top >= c0 + 2u;
    uint32_t m_a = match[begin + top - 1u], m_b = two ? match[begin + top - 2u] : 0u;
    uint32_t n_a = near[begin + top - 1u], n_b = two ? near[begin + top - 2u] : 0u;
    uint32_t b_a = s[begin + top - 1u], b_b = two ? s[begin + top - 2u] : 0u;
    for (uint32_t i = top; i-- > c0;) {
        const uint32_t p = begin + i;
        const uint32_t byte = b_a;
        uint32_t rec[2] = { dfl_clip(m_a, p, end, min_len), dfl_clip(n_a, p, end, min_len) };
        m_a = m_b; n_a = n_b; b_a = b_b;
        if (i >= c0 + 2u) { m_b = match[p - 2u]; n_b = near[p - 2u]; b_b = s[p - 2u]; }
        slot = slot ? slot - 1u : DFL_DP_RING - 1u;
        const uint32_t lit = w->len_ll[byte] ? w->len_ll[byte] : DFL_UNSEEN_PRICE;
        uint32_t best = lit + next_cost, best_tok = byte;
        if (rec[1] && rec[0] && DFL_TOK_DIST(rec[1]) == DFL_TOK_DIST(rec[0])) rec[1] = 0;    /* the same match, or a prefix of it */
        for (uint32_t r = 0; r < 2; r++) {
            const uint32_t m = rec[r];
            if (!m) continue;
            uint32_t len = DFL_TOK_LEN(m);
            if (len > limit - i) len = limit - i;
            uint32_t sym, eb, ex;
            dfl_dist_symbol(DFL_TOK_DIST(m), &sym, &eb, &ex);
            const uint32_t dist_price = (w->len_d[sym] ? w->len_d[sym] : DFL_UNSEEN_PRICE) + eb;
            /* candidates: the DFL_DP_NEAR lengths just below the longest, then one per length code -- the longest
             * length the code can express (within a code the price is the same and the cost to go almost never
             * rises with distance) -- for the DFL_DP_TOPS codes from the match's own downwards: <= 9 candidates
             * instead of <= 253, for 0.2 % of size (cutting a match to less than a quarter almost never pays) */
            for (uint32_t j = 1; j <= DFL_DP_NEAR && len >= min_len + j; j++) {
                const uint32_t l = len - j;
                const uint32_t at = slot + l < DFL_DP_RING ? slot + l : slot + l - DFL_DP_RING;
                const uint32_t c = w->lenprice[l] + dist_price + ring[at * stride];
                if (c < best) { best = c; best_tok = DFL_MAKE_MATCH(l, DFL_TOK_DIST(m)); }
            }
            uint32_t tops = 0;
            for (uint32_t l = len; l >= min_len && tops < DFL_DP_TOPS; l = dfl_prev_code_top(l), tops++) {
                const uint32_t at = slot + l < DFL_DP_RING ? slot + l : slot + l - DFL_DP_RING;
                const uint32_t c = w->lenprice[l] + dist_price + ring[at * stride];
                if (c < best) { best = c; best_tok = DFL_MAKE_MATCH(l, DFL_TOK_DIST(m)); }
            }
        }
        if (i < c1) choice[p] = best_tok;
        ring[slot * stride] = (uint16_t)best;
        next_cost = best;
    }
}

/* tokens of the block from the choices: the orbit of `begin` */
DFL_HD uint32_t dfl_parse_block_chosen(const uint8_t *s, const uint32_t *choice, uint32_t begin, uint32_t end,
                                       uint32_t *tok, dfl_work *w)
{
    uint32_t n = 0;
    for (uint32_t p = begin; p < end;) {
        const uint32_t t = choice[p];
        if (DFL_IS_MATCH(t)) {
            uint32_t sym, eb, ex;
            dfl_len_symbol(DFL_TOK_LEN(t), &sym, &eb, &ex);
            w->freq_ll[sym]++;
            dfl_dist_symbol(DFL_TOK_DIST(t), &sym, &eb, &ex);
            w->freq_d[sym]++;
            p += DFL_TOK_LEN(t);
        } else {
            w->freq_ll[t]++;
            ++p;
        }
        tok[n++] = t;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------------
 * 3b. length-limited Huffman code for `n` symbols: sort by frequency, two-queue Huffman merge, clamp depths to
 * `limit` and repair the Kraft sum, hand the lengths out by rank, then canonical codes (stored bit-reversed, the
 * order deflate sends them in).  At least two symbols get a code so that a decoder always sees a complete tree.
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD uint32_t dfl_bitrev(uint32_t v, uint32_t bits)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

/* everything after the sort: w->order[0 .. used) holds the coded symbols by ascending (frequency, symbol) */
DFL_HD void dfl_build_code_sorted(const uint32_t *freq, uint32_t n, uint32_t used, uint32_t limit, uint8_t *len,
                                  uint16_t *code, dfl_work *w)
{
    /* leaves 0..used-1 (sorted), internal nodes used..2*used-2 created in non-decreasing weight order */
    for (uint32_t i = 0; i < used; i++) w->weight[i] = freq[w->order[i]];
    uint32_t leaf = 0, inode = used, next = used;
    while (next < 2 * used - 1) {
        uint32_t pick[2];
        for (int k = 0; k < 2; k++) {
            if (leaf < used && (inode >= next || w->weight[leaf] <= w->weight[inode])) pick[k] = leaf++;
            else pick[k] = inode++;
        }
        w->weight[next] = w->weight[pick[0]] + w->weight[pick[1]];
        w->parent[pick[0]] = (uint16_t)next;
        w->parent[pick[1]] = (uint16_t)next;
        ++next;
    }
    const uint32_t root = 2 * used - 2;
    w->depth[root] = 0;
    uint32_t count[32];
    for (uint32_t i = 0; i < 32; i++) count[i] = 0;
    for (uint32_t i = root; i-- > 0;) {
        uint32_t d = (uint32_t)w->depth[w->parent[i]] + 1u;
        if (d > 255u) d = 255u;
        w->depth[i] = (uint8_t)d;
        if (i < used) count[d > limit ? limit : d]++;
    }
    /* Kraft repair after clamping: total is in units of 2^-limit */
    uint32_t total = 0;
    for (uint32_t l = 1; l <= limit; l++) total += count[l] << (limit - l);
    while (total > (1u << limit)) {
        count[limit]--;
        for (uint32_t l = limit - 1; l >= 1; l--)
            if (count[l]) { count[l]--; count[l + 1] += 2; break; }
        total--;
    }
    /* least frequent symbols get the longest codes */
    uint32_t k = 0;
    for (uint32_t l = limit; l >= 1; l--)
        for (uint32_t c = count[l]; c; c--) len[w->order[k++]] = (uint8_t)l;
    /* canonical codes */
    uint32_t next_code[17];
    uint32_t c = 0;
    count[0] = 0;
    for (uint32_t l = 1; l <= limit; l++) { c = (c + count[l - 1]) << 1; next_code[l] = c; }
    for (uint32_t i = 0; i < n; i++)
        if (len[i]) code[i] = (uint16_t)dfl_bitrev(next_code[len[i]]++, len[i]);
}

DFL_HD void dfl_build_code(uint32_t *freq, uint32_t n, uint32_t limit, uint8_t *len, uint16_t *code, dfl_work *w)
{
    uint32_t used = 0;
    for (uint32_t i = 0; i < n; i++) { len[i] = 0; code[i] = 0; if (freq[i]) w->order[used++] = (uint16_t)i; }
    for (uint32_t i = 0; used < 2 && i < n; i++)            /* force two coded symbols */
        if (!freq[i]) { freq[i] = 1; w->order[used++] = (uint16_t)i; }
    /* sort ascending by (freq, symbol): insertion sort over <= 288 entries */
    for (uint32_t i = 1; i < used; i++) {
        const uint16_t v = w->order[i];
        const uint32_t fv = freq[v];
        uint32_t j = i;
        while (j && (freq[w->order[j - 1]] > fv || (freq[w->order[j - 1]] == fv && w->order[j - 1] > v))) {
            w->order[j] = w->order[j - 1];
            --j;
        }
        w->order[j] = v;
    }
    dfl_build_code_sorted(freq, n, used, limit, len, code, w);
}

/* ---------------------------------------------------------------------------------------------------------------
 * bit writer: LSB-first into 32-bit words of a zero-based, 4-byte aligned buffer
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *out; uint32_t pos; uint64_t acc; uint32_t nbits; } dfl_bits;

DFL_HD void dfl_put(dfl_bits *b, uint32_t value, uint32_t n)
{
    b->acc |= (uint64_t)value << b->nbits;
    b->nbits += n;
    if (b->nbits >= 32u) {
        const uint32_t v = (uint32_t)b->acc;
        __builtin_memcpy(b->out + b->pos, &v, 4);
        b->pos += 4;
        b->acc >>= 32;
        b->nbits -= 32u;
    }
}

DFL_HD void dfl_flush_to_byte(dfl_bits *b)
{
    while (b->nbits > 0) {
        b->out[b->pos++] = (uint8_t)b->acc;
        b->acc >>= 8;
        b->nbits = b->nbits > 8u ? b->nbits - 8u : 0u;
    }
    b->acc = 0;
}

/* run-length code the concatenated code lengths (RFC 1951 3.2.7) into cl_sym/cl_arg; returns the item count */
DFL_HD uint32_t dfl_rle_lengths(const uint8_t *lens, uint32_t n, dfl_work *w)
{
    uint32_t items = 0, i = 0;
    while (i < n) {
        const uint8_t v = lens[i];
        uint32_t run = 1;
        while (i + run < n && lens[i + run] == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11) { const uint32_t r = run > 138 ? 138 : run; w->cl_sym[items] = 18; w->cl_arg[items++] = (uint8_t)(r - 11); run -= r; }
            if (run >= 3) { w->cl_sym[items] = 17; w->cl_arg[items++] = (uint8_t)(run - 3); run = 0; }
            while (run--) { w->cl_sym[items] = 0; w->cl_arg[items++] = 0; }
        } else {
            w->cl_sym[items] = v; w->cl_arg[items++] = 0; --run;
            while (run >= 3) { const uint32_t r = run > 6 ? 6 : run; w->cl_sym[items] = 16; w->cl_arg[items++] = (uint8_t)(r - 3); run -= r; }
            while (run--) { w->cl_sym[items] = v; w->cl_arg[items++] = 0; }
        }
    }
    return items;
}

DFL_HD void dfl_fixed_lengths(uint8_t *len_ll, uint8_t *len_d)
{
    for (uint32_t i = 0; i < DFL_NUM_LL; i++) len_ll[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
    for (uint32_t i = 0; i < DFL_NUM_D; i++) len_d[i] = 5;
}

DFL_HD void dfl_canonical(const uint8_t *len, uint32_t n, uint16_t *code)
{
    uint32_t count[17], next_code[17];
    for (uint32_t i = 0; i < 17; i++) count[i] = 0;
    for (uint32_t i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    uint32_t c = 0;
    for (uint32_t l = 1; l <= 16; l++) { c = (c + count[l - 1]) << 1; next_code[l] = c; }
    for (uint32_t i = 0; i < n; i++) code[i] = len[i] ? (uint16_t)dfl_bitrev(next_code[len[i]]++, len[i]) : 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * 3c. one whole block: parse, code, choose the representation, write.  `out` must be 4-byte aligned, zero-offset for
 * this block, with room for the stored form (input + 5 bytes per 65535 + 16).  Returns the result record.
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD dfl_block_result dfl_encode_block(const uint8_t *s, const uint32_t *match, const uint32_t *near, const dfl_block_desc *d,
                                         const dfl_params *prm, uint32_t *tok, uint32_t *choice, uint8_t *out, dfl_work *w)
{
    static const uint8_t cl_order[DFL_NUM_CL] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    dfl_block_result res;
    const uint32_t L = d->end - d->begin;
    for (uint32_t i = 0; i < DFL_NUM_LL; i++) w->freq_ll[i] = 0;
    for (uint32_t i = 0; i < DFL_NUM_D; i++) w->freq_d[i] = 0;
    for (uint32_t i = 0; i < DFL_NUM_CL; i++) w->freq_cl[i] = 0;

    res.adler_a = 0;                 /* filled in by the caller (dfl_adler_partial, lane-parallel on the device) */
    res.adler_b = 0;

    uint32_t ntok = dfl_parse_block(s, match, d->begin, d->end, prm->min_len, tok, w);
    if (choice && ntok < L) {                                    /* there are matches: price them and parse again */
        uint16_t ring[DFL_DP_RING];
        for (int it = 0; it < DFL_DP_ITERATIONS; it++) {
            w->freq_ll[256] = 1;
            dfl_build_code(w->freq_ll, 286, 15, w->len_ll, w->code_ll, w);
            dfl_build_code(w->freq_d, 30, 15, w->len_d, w->code_d, w);
            dfl_length_prices(w, 0, 1);
            for (uint32_t c = 0; c * DFL_DP_CHUNK < L; c++) dfl_dp_chunk(s, match, near, d->begin, d->end, prm->min_len, c, w, choice, ring, 1);
            for (uint32_t i = 0; i < DFL_NUM_LL; i++) w->freq_ll[i] = 0;
            for (uint32_t i = 0; i < DFL_NUM_D; i++) w->freq_d[i] = 0;
            ntok = dfl_parse_block_chosen(s, choice, d->begin, d->end, tok, w);
        }
    }
    w->freq_ll[256] = 1;
    res.tokens = ntok;

    /* extra bits are the same for fixed and dynamic */
    uint64_t extra_bits = 0;
    for (uint32_t i = 265; i < 285; i++) extra_bits += (uint64_t)w->freq_ll[i] * ((i - 261u) >> 2);
    for (uint32_t i = 4; i < 30; i++) extra_bits += (uint64_t)w->freq_d[i] * ((i - 2u) >> 1);

    /* fixed cost (before dfl_build_code may force extra symbols into the histograms) */
    uint64_t fixed_bits = 3 + extra_bits;
    for (uint32_t i = 0; i < DFL_NUM_LL; i++) fixed_bits += (uint64_t)w->freq_ll[i] * (i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
    for (uint32_t i = 0; i < 30; i++) fixed_bits += (uint64_t)w->freq_d[i] * 5;

    /* dynamic: codes, header */
    uint64_t dyn_bits = 3 + 14 + extra_bits;
    uint64_t body = 0;
    dfl_build_code(w->freq_ll, 286, 15, w->len_ll, w->code_ll, w);
    dfl_build_code(w->freq_d, 30, 15, w->len_d, w->code_d, w);
    for (uint32_t i = 0; i < 286; i++) body += (uint64_t)w->freq_ll[i] * w->len_ll[i];
    for (uint32_t i = 0; i < 30; i++) body += (uint64_t)w->freq_d[i] * w->len_d[i];
    uint32_t hlit = 286, hdist = 30;
    while (hlit > 257 && !w->len_ll[hlit - 1]) --hlit;
    while (hdist > 1 && !w->len_d[hdist - 1]) --hdist;
    /* concatenate the two length vectors for the run-length coder (reuse weight[] as byte scratch) */
    uint8_t *all = (uint8_t *)w->weight;
    for (uint32_t i = 0; i < hlit; i++) all[i] = w->len_ll[i];
    for (uint32_t i = 0; i < hdist; i++) all[hlit + i] = w->len_d[i];
    const uint32_t items = dfl_rle_lengths(all, hlit + hdist, w);
    for (uint32_t i = 0; i < items; i++) w->freq_cl[w->cl_sym[i]]++;
    dfl_build_code(w->freq_cl, DFL_NUM_CL, 7, w->len_cl, w->code_cl, w);
    uint32_t hclen = DFL_NUM_CL;
    while (hclen > 4 && !w->len_cl[cl_order[hclen - 1]]) --hclen;
    dyn_bits += 3ull * hclen;
    for (uint32_t i = 0; i < items; i++) {
        const uint32_t sy = w->cl_sym[i];
        dyn_bits += w->len_cl[sy] + (sy == 16 ? 2u : (sy == 17 ? 3u : (sy == 18 ? 7u : 0u)));
    }
    /* forced symbols were given frequency 1 and are counted in `body`; at most 2+2 codes, harmless over-estimate */
    dyn_bits += body;

    const uint32_t chunks = L ? (L + 65534u) / 65535u : 1u;
    const uint64_t stored_bits = 8ull * ((uint64_t)L + 5ull * chunks);
    const uint64_t sync_bits = d->last ? 0u : 8ull * 5;          /* upper bound of the closing sync marker */

    dfl_bits bw;
    bw.out = out; bw.pos = 0; bw.acc = 0; bw.nbits = 0;
    if (stored_bits <= fixed_bits + sync_bits && stored_bits <= dyn_bits + sync_bits) {
        res.kind = 0;
        uint32_t off = 0;
        for (uint32_t c = 0; c < chunks; c++) {
            const uint32_t len = L - off > 65535u ? 65535u : L - off;
            out[bw.pos++] = (d->last && c + 1 == chunks) ? 1 : 0;   /* BFINAL, BTYPE 00, padding */
            out[bw.pos++] = (uint8_t)len; out[bw.pos++] = (uint8_t)(len >> 8);
            out[bw.pos++] = (uint8_t)~len; out[bw.pos++] = (uint8_t)(~len >> 8);
            for (uint32_t i = 0; i < len; i++) out[bw.pos++] = s[d->begin + off + i];
            off += len;
        }
        res.bytes = bw.pos;
        return res;
    }
    if (fixed_bits <= dyn_bits) {
        res.kind = 1;
        dfl_fixed_lengths(w->len_ll, w->len_d);
        dfl_canonical(w->len_ll, DFL_NUM_LL, w->code_ll);
        dfl_canonical(w->len_d, DFL_NUM_D, w->code_d);
        dfl_put(&bw, d->last | (1u << 1), 3);                     /* BFINAL, BTYPE 01 */
    } else {
        res.kind = 2;
        dfl_put(&bw, d->last | (2u << 1), 3);                     /* BFINAL, BTYPE 10 */
        dfl_put(&bw, hlit - 257u, 5);
        dfl_put(&bw, hdist - 1u, 5);
        dfl_put(&bw, hclen - 4u, 4);
        for (uint32_t i = 0; i < hclen; i++) dfl_put(&bw, w->len_cl[cl_order[i]], 3);
        for (uint32_t i = 0; i < items; i++) {
            const uint32_t sy = w->cl_sym[i];
            dfl_put(&bw, w->code_cl[sy], w->len_cl[sy]);
            if (sy == 16) dfl_put(&bw, w->cl_arg[i], 2);
            else if (sy == 17) dfl_put(&bw, w->cl_arg[i], 3);
            else if (sy == 18) dfl_put(&bw, w->cl_arg[i], 7);
        }
    }
    for (uint32_t i = 0; i < ntok; i++) {
        const uint32_t t = tok[i];
        if (DFL_IS_MATCH(t)) {
            uint32_t sym, eb, ex;
            dfl_len_symbol(DFL_TOK_LEN(t), &sym, &eb, &ex);
            dfl_put(&bw, (uint32_t)w->code_ll[sym] | (ex << w->len_ll[sym]), w->len_ll[sym] + eb);
            dfl_dist_symbol(DFL_TOK_DIST(t), &sym, &eb, &ex);
            dfl_put(&bw, (uint32_t)w->code_d[sym] | (ex << w->len_d[sym]), w->len_d[sym] + eb);
        } else {
            dfl_put(&bw, w->code_ll[t], w->len_ll[t]);
        }
    }
    dfl_put(&bw, w->code_ll[256], w->len_ll[256]);
    if (d->last) {
        dfl_flush_to_byte(&bw);                                  /* the stream ends here */
    } else {
        /* sync marker: empty stored block, which byte-aligns the stream */
        dfl_put(&bw, 0, 3);
        dfl_flush_to_byte(&bw);
        out[bw.pos++] = 0; out[bw.pos++] = 0; out[bw.pos++] = 0xff; out[bw.pos++] = 0xff;
    }
    res.bytes = bw.pos;
    return res;
}

/* capacity the output of one block needs (stored form is the worst case; the coded forms are only chosen when
 * they are not larger than it, plus slack for the writer's 4-byte granularity) */
DFL_HD uint32_t dfl_block_bound(uint32_t input_bytes)
{
    return ((input_bytes + 5u * (input_bytes / 65535u + 1u) + 64u) + 15u) & ~15u;
}

/* Adler-32 partial sums of a block's input over the lanes of a wave: lane `lane` of `nlanes` adds every nlanes-th
 * byte; the caller adds the lanes up.  a = sum of bytes, b = sum of (L - i) * byte[i]. */
DFL_HD void dfl_adler_partial(const uint8_t *s, uint32_t begin, uint32_t end, uint32_t lane, uint32_t nlanes,
                              uint32_t *a_out, uint64_t *b_out)
{
    const uint32_t L = end - begin;
    uint32_t a = 0;
    uint64_t b = 0;
    for (uint32_t i = lane; i < L; i += nlanes) { a += s[begin + i]; b += (uint64_t)(L - i) * s[begin + i]; }
    *a_out = a;
    *b_out = b;
}

/* Adler-32 over the concatenation of blocks from their partial sums: adler = s2 << 16 | s1, starts at 1 */
DFL_HD uint32_t dfl_adler_fold(uint32_t adler, uint32_t a, uint64_t b, uint32_t L)
{
    const uint64_t P = 65521u;
    uint64_t s1 = adler & 0xffffu, s2 = adler >> 16;
    s2 = (s2 + (L % P) * s1 + b % P) % P;
    s1 = (s1 + a % P) % P;
    return (uint32_t)(s2 << 16 | s1);
}

/* zlib framing around the concatenated blocks: 78 DA | blocks (the last one carries BFINAL) | Adler-32 (BE) */
#define DFL_ZLIB_HEAD_BYTES 2u
#define DFL_ZLIB_TAIL_BYTES 4u

#endif
