/*
 * pl_inflate_core.h -- INFLATE of a PNG's image data on the device: the zlib stream that a file's IDAT chunks concatenate to
 * (RFC 1950 wrapper, RFC 1951 blocks: stored / fixed / dynamic Huffman) -> the filtered scanlines.  Replaces what zlib does for
 * libpng inside rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400: png_read_image -> inflate) for the files of a window --
 * independent streams, one WAVE each (the bit stream of one file is serial; a window has dozens to hundreds of files).
 *
 * One wave per stream:
 *   lane 0      the bit reader and the Huffman decoder (state in its registers); literals go straight into the window
 *   all lanes   input staging (the compressed bytes come through a 16 KB buffer in shared memory, loaded 16 bytes a lane), table
 *               construction (a symbol a lane), LZ77 copies (a byte a lane: source index pos - dist + k mod dist is always behind
 *               pos, so the lanes do not depend on each other even when the match overlaps itself), stored blocks, and the way out:
 *               the 32 KB window lives in shared memory and leaves for device memory in 4 KB pieces, 64 bytes a lane, with the
 *               piece's share of the Adler-32 (RFC 1950) computed on the way.
 * Tables: a 10-bit direct table for literal/length codes and a 9-bit one for distances (entry = symbol | code length << 16);
 * longer codes (rare) are decoded canonically, bit by bit.
 *
 * Anything malformed (reserved block type, over-subscribed / incomplete code set beyond what zlib accepts, a distance beyond the
 * bytes produced, more or fewer bytes than the image needs, a wrong Adler-32, a preset dictionary) ends the stream with an error code:
 * the caller reads that file with zlib / libpng on the host, which also words the complaint.
 *
 * Compiled twice, like pl_seg_core.h: by hipcc into pl_inflate.hip's kernel, and by g++ into tests/c/inflate_host.cpp (lane loops
 * instead of lanes), where the CPU suite checks it against zlib -- test infrastructure; the product has no CPU path.
 */
#ifndef PL_INFLATE_CORE_H
#define PL_INFLATE_CORE_H

#include <stdint.h>
#include <stddef.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define PLI_LANES(lane) for (int lane = (int)threadIdx.x, pli_once_ = 1; pli_once_; pli_once_ = 0)
#define PLI_LANE0(lane) for (int lane = (int)threadIdx.x, pli_once_ = 1; pli_once_ && lane == 0; pli_once_ = 0)
#define PLI_SYNC() __syncthreads()
#define PLI_NL 64
__device__ __forceinline__ uint64_t pli_wave_sum(uint64_t v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
#define PLI_HD __host__ __device__ __forceinline__
#else
#define PLI_LANES(lane) for (int lane = 0; lane < PLI_NL; ++lane)
#define PLI_LANE0(lane) for (int lane = 0; lane < 1; ++lane)
#define PLI_SYNC() ((void)0)
#define PLI_NL 64
#if defined(__HIPCC__)
#define PLI_HD __host__ __device__ __forceinline__
#else
#define PLI_HD inline
#endif
#endif

#define PLI_WIN 32768u            /* the window (RFC 1951: distances up to 32768) */
#define PLI_PIECE 4096u           /* what leaves the window at a time */
#define PLI_IN 16384u             /* staged input */
#define PLI_LBITS 10
#define PLI_WAVE_COPY 32u         /* matches from this length on are copied by the whole wave */
#define PLI_DBITS 9
enum { PLI_OK = 0, PLI_E_HEADER = 1, PLI_E_BLOCK = 2, PLI_E_CODES = 3, PLI_E_SYMBOL = 4, PLI_E_DIST = 5, PLI_E_SIZE = 6, PLI_E_ADLER = 7, PLI_E_INPUT = 8 };

/* shared memory of one stream */
struct PliShared {
    uint8_t win[PLI_WIN];
    union { uint8_t in[PLI_IN + 16]; uint32_t in32[PLI_IN / 4 + 4]; };
    uint32_t ltab[1u << PLI_LBITS];   /* symbol | length << 16; 0: a longer code */
    uint32_t dtab[1u << PLI_DBITS];
    uint16_t lcount[16], dcount[16];  /* codes per length */
    uint16_t lfirst[16], dfirst[16];  /* first code of a length (canonical) */
    uint16_t loffs[16], doffs[16];    /* index of a length's first symbol in lsym / dsym */
    uint16_t lsym[288], dsym[32];     /* symbols in canonical order */
    uint8_t lens[320];                /* code lengths of the block: literal/length codes, then distance codes */
    uint16_t lbase[32], dbase[32];    /* RFC 1951 3.2.5: base of a length / distance code and its extra bits (from constant memory once: a lookup there */
    uint8_t lext[32], dext[32];       /*  costs a round trip to device memory per use) */
    uint32_t tok[8];                  /* lane 0 -> wave: [0] kind, [1] length, [2] distance, [3] error, [4] hlit, [5] hdist */
    uint64_t red[2];
};

enum { PLI_T_MATCH = 1, PLI_T_END = 2, PLI_T_FLUSH = 3, PLI_T_INPUT = 4, PLI_T_ERROR = 5 };

struct PliStream {
    const uint8_t *z;       /* the zlib stream (device / host memory) */
    uint32_t zbytes;
    uint8_t *out;           /* receives `expect` bytes */
    uint32_t expect;
    int32_t *status;        /* 0 or PLI_E_* */
};

PLI_HD uint32_t pli_rev(uint32_t code, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; } return r; }

/* the decoder's bit reader (lane 0): bytes come from the staged input, `ip` = next byte of the stage, `iend` = bytes in it */
struct PliBits { uint64_t buf; int n; uint32_t ip, iend; };
/* at most 32 bits short: four bytes at once -- two aligned words of the stage (one round trip to shared memory), shifted to the byte position;
 * byte by byte only at the stage's end.  (A symbol with its extra bits takes at most 28 bits: one refill in front of each is enough.)
 * (S, not a pointer into it: through a plain pointer the accesses become FLAT ones, several times slower than ds_read) */
PLI_HD uint32_t pli_peek(const PliBits &b, int k) { return (uint32_t)(b.buf & ((1ull << k) - 1ull)); }
PLI_HD void pli_drop(PliBits &b, int k) { b.buf >>= k; b.n -= k; }

/* canonical tables of one alphabet from its code lengths: counts, first codes, offsets, symbols in order, and the direct table (the wave).
 * Returns false for an over-subscribed set, or an incomplete one that zlib rejects (incomplete is fine only for a single code). */
PLI_HD bool pli_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *first, uint16_t *offs, uint16_t *sym, uint32_t *tab, int tbits, bool single_ok)
{
    /* (the small serial part by every lane alike: each lane needs the results, and that costs less than handing them round) */
    uint16_t cnt[16], fst[16], off[16];
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (int i = 0; i < n; i++) cnt[lens[i] & 15]++;
    cnt[0] = 0;
    int left = 1, total = 0;
    for (int l = 1; l < 16; l++) { left = (left << 1) - (int)cnt[l]; if (left < 0) return false; total += cnt[l]; }
    if (left > 0 && !(single_ok && (total == 0 || (total == 1 && cnt[1] == 1)))) return false;     /* zlib's inftrees.c: an incomplete set passes only as no code at all or ONE code of length 1 */
    uint32_t code = 0; uint16_t o = 0;
    fst[0] = 0; off[0] = 0;
    for (int l = 1; l < 16; l++) { code = (code + cnt[l - 1]) << 1; fst[l] = (uint16_t)code; off[l] = o; o = (uint16_t)(o + cnt[l]); }
    PLI_SYNC();
    PLI_LANES(lane) {
        if (lane == 0) for (int l = 0; l < 16; l++) { count[l] = cnt[l]; first[l] = fst[l]; offs[l] = off[l]; }
        for (int i = lane; i < (1 << tbits); i += PLI_NL) tab[i] = 0u;
    }
    PLI_SYNC();
    /* a symbol a lane: its rank among the symbols of its length gives its code */
    PLI_LANES(lane) {
        for (int s = lane; s < n; s += PLI_NL) {
            const int l = lens[s] & 15;
            if (!l) continue;
            int rank = 0;
            for (int t = 0; t < s; t++) rank += (lens[t] & 15) == l;
            sym[off[l] + rank] = (uint16_t)s;
            if (l <= tbits) {
                const uint32_t c = pli_rev((uint32_t)fst[l] + (uint32_t)rank, l);
                for (uint32_t i = c; i < (1u << tbits); i += 1u << l) tab[i] = (uint32_t)s | ((uint32_t)l << 16);
            }
        }
    }
    PLI_SYNC();
    return true;
}

PLI_HD void pli_refill(PliBits &b, const PliShared &S)
{
    if (b.n <= 32 && b.ip + 4u <= b.iend) {
        const uint32_t i = b.ip >> 2;
        const uint64_t two = (uint64_t)S.in32[i] | ((uint64_t)S.in32[i + 1] << 32);      /* (the stage has 16 bytes of slack behind it) */
        b.buf |= ((two >> (8u * (b.ip & 3u))) & 0xffffffffull) << b.n;
        b.ip += 4u; b.n += 32;
    }
    while (b.n <= 56 && b.ip < b.iend && b.iend - b.ip < 4u) { b.buf |= (uint64_t)S.in[b.ip++] << b.n; b.n += 8; }
}

/* one symbol: direct table, else canonically bit by bit (lane 0).  DIST: the distance alphabet.  -1: no such code / out of input */
template <bool DIST>
PLI_HD int pli_symbol(PliBits &b, const PliShared &S)
{
    pli_refill(b, S);
    const uint32_t e = DIST ? S.dtab[pli_peek(b, PLI_DBITS)] : S.ltab[pli_peek(b, PLI_LBITS)];
    if (e) { const int l = (int)(e >> 16); if (l > b.n) return -1; pli_drop(b, l); return (int)(e & 0xffffu); }
    uint32_t code = 0;
    for (int l = 1; l < 16; l++) {
        if (b.n < l) return -1;
        code = (code << 1) | (uint32_t)((b.buf >> (l - 1)) & 1u);
        const uint32_t first = DIST ? S.dfirst[l] : S.lfirst[l], count = DIST ? S.dcount[l] : S.lcount[l];
        const uint32_t d = code - first;
        if (code >= first && d < count) { pli_drop(b, l); return (int)(DIST ? S.dsym[S.doffs[l] + d] : S.lsym[S.loffs[l] + d]); }
    }
    return -1;
}

/* the stream `st` by the lanes of one wave (device) / by lane loops (host); S: the stream's shared memory */
PLI_HD void pli_inflate(const PliStream &st, PliShared &S)
{
    static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    static const uint8_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    static const uint8_t clorder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    const uint8_t *z = st.z;
    const uint32_t zbytes = st.zbytes, expect = st.expect;
    /* state of lane 0 (registers on the device; on the host the lane loops below run lane 0 only where it matters) */
    PliBits B; B.buf = 0; B.n = 0; B.ip = 0; B.iend = 0;
    uint32_t zpos = 0;          /* bytes of the stream staged so far */
    uint32_t pos = 0;           /* bytes produced */
    uint32_t flushed = 0;       /* bytes that have left the window */
    uint32_t s1 = 1, s2 = 0;    /* Adler-32 of what has left */
    int err = PLI_OK;
    bool final_block = false, done = false;

    /* -- stage input: the unread tail moves to the front, the lanes load what follows it -- */
    auto stage = [&]() {
        uint32_t keep = 0;
        PLI_SYNC();
        /* (every lane computes the same numbers from tok: lane 0 published its read position) */
        keep = S.tok[6] <= S.tok[7] ? S.tok[7] - S.tok[6] : 0u;          /* iend - ip */
        const uint32_t ip = S.tok[6];
        if (keep && ip) {
            /* move in pieces the lanes can do without overlapping each other: keep <= 16 here in practice (the decoder asks when it runs dry) */
            PLI_LANE0(lane) { for (uint32_t i = 0; i < keep; i++) S.in[i] = S.in[ip + i]; }
        }
        PLI_SYNC();
        const uint32_t room = PLI_IN - keep, take = zbytes - zpos < room ? zbytes - zpos : room;
        PLI_LANES(lane) { for (uint32_t i = (uint32_t)lane; i < take; i += PLI_NL) S.in[keep + i] = z[zpos + i]; }
        zpos += take;
        PLI_SYNC();
        B.ip = 0; B.iend = keep + take;
    };
    /* -- a piece leaves the window: bytes [flushed, flushed + n) -- */
    auto flush = [&](uint32_t n) {
        PLI_SYNC();
        PLI_LANES(lane) { if (lane == 0) { S.red[0] = 0; S.red[1] = 0; } }
        PLI_SYNC();
        PLI_LANES(lane) {
            /* lane l: bytes [64 l, 64 l + 64) of the piece */
            uint64_t a = 0, w = 0;
            const uint32_t lo = (uint32_t)lane * 64u;
            for (uint32_t i = 0; i < 64u && lo + i < n; i++) {
                const uint32_t g = lo + i;
                const uint8_t v = S.win[(flushed + g) & (PLI_WIN - 1)];
                if (flushed + g < expect) st.out[flushed + g] = v;
                a += v; w += (uint64_t)(n - g) * v;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            a = pli_wave_sum(a); w = pli_wave_sum(w);
            if (lane == 0) { S.red[0] = a; S.red[1] = w; }
#else
            S.red[0] += a; S.red[1] += w;
#endif
        }
        PLI_SYNC();
        s2 = (uint32_t)((s2 + (uint64_t)n * s1 + S.red[1]) % 65521u);
        s1 = (uint32_t)((s1 + S.red[0]) % 65521u);
        flushed += n;
        PLI_SYNC();
    };

    PLI_LANES(lane) { if (lane < 29) { S.lbase[lane] = lbase[lane]; S.lext[lane] = lext[lane]; } if (lane < 30) { S.dbase[lane] = dbase[lane]; S.dext[lane] = dext[lane]; } }
    PLI_LANE0(lane) { S.tok[6] = 0; S.tok[7] = 0; S.tok[3] = 0; }
    stage();
    /* zlib header (RFC 1950): deflate, window <= 32K, no preset dictionary, check bits */
    PLI_LANE0(lane) {
        pli_refill(B, S);
        if (B.n < 16) S.tok[3] = PLI_E_HEADER;
        else {
            const uint32_t cmf = pli_peek(B, 8), flg = (uint32_t)((B.buf >> 8) & 255u);
            if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 32u) || ((cmf << 8) | flg) % 31u) S.tok[3] = PLI_E_HEADER;
            pli_drop(B, 16);
        }
    }
    PLI_SYNC();
    err = (int)S.tok[3];
    while (!err && !done) {
        /* ---- block header (lane 0); a dynamic block's code lengths too ---- */
        PLI_LANE0(lane) {
            S.tok[0] = 0; S.tok[3] = 0;
            pli_refill(B, S);
            if (B.n < 3) {
                if (zpos < zbytes) { S.tok[0] = PLI_T_INPUT; }
                else S.tok[3] = PLI_E_INPUT;
            } else {
                /* (a dynamic header can be ~300 bytes: make sure the stage holds it, else ask for input first) */
                if (B.iend - B.ip < 400u && zpos < zbytes) S.tok[0] = PLI_T_INPUT;
                else {
                    const uint32_t bfinal = pli_peek(B, 1), btype = (uint32_t)((B.buf >> 1) & 3u);
                    pli_drop(B, 3);
                    S.tok[1] = bfinal; S.tok[2] = btype;
                    if (btype == 3u) S.tok[3] = PLI_E_BLOCK;
                    else if (btype == 0u) {
                        pli_drop(B, B.n & 7);                       /* to the byte boundary */
                        pli_refill(B, S);
                        if (B.n < 32) S.tok[3] = PLI_E_INPUT;
                        else {
                            const uint32_t len = pli_peek(B, 16), nlen = (uint32_t)((B.buf >> 16) & 0xffffu);
                            pli_drop(B, 32);
                            if ((len ^ nlen) != 0xffffu) S.tok[3] = PLI_E_BLOCK;
                            S.tok[4] = len;
                            /* whole bytes still in the bit buffer go back to the stage */
                            B.ip -= (uint32_t)(B.n >> 3); B.buf = 0; B.n = 0;
                        }
                    } else if (btype == 1u) {
                        for (int i = 0; i < 144; i++) S.lens[i] = 8;
                        for (int i = 144; i < 256; i++) S.lens[i] = 9;
                        for (int i = 256; i < 280; i++) S.lens[i] = 7;
                        for (int i = 280; i < 288; i++) S.lens[i] = 8;
                        for (int i = 0; i < 32; i++) S.lens[288 + i] = 5;         /* (30 and 31 never occur in a valid stream: checked at the symbol) */
                        S.tok[4] = 288; S.tok[5] = 32;
                    } else {
                        pli_refill(B, S);
                        const uint32_t hlit = pli_peek(B, 5) + 257u, hdist = (uint32_t)((B.buf >> 5) & 31u) + 1u, hclen = (uint32_t)((B.buf >> 10) & 15u) + 4u;
                        pli_drop(B, 14);
                        if (hlit > 286u || hdist > 30u) S.tok[3] = PLI_E_CODES;
                        else {
                            /* the code-length code: 19 symbols of 3 bits, a 7-bit table in the distance table's place */
                            uint8_t cl[19];
                            for (int i = 0; i < 19; i++) cl[i] = 0;
                            for (uint32_t i = 0; i < hclen; i++) { pli_refill(B, S); cl[clorder[i]] = (uint8_t)pli_peek(B, 3); pli_drop(B, 3); }
                            int left = 1, tot = 0; uint16_t cnt[8], fst[8];
                            for (int l = 0; l < 8; l++) cnt[l] = 0;
                            for (int i = 0; i < 19; i++) cnt[cl[i]]++;
                            cnt[0] = 0;
                            for (int l = 1; l < 8; l++) { left = (left << 1) - (int)cnt[l]; tot += cnt[l]; }
                            if (left < 0 || (left > 0 && tot != 1)) S.tok[3] = PLI_E_CODES;
                            else {
                                uint32_t code = 0;
                                for (int l = 1; l < 8; l++) { code = (code + cnt[l - 1]) << 1; fst[l] = (uint16_t)code; }
                                for (int i = 0; i < 128; i++) S.dtab[i] = 0u;
                                uint16_t nxt[8];
                                for (int l = 1; l < 8; l++) nxt[l] = fst[l];
                                for (int s = 0; s < 19; s++) {
                                    const int l = cl[s];
                                    if (!l) continue;
                                    const uint32_t c = pli_rev(nxt[l]++, l);
                                    for (uint32_t i = c; i < 128u; i += 1u << l) S.dtab[i] = (uint32_t)s | ((uint32_t)l << 16);
                                }
                                uint32_t i = 0; const uint32_t nn = hlit + hdist;
                                int prevl = 0;
                                while (i < nn && !S.tok[3]) {
                                    pli_refill(B, S);
                                    const uint32_t e = S.dtab[pli_peek(B, 7)];
                                    if (!e || (int)(e >> 16) > B.n) { S.tok[3] = PLI_E_CODES; break; }
                                    pli_drop(B, (int)(e >> 16));
                                    const uint32_t s = e & 0xffffu;
                                    if (s < 16u) { S.lens[i++] = (uint8_t)s; prevl = (int)s; }
                                    else {
                                        uint32_t rep; int v;
                                        if (s == 16u) { if (!i) { S.tok[3] = PLI_E_CODES; break; } rep = 3u + pli_peek(B, 2); pli_drop(B, 2); v = prevl; }
                                        else if (s == 17u) { rep = 3u + pli_peek(B, 3); pli_drop(B, 3); v = 0; }
                                        else { rep = 11u + pli_peek(B, 7); pli_drop(B, 7); v = 0; }
                                        if (B.n < 0 || i + rep > nn) { S.tok[3] = PLI_E_CODES; break; }
                                        while (rep--) S.lens[i++] = (uint8_t)v;
                                        prevl = v;
                                    }
                                }
                                if (!S.tok[3] && S.lens[256] == 0) S.tok[3] = PLI_E_CODES;         /* no end-of-block code */
                                /* the distance lengths behind the literal/length ones at a fixed place */
                                if (!S.tok[3]) {
                                    uint8_t dl[32];
                                    for (uint32_t k = 0; k < 32u; k++) dl[k] = k < hdist ? S.lens[hlit + k] : 0;
                                    for (uint32_t k = hlit; k < 288u; k++) S.lens[k] = 0;
                                    for (uint32_t k = 0; k < 32u; k++) S.lens[288 + k] = dl[k];
                                }
                                S.tok[4] = hlit; S.tok[5] = hdist;
                            }
                        }
                    }
                }
            }
            S.tok[6] = B.ip; S.tok[7] = B.iend;
            if (B.n < 0) S.tok[3] = PLI_E_INPUT;
        }
        PLI_SYNC();
        if (S.tok[3]) { err = (int)S.tok[3]; break; }
        if (S.tok[0] == PLI_T_INPUT) {
            /* lane 0 gives the bits it holds back: whole bytes by position; of a byte read in part, the unread bits are taken again behind the staging */
            PLI_LANE0(lane) {
                B.ip -= (uint32_t)(B.n >> 3);
                const int frac = B.n & 7;
                if (frac) B.ip -= 1;
                S.tok[6] = B.ip; S.tok[7] = B.iend; S.tok[5] = (uint32_t)frac;
            }
            PLI_SYNC();
            const uint32_t frac = S.tok[5];
            stage();
            PLI_LANE0(lane) { B.buf = 0; B.n = 0; if (frac) { pli_refill(B, S); pli_drop(B, 8 - (int)frac); } }
            continue;
        }
        final_block = S.tok[1] != 0;
        const uint32_t btype = S.tok[2];
        if (btype == 0u) {
            /* ---- stored: LEN bytes from the stream into the window, piece by piece ---- */
            uint32_t len = S.tok[4];
            if (pos + len > expect) { err = PLI_E_SIZE; break; }
            while (len) {
                const uint32_t avail = S.tok[7] - S.tok[6];
                if (!avail) {
                    if (zpos >= zbytes) { err = PLI_E_INPUT; break; }
                    stage();
                    PLI_LANE0(lane) { S.tok[6] = B.ip; S.tok[7] = B.iend; }
                    PLI_SYNC();
                    continue;
                }
                uint32_t n = len < avail ? len : avail;
                const uint32_t to_piece = PLI_PIECE - (pos & (PLI_PIECE - 1));
                if (n > to_piece) n = to_piece;
                const uint32_t ip = S.tok[6];
                PLI_LANES(lane) { for (uint32_t i = (uint32_t)lane; i < n; i += PLI_NL) S.win[(pos + i) & (PLI_WIN - 1)] = S.in[ip + i]; }
                pos += n; len -= n;
                PLI_SYNC();
                PLI_LANE0(lane) { B.ip += n; S.tok[6] = B.ip; }
                PLI_SYNC();
                if ((pos & (PLI_PIECE - 1)) == 0) flush(PLI_PIECE);
            }
            if (err) break;
        } else {
            /* ---- Huffman block: tables by the wave, symbols by lane 0, copies by the wave ---- */
            const bool ok = pli_build(S.lens, 288, S.lcount, S.lfirst, S.loffs, S.lsym, S.ltab, PLI_LBITS, false)
                            && pli_build(S.lens + 288, 32, S.dcount, S.dfirst, S.doffs, S.dsym, S.dtab, PLI_DBITS, true);
            if (!ok) { err = PLI_E_CODES; break; }
            bool end = false;
            while (!end && !err) {
                PLI_LANE0(lane) {
                    S.tok[0] = 0; S.tok[3] = 0;
                    for (;;) {
                        /* (input: a symbol with its extra bits is at most 48 bits; refills stop at the stage's end) */
                        if (B.iend - B.ip < 8u && zpos < zbytes) { S.tok[0] = PLI_T_INPUT; break; }
                        const int s = pli_symbol<false>(B, S);
                        if (s < 0) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_SYMBOL; break; }
                        if (s < 256) {
                            if (pos >= expect) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_SIZE; break; }
                            S.win[pos & (PLI_WIN - 1)] = (uint8_t)s;
                            pos++;
                            if ((pos & (PLI_PIECE - 1)) == 0) { S.tok[0] = PLI_T_FLUSH; break; }
                            continue;
                        }
                        if (s == 256) { S.tok[0] = PLI_T_END; break; }
                        if (s > 285) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_SYMBOL; break; }
                        pli_refill(B, S);
                        const uint32_t lx = S.lext[s - 257];
                        const uint32_t len = S.lbase[s - 257] + pli_peek(B, (int)lx);
                        pli_drop(B, (int)lx);
                        const int d = pli_symbol<true>(B, S);
                        if (d < 0 || d > 29) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_SYMBOL; break; }
                        pli_refill(B, S);
                        const uint32_t dx = S.dext[d];
                        const uint32_t dist = S.dbase[d] + pli_peek(B, (int)dx);
                        pli_drop(B, (int)dx);
                        if (B.n < 0) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_INPUT; break; }
                        if (dist > pos) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_DIST; break; }
                        if (pos + len > expect) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_SIZE; break; }
                        if (len < PLI_WAVE_COPY) {
                            /* a short match (most are): by lane 0 itself, byte after byte (an overlap copies what it has just written) -- handing it to
                             * the wave costs two barriers and a round of tokens, ~1500 cycles against ~70 a byte here */
                            for (uint32_t k = 0; k < len; k++) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + k) & (PLI_WIN - 1)];
                            const uint32_t before = pos / PLI_PIECE;
                            pos += len;
                            if (pos / PLI_PIECE != before) { S.tok[0] = PLI_T_FLUSH; break; }
                            continue;
                        }
                        S.tok[0] = PLI_T_MATCH; S.tok[1] = len; S.tok[2] = dist;
                        break;
                    }
                    S.tok[4] = pos;
                    if (B.n < 0 && !S.tok[3]) { S.tok[0] = PLI_T_ERROR; S.tok[3] = PLI_E_INPUT; }
                }
                PLI_SYNC();
                const uint32_t kind = S.tok[0];
                pos = S.tok[4];
                if (kind == PLI_T_ERROR) { err = (int)S.tok[3]; break; }
                if (kind == PLI_T_END) { end = true; break; }
                if (kind == PLI_T_FLUSH) { flush(PLI_PIECE); continue; }
                if (kind == PLI_T_INPUT) {
                    PLI_LANE0(lane) {
                        B.ip -= (uint32_t)(B.n >> 3);
                        const int frac = B.n & 7;
                        if (frac) B.ip -= 1;
                        S.tok[6] = B.ip; S.tok[7] = B.iend; S.tok[5] = (uint32_t)frac;
                    }
                    PLI_SYNC();
                    const uint32_t frac = S.tok[5];
                    stage();
                    PLI_LANE0(lane) { B.buf = 0; B.n = 0; if (frac) { pli_refill(B, S); pli_drop(B, 8 - (int)frac); } }
                    continue;
                }
                /* a match: a byte a lane; source pos - dist + (k mod dist) lies behind pos whatever the overlap */
                {
                    const uint32_t len = S.tok[1], dist = S.tok[2];
                    PLI_LANES(lane) {
                        /* dist + len beyond the window: slot (pos + k) & mask of a late byte IS the source slot of an earlier one (zlib never emits
                         * such a match -- its distances end at 32506 --, libdeflate and zopfli do).  The wave's lock step would still read before it
                         * writes, the host harness (a lane after the other) would not: one lane copies in order, which is LZ77's own definition */
                        if (dist + len > PLI_WIN) { if (lane == 0) for (uint32_t k = 0; k < len; k++) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + k) & (PLI_WIN - 1)]; }
                        else if (dist >= len) { for (uint32_t k = (uint32_t)lane; k < len; k += PLI_NL) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + k) & (PLI_WIN - 1)]; }
                        else { for (uint32_t k = (uint32_t)lane; k < len; k += PLI_NL) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + (k % dist)) & (PLI_WIN - 1)]; }
                    }
                    PLI_SYNC();
                    const uint32_t before = pos / PLI_PIECE;
                    pos += len;
                    if (pos / PLI_PIECE != before) flush(PLI_PIECE);
                }
            }
            if (err) break;
        }
        if (final_block) done = true;
    }
    if (!err) {
        if (pos != expect) err = PLI_E_SIZE;
        else {
            if (pos > flushed) flush(pos - flushed);
            /* Adler-32 trailer: big endian, behind the last block at the next byte boundary */
            PLI_LANE0(lane) {
                pli_drop(B, B.n & 7);
                uint32_t want = 0; bool have = true;
                for (int i = 0; i < 4; i++) {
                    pli_refill(B, S);
                    if (B.n < 8) { have = false; break; }
                    want = (want << 8) | pli_peek(B, 8); pli_drop(B, 8);
                }
                S.tok[3] = (have && want == ((s2 << 16) | s1)) ? 0u : (uint32_t)(have ? PLI_E_ADLER : PLI_E_INPUT);
            }
            PLI_SYNC();
            err = (int)S.tok[3];
            /* (a trailer that lies beyond the stage: stage once more and look again) */
            if (err == PLI_E_INPUT && zpos < zbytes) {
                PLI_LANE0(lane) { B.ip = B.iend; S.tok[6] = B.ip; S.tok[7] = B.iend; }
                err = PLI_E_ADLER;       /* (kept simple: such a stream goes to the host reader) */
            }
        }
    }
    PLI_LANE0(lane) { *st.status = err; }
}

#endif /* PL_INFLATE_CORE_H */
