/*
 * pl_inflate_core.h -- INFLATE of a PNG's image data on the device: the zlib stream that a file's IDAT chunks concatenate to
 * (RFC 1950 wrapper, RFC 1951 blocks: stored / fixed / dynamic Huffman) -> the filtered scanlines.  Replaces what zlib does for
 * libpng inside rwpng_read_image24_libpng (/root/reference/src/rwpng.c:179-400: png_read_image -> inflate) for the files of a window --
 * independent streams, one WAVE each (the bit stream of one file is serial; a window has dozens to hundreds of files).
 *
 * One wave per stream, and (round 6) every lane of it on the bit stream:
 *   ROUNDS      the Huffman loop does not decode a symbol at a time.  Every lane looks up, for ITS bit of the next 64 (three such sets a round, 39 bits apart), the 32 bits of
 *               the stream that start there and what the two direct tables say about them; then the wave -- all lanes alike, in SCALAR registers, reading the lanes' answers
 *               with v_readlane -- follows the chain of code lengths through them: one trip to shared memory serves ~120 bits of the stream (a dozen symbols) instead of one to
 *               three trips a symbol.  A run of literals is eight scalar instructions a literal (its bytes leave together, written by the lanes that hold them); a plain match
 *               (both codes in the direct tables, <= 64 bytes) is one straight piece of code with ONE test for everything unusual, a byte a lane, its bytes read at once and
 *               written when the next reader of the window comes up (settle); everything else -- longer codes, long matches, the block's end, whatever is wrong -- takes a
 *               general path that also words the errors.  What a lone wave pays for is dependent steps and taken branches (~10 cycles an instruction as measured), not
 *               arithmetic: the layout of that loop was measured (profiles/r06_inflate.txt).
 *   all lanes   input staging (the compressed bytes come through a 4 KB buffer in shared memory), table construction (a symbol a lane), stored blocks, and the way out:
 *               the 32 KB window lives in shared memory and leaves for device memory in 4 KB pieces, 64 bytes a lane, with the piece's share of the Adler-32 (RFC 1950)
 *               computed on the way.
 *   lane 0      block headers (a dynamic block's code lengths), the zlib header and trailer.
 * Tables: an 11-bit direct table for literal/length codes and a 9-bit one for distances (pli_entry: code length, extra bits, kind, literal or base -- everything a symbol needs in one
 * look-up); longer codes (1 - 3 % of a photograph's symbols) are decoded canonically from per-length first codes held in lanes 0..15 (pli_long).
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
/* a value a lane (a register) and its uniform reads: PLI_AT(v, k) = lane k's value (k uniform), PLI_U(x) = a value every lane holds alike, made a scalar */
#define PLI_VEC(T, name) T name
#define PLI_VARG(T) const T
#define PLI_V(name, lane) name
#define PLI_AT(name, k) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(k)))
#define PLI_U(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define PLI_RCP(d) __builtin_amdgcn_rcpf((float)(d))
#define PLI_BREV32(x) __builtin_bitreverse32(x)
/* what the lanes of the wave have written to shared memory is what its lanes read from here on (the hardware runs a wave's shared-memory instructions in order; this tells the compiler) */
#define PLI_WAVE_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define PLI_WAVE_ORDER() ((void)0)
#define PLI_VEC(T, name) T name[PLI_NL]
#define PLI_VARG(T) const T *
#define PLI_V(name, lane) name[lane]
#define PLI_AT(name, k) ((uint32_t)(name)[k])
#define PLI_U(x) ((uint32_t)(x))
#define PLI_RCP(d) (1.0f / (float)(d))
#define PLI_BREV32(x) pli_brev32_host(x)
static inline uint32_t pli_brev32_host(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
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

#ifndef PLI_STAT
#define PLI_STAT(i, n)            /* (the CPU harness can count rounds, sets, literals, matches, matched bytes, symbols with a long code: tests/c/inflate_host.cpp) */
#endif
#define PLI_WIN 32768u            /* the window (RFC 1951: distances up to 32768) */
#define PLI_PIECE 4096u           /* what leaves the window at a time */
#ifndef PLI_IN
#define PLI_IN 4096u              /* staged input (4 KB against 16: 50 KB of shared memory a stream instead of 62 = THREE streams a CU; one stream alone 10.0 MB/s either way, 768 files
                                     of 1280x720 in one call 452 ms against 813: profiles/r06_inflate.txt) */
#endif
#ifndef PLI_LBITS
#define PLI_LBITS 11             /* (10 against 11, the suite's files one stream at a time: 9.1 / 9.4 / 8.7 / 14.6 MB/s against 9.2 / 9.5 / 9.5 / 15.9 for barbara / lena / ssr / tenko: profiles/r06_inflate.txt) */
#endif
#define PLI_ROUND_BITS (63u - (PLI_LBITS + 5u + PLI_DBITS))    /* a set of 64 looked-up positions serves the symbols that START at most this many bits into it (38): a length /
                                     distance pair through the direct tables is at most 11 + 5 + 9 bits of codes and extras in front of its last field */
#define PLI_SETS 3                /* sets of positions a round looks up at once, PLI_ROUND_BITS + 1 bits apart: one wait for shared memory serves ~120 bits of the stream */
#define PLI_DBITS 9
enum { PLI_OK = 0, PLI_E_HEADER = 1, PLI_E_BLOCK = 2, PLI_E_CODES = 3, PLI_E_SYMBOL = 4, PLI_E_DIST = 5, PLI_E_SIZE = 6, PLI_E_ADLER = 7, PLI_E_INPUT = 8 };

/* shared memory of one stream */
struct PliShared {
    uint8_t win[PLI_WIN];
    union { uint8_t in[PLI_IN + 32]; uint32_t in32[PLI_IN / 4 + 8]; };
    uint32_t ltab[1u << PLI_LBITS];   /* symbol | length << 16; 0: a longer code */
    uint32_t dtab[1u << PLI_DBITS];
    uint16_t lcount[16], dcount[16];  /* codes per length */
    uint16_t lfirst[16], dfirst[16];  /* first code of a length (canonical) */
    uint16_t loffs[16], doffs[16];    /* index of a length's first symbol in lsym / dsym */
    uint16_t lsym[288], dsym[32];     /* symbols in canonical order */
    uint8_t lens[320];                /* code lengths of the block: literal/length codes, then distance codes */
    uint16_t lbase[32], dbase[32];    /* RFC 1951 3.2.5: base of a length / distance code and its extra bits (from constant memory once: a lookup there */
    uint8_t lext[32], dext[32];       /*  costs a round trip to device memory per use) */
    uint32_t inv[260];                /* [d] = 2^20 / d + 1: k / d = k * inv[d] >> 20 for k < 258 (a match that overlaps itself: source index k mod d) */
    uint32_t tok[8];                  /* lane 0 -> wave: [0] what it asks for, [1] final block, [2] block type, [3] error, [4] hlit / stored length, [5] hdist, [6] [7] read position */
    uint64_t red[2];
};

enum { PLI_T_INPUT = 4 };

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
/* an entry of the direct tables: bits 0..3 the code's length, 4..7 extra bits that follow it, 8..9 what it is, 10 no such entry (PLI_NODIRECT), 16..31 the literal / the base of the
 * length or distance -- everything a symbol needs in ONE lookup (the base and extra-bit tables are read when the block's tables are built, not per symbol) */
enum { PLI_K_LIT = 0, PLI_K_BASE = 1, PLI_K_END = 2, PLI_K_BAD = 3 };
#define PLI_SETEND 0x800u         /* (not in the tables: what the lanes behind PLI_ROUND_BITS add to what they looked up, so that the loop over a run of literals tests one thing) */
#define PLI_NODIRECT 0x400u       /* the entry of an index whose code is longer than the table's (or that no code has): a literal through the table is (entry & 0x700) == 0 */
template <bool DIST>
PLI_HD uint32_t pli_entry(int s, int l, const uint16_t *base, const uint8_t *ext)
{
    if (DIST) return s > 29 ? (uint32_t)l | (PLI_K_BAD << 8) : (uint32_t)l | ((uint32_t)ext[s] << 4) | (PLI_K_BASE << 8) | ((uint32_t)base[s] << 16);
    if (s < 256) return (uint32_t)l | (PLI_K_LIT << 8) | ((uint32_t)s << 16);
    if (s == 256) return (uint32_t)l | (PLI_K_END << 8);
    if (s > 285) return (uint32_t)l | (PLI_K_BAD << 8);
    return (uint32_t)l | ((uint32_t)ext[s - 257] << 4) | (PLI_K_BASE << 8) | ((uint32_t)base[s - 257] << 16);
}
template <bool DIST>
PLI_HD bool pli_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *first, uint16_t *offs, uint16_t *sym, uint32_t *tab, int tbits, bool single_ok, const uint16_t *base, const uint8_t *ext)
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
        for (int i = lane; i < (1 << tbits); i += PLI_NL) tab[i] = PLI_NODIRECT;
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
                const uint32_t e = pli_entry<DIST>(s, l, base, ext);
                for (uint32_t i = c; i < (1u << tbits); i += 1u << l) tab[i] = e;
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

/* a code longer than the direct table's index (1 - 3 % of a photograph's symbols): canonically, length by length from the table's index size + 1 on, by every lane alike --
 * first code, count and first symbol of every length sit in lanes 0..15 of two registers (fc = first | count << 16, of = index of the length's first symbol), read with
 * PLI_AT, so that a length costs eight scalar instructions and no trip to shared memory; the symbol itself is one.  DIST: the distance alphabet.  -1: no such code */
template <bool DIST, typename VFC, typename VOF>
PLI_HD int pli_long(uint64_t buf, const PliShared &S, VFC fc, VOF of, int &lout)
{
    const uint32_t rev = PLI_BREV32((uint32_t)buf);                 /* a code's first bit is the stream's lowest */
    for (int l = (DIST ? PLI_DBITS : PLI_LBITS) + 1; l < 16; l++) {
        const uint32_t code = rev >> (32 - l), f = PLI_AT(fc, l), first = f & 0xffffu, count = f >> 16;
        const uint32_t d = code - first;
        if (code >= first && d < count) { lout = l; const uint32_t o = PLI_AT(of, l); return (int)PLI_U(DIST ? S.dsym[o + d] : S.lsym[o + d]); }
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
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lanebit_lo = threadIdx.x < 32u ? 1u << threadIdx.x : 0u, lanebit_hi = threadIdx.x < 32u ? 0u : 1u << (threadIdx.x - 32u);
#endif
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
        const uint32_t ip = PLI_U(S.tok[6]), ie = PLI_U(S.tok[7]);
        keep = ip <= ie ? ie - ip : 0u;
        if (keep && ip) {
            /* move in pieces the lanes can do without overlapping each other: keep <= 16 here in practice (the decoder asks when it runs dry) */
            PLI_LANE0(lane) { for (uint32_t i = 0; i < keep; i++) S.in[i] = S.in[ip + i]; }
        }
        PLI_SYNC();
        const uint32_t room = PLI_IN - keep, take = zbytes - zpos < room ? zbytes - zpos : room;
        PLI_LANES(lane) {
            for (uint32_t i = (uint32_t)lane; i < take; i += PLI_NL) S.in[keep + i] = z[zpos + i];
            if (lane < 32) S.in[keep + take + (uint32_t)lane] = 0;           /* (the Huffman loop's lanes look up to 26 bytes beyond the byte they are at: zeros behind the stream's end) */
        }
        zpos += take;
        PLI_SYNC();
        B.ip = 0; B.iend = keep + take;
    };
    /* -- a short match's bytes are READ where it is decoded and WRITTEN when the next thing that could read them comes up (the next match, a piece leaving the window, the block's
     * end): the wave goes on decoding while the read is under way instead of waiting for it.  What is decoded in between writes elsewhere (literals lie behind the match). -- */
    PLI_VEC(uint32_t, pend_a); PLI_VEC(uint32_t, pend_v);
    PLI_LANES(lane) { (void)lane; PLI_V(pend_a, lane) = 0; PLI_V(pend_v, lane) = 0; }
    uint32_t pend_n = 0;                                   /* lanes below hold a byte and its place */
    auto settle = [&]() {
        if (!pend_n) return;
        PLI_LANES(lane) { if ((uint32_t)lane < pend_n) S.win[PLI_V(pend_a, lane)] = (uint8_t)PLI_V(pend_v, lane); }
        pend_n = 0;
    };
    /* -- a piece leaves the window: bytes [flushed, flushed + n) -- */
    auto flush = [&](uint32_t n) {
        settle();
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
    PLI_LANES(lane) { for (uint32_t d = 1u + (uint32_t)lane; d < 260u; d += PLI_NL) S.inv[d] = (1u << 20) / d + 1u; }
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
            /* ---- Huffman block: tables by the wave, then ROUNDS: every lane looks up the codes that would start at its bit of the next 64, and the wave -- all lanes alike, in scalar
             * registers -- follows the chain of code lengths through what the lanes found: one trip to shared memory for ~5 symbols instead of one (or three) a symbol ---- */
            const bool ok = pli_build<false>(S.lens, 288, S.lcount, S.lfirst, S.loffs, S.lsym, S.ltab, PLI_LBITS, false, S.lbase, S.lext)
                            && pli_build<true>(S.lens + 288, 32, S.dcount, S.dfirst, S.doffs, S.dsym, S.dtab, PLI_DBITS, true, S.dbase, S.dext);
            if (!ok) { err = PLI_E_CODES; break; }
            /* (the canonical code of both alphabets by length, in lanes 0..15: pli_long) */
            PLI_VEC(uint32_t, cLfc); PLI_VEC(uint32_t, cLof); PLI_VEC(uint32_t, cDfc); PLI_VEC(uint32_t, cDof);
            PLI_LANES(lane) {
                const int l = lane & 15;
                PLI_V(cLfc, lane) = (uint32_t)S.lfirst[l] | ((uint32_t)S.lcount[l] << 16); PLI_V(cLof, lane) = S.loffs[l];
                PLI_V(cDfc, lane) = (uint32_t)S.dfirst[l] | ((uint32_t)S.dcount[l] << 16); PLI_V(cDof, lane) = S.doffs[l];
            }
            /* lane 0's reader hands its position over: bit bp of the stage */
            PLI_LANE0(lane) { S.tok[6] = B.ip * 8u - (uint32_t)B.n; S.tok[7] = B.iend; }
            PLI_SYNC();
            uint32_t bp = PLI_U(S.tok[6]), iend = PLI_U(S.tok[7]);
            bool end = false;
            while (!end && !err) {
                if (iend * 8u < bp + 256u && zpos < zbytes) {
                    PLI_SYNC();
                    PLI_LANE0(lane) { S.tok[6] = bp >> 3; S.tok[7] = iend; }
                    stage();
                    bp &= 7u; iend = PLI_U(B.iend);
                    continue;
                }
                if (bp > iend * 8u) { err = PLI_E_INPUT; break; }
                /* every lane looks up what the stream holds from its bit on, in PLI_SETS sets of 64 positions: 32 bits of the stream, and what the two direct tables say about them */
                PLI_VEC(uint32_t, w0); PLI_VEC(uint32_t, vL0); PLI_VEC(uint32_t, vD0);
                PLI_VEC(uint32_t, w1); PLI_VEC(uint32_t, vL1); PLI_VEC(uint32_t, vD1);
                PLI_VEC(uint32_t, w2); PLI_VEC(uint32_t, vL2); PLI_VEC(uint32_t, vD2);
                PLI_LANES(lane) {
                    uint32_t ww[PLI_SETS];
                    const uint32_t setend = (uint32_t)lane > PLI_ROUND_BITS ? PLI_SETEND : 0u;
                    for (int q = 0; q < PLI_SETS; q++) {
                        const uint32_t bit = bp + (uint32_t)q * (PLI_ROUND_BITS + 1u) + (uint32_t)lane, i = bit >> 5;
                        const uint64_t two = (uint64_t)S.in32[i] | ((uint64_t)S.in32[i + 1] << 32);
                        ww[q] = (uint32_t)(two >> (bit & 31u));
                    }
                    PLI_V(w0, lane) = ww[0]; PLI_V(vL0, lane) = S.ltab[ww[0] & ((1u << PLI_LBITS) - 1u)] | setend; PLI_V(vD0, lane) = S.dtab[ww[0] & ((1u << PLI_DBITS) - 1u)];
                    PLI_V(w1, lane) = ww[1]; PLI_V(vL1, lane) = S.ltab[ww[1] & ((1u << PLI_LBITS) - 1u)] | setend; PLI_V(vD1, lane) = S.dtab[ww[1] & ((1u << PLI_DBITS) - 1u)];
                    PLI_V(w2, lane) = ww[2]; PLI_V(vL2, lane) = S.ltab[ww[2] & ((1u << PLI_LBITS) - 1u)] | setend; PLI_V(vD2, lane) = S.dtab[ww[2] & ((1u << PLI_DBITS) - 1u)];
                }
                PLI_STAT(0, 1);
                const uint32_t avail = iend * 8u - bp;
                uint32_t done_bits = 0;                 /* bits of the stream this round has decoded */
                bool more = true;                       /* the round goes on with the next set */
                /* the chain through ONE set: `off` counts from the set's first position */
                auto chain = [&](PLI_VARG(uint32_t) w, PLI_VARG(uint32_t) vL, PLI_VARG(uint32_t) vD, const uint32_t base) __attribute__((always_inline)) {
                uint32_t off = done_bits - base;
                PLI_STAT(1, 1);
                const uint32_t avail_here = avail - base;      /* (avail >= base: the bits in front of this set were checked against it) */
                more = false;
                /* literals are not written one by one: the bits they start at are collected (a mask over the lanes) and the lanes that hold them write the run at once -- a literal
                 * costs the chain ten scalar instructions, its byte leaves with the run's */
                uint64_t run = 0;
                auto emit_run = [&]() {
                    if (!run) return;
                    uint32_t n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
                    /* (a lane's place in the run: the set bits below it -- two instructions) */
                    const uint32_t rlo = (uint32_t)run, rhi = (uint32_t)(run >> 32);
                    n = (uint32_t)__builtin_popcountll(run);
                    if ((rlo & lanebit_lo) | (rhi & lanebit_hi)) S.win[(pos + __builtin_amdgcn_mbcnt_hi(rhi, __builtin_amdgcn_mbcnt_lo(rlo, 0u))) & (PLI_WIN - 1)] = (uint8_t)(vL >> 16);
#else
                    for (int lane = 0; lane < PLI_NL; lane++) if ((run >> lane) & 1ull) { S.win[(pos + n) & (PLI_WIN - 1)] = (uint8_t)(vL[lane] >> 16); n++; }
#endif
                    PLI_STAT(2, n); PLI_STAT(6, 1);
                    run = 0;
                    if (off > avail_here) { err = PLI_E_INPUT; return; }               /* (the stage is zero behind the stream's last byte: what was decoded from there is not the stream's) */
                    if (pos + n > expect) { err = PLI_E_SIZE; return; }
                    const uint32_t before = pos / PLI_PIECE;
                    pos += n;
                    if (pos / PLI_PIECE != before) flush(PLI_PIECE);
                };
                while (off <= PLI_ROUND_BITS) {
                    uint32_t e = PLI_AT(vL, off);
                    /* (the run of literals: a loop of its own, so that it compiles to the eight instructions it is) */
                    while (!(e & 0xf00u)) {
                        run |= 1ull << off; off += e & 15u;
                        e = PLI_AT(vL, off);                    /* (off <= PLI_ROUND_BITS + PLI_LBITS: a lane of the set; those behind PLI_ROUND_BITS carry PLI_SETEND, which ends the run) */
                    }
                    if ((e & 0xf00u) == ((uint32_t)PLI_K_BASE << 8)) {
                        /* THE PLAIN MATCH -- both codes in the direct tables, at most PLI_NL bytes, nothing wrong with it: straight through, ONE test for everything that is not
                         * plain (a lone wave pays for every branch it takes, far more than for the scalar instructions in between; anything else takes the general path below,
                         * which also words the errors).  The run of literals in front of it and the match before it reach the window, this match's bytes are read (settle
                         * writes them). */
                        const uint32_t l = e & 15u, lx = (e >> 4) & 15u, o1 = off + l, o2 = o1 + lx;
                        const uint32_t d = PLI_AT(vD, o2);
                        const uint32_t o3 = o2 + (d & 15u), dx = (d >> 4) & 15u;             /* (o3 <= PLI_ROUND_BITS + PLI_LBITS + 5 + PLI_DBITS = 63: a lane, whatever d is) */
                        const uint32_t mlen = (e >> 16) + (PLI_AT(w, o1) & ((1u << lx) - 1u));
                        const uint32_t mdist = (d >> 16) + (PLI_AT(w, o3) & ((1u << dx) - 1u));
                        const uint32_t mend = o3 + dx;
#if defined(__HIP_DEVICE_COMPILE__)
                        const uint32_t nrun = (uint32_t)__builtin_popcountll(run);
#else
                        uint32_t nrun = 0; for (int q = 0; q < 64; q++) nrun += (uint32_t)((run >> q) & 1ull);
#endif
                        const uint32_t p1 = pos + nrun, p2 = p1 + mlen;
                        /* (each term is negative exactly when its rule is broken -- every quantity is far below 2^31 --: one sign test for six rules) */
                        const uint32_t odd = (avail_here - mend) | (p1 - mdist) | (expect - p2) | (PLI_WIN - (mdist + mlen)) | ((uint32_t)PLI_NL - mlen) | (0u - ((d & 0xf00u) ^ ((uint32_t)PLI_K_BASE << 8)));
                        if (__builtin_expect((int32_t)odd < 0, 0)) goto general;
                        PLI_STAT(2, nrun); PLI_STAT(6, nrun ? 1 : 0); PLI_STAT(3, 1); PLI_STAT(4, mlen);
#if defined(__HIP_DEVICE_COMPILE__)
                        {
                            const uint32_t rlo = (uint32_t)run, rhi = (uint32_t)(run >> 32);
                            if ((rlo & lanebit_lo) | (rhi & lanebit_hi)) S.win[(pos + __builtin_amdgcn_mbcnt_hi(rhi, __builtin_amdgcn_mbcnt_lo(rlo, 0u))) & (PLI_WIN - 1)] = (uint8_t)(vL >> 16);
                            if (threadIdx.x < pend_n) S.win[pend_a] = (uint8_t)pend_v;
                        }
#else
                        { uint32_t r = 0; for (int lane = 0; lane < PLI_NL; lane++) if ((run >> lane) & 1ull) { S.win[(pos + r) & (PLI_WIN - 1)] = (uint8_t)(vL[lane] >> 16); r++; } }
                        for (uint32_t lane = 0; lane < pend_n; lane++) S.win[pend_a[lane]] = (uint8_t)pend_v[lane];
#endif
                        run = 0;
                        PLI_WAVE_ORDER();
                        const float rcp = mdist >= mlen ? 0.0f : PLI_RCP(mdist);            /* (k mod dist as below; a match that does not overlap itself: k) */
                        PLI_LANES(lane) {
                            if ((uint32_t)lane < mlen) {
                                const uint32_t q = (uint32_t)((float)lane * rcp + 0.004f);
                                PLI_V(pend_v, lane) = S.win[(p1 - mdist + ((uint32_t)lane - q * mdist)) & (PLI_WIN - 1)];
                                PLI_V(pend_a, lane) = (p1 + (uint32_t)lane) & (PLI_WIN - 1);
                            }
                        }
                        pend_n = mlen;
                        const bool piece = (pos ^ p2) >= PLI_PIECE;                        /* (at most one piece boundary in 40 + 64 bytes) */
                        pos = p2; off = mend;
                        if (__builtin_expect(piece, 0)) flush(PLI_PIECE);
                        continue;
                    }
                    if (off > PLI_ROUND_BITS) break;
                    general:
                    uint32_t used = e & 15u, len = 0, dist = 0;
                    bool direct = !(e & PLI_NODIRECT);
                    if (direct && ((e >> 8) & 3u) == PLI_K_BASE) {
                        const uint32_t o1 = off + used, lx = (e >> 4) & 15u, o2 = o1 + lx;
                        len = (e >> 16) + (PLI_AT(w, o1) & ((1u << lx) - 1u));
                        const uint32_t d = PLI_AT(vD, o2);
                        if (d & PLI_NODIRECT) direct = false;
                        else if (((d >> 8) & 3u) == PLI_K_BAD) { err = PLI_E_SYMBOL; break; }
                        else {
                            const uint32_t o3 = o2 + (d & 15u), dx = (d >> 4) & 15u;
                            dist = (d >> 16) + (PLI_AT(w, o3) & ((1u << dx) - 1u));
                            used = o3 + dx - off;
                        }
                    }
                    if (!direct) {
                        /* a code beyond a direct table: the whole symbol again from 64 bits (at most 15 + 5 + 15 + 13 are needed) -- lanes off and off + 32 hold them */
                        if (off > 31u) break;
                        PLI_STAT(5, 1);
                        const uint64_t buf = (uint64_t)PLI_AT(w, off) | ((uint64_t)PLI_AT(w, off + 32u) << 32);
                        e = PLI_U(S.ltab[(uint32_t)buf & ((1u << PLI_LBITS) - 1u)]);
                        if (e & PLI_NODIRECT) {
                            int l = 0; const int sy = pli_long<false>(buf, S, cLfc, cLof, l);
                            if (sy < 0) { err = PLI_E_SYMBOL; break; }
                            e = PLI_U(pli_entry<false>(sy, l, S.lbase, S.lext));
                        }
                        used = e & 15u;
                        if (((e >> 8) & 3u) == PLI_K_BASE) {
                            const uint32_t lx = (e >> 4) & 15u;
                            len = (e >> 16) + ((uint32_t)(buf >> used) & ((1u << lx) - 1u));
                            used += lx;
                            uint32_t d = PLI_U(S.dtab[(uint32_t)(buf >> used) & ((1u << PLI_DBITS) - 1u)]);
                            if (d & PLI_NODIRECT) {
                                int l = 0; const int sy = pli_long<true>(buf >> used, S, cDfc, cDof, l);
                                if (sy < 0) { err = PLI_E_SYMBOL; break; }
                                d = PLI_U(pli_entry<true>(sy, l, S.dbase, S.dext));
                            }
                            if (((d >> 8) & 3u) == PLI_K_BAD) { err = PLI_E_SYMBOL; break; }
                            used += d & 15u;
                            const uint32_t dx = (d >> 4) & 15u;
                            dist = (d >> 16) + ((uint32_t)(buf >> used) & ((1u << dx) - 1u));
                            used += dx;
                        }
                    }
                    emit_run();                                                          /* (what comes now may read the run's bytes, and ends the block or the stream in their order) */
                    if (err) break;
                    if (off + used > avail_here) { err = PLI_E_INPUT; break; }
                    off += used;
                    const uint32_t kind = (e >> 8) & 3u;
                    if (kind == PLI_K_LIT) {
                        /* (a literal with a long code) */
                        if (pos >= expect) { err = PLI_E_SIZE; break; }
                        PLI_LANE0(lane) { S.win[pos & (PLI_WIN - 1)] = (uint8_t)(e >> 16); }
                        pos++;
                        if ((pos & (PLI_PIECE - 1)) == 0) flush(PLI_PIECE);
                        continue;
                    }
                    if (kind == PLI_K_END) { end = true; break; }
                    if (kind == PLI_K_BAD) { err = PLI_E_SYMBOL; break; }
                    if (dist > pos) { err = PLI_E_DIST; break; }
                    if (pos + len > expect) { err = PLI_E_SIZE; break; }
                    /* the match, a byte a lane: the source pos - dist + (k mod dist) lies behind pos whatever the overlap */
                    settle();
                    PLI_WAVE_ORDER();
                    PLI_STAT(3, 1); PLI_STAT(4, len);
                    if (dist + len > PLI_WIN) {
                        /* slot (pos + k) & mask of a late byte IS the source slot of an earlier one (zlib never emits such a match -- its distances end at 32506 --, libdeflate and
                         * zopfli do): one lane copies in order, which is LZ77's own definition */
                        PLI_LANE0(lane) { for (uint32_t k = 0; k < len; k++) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + k) & (PLI_WIN - 1)]; }
                    } else if (len <= PLI_NL) {
                        /* (nearly every match: one byte a lane, no loop; one that overlaps itself takes k mod dist from k * (1 / dist) in floating point -- exact for k, dist < 64 with the
                         * bias: k / dist is either whole or at least 1/63 away from the next whole number -- instead of a table in shared memory) */
                        if (dist >= len) { PLI_LANES(lane) { if ((uint32_t)lane < len) { PLI_V(pend_v, lane) = S.win[(pos - dist + (uint32_t)lane) & (PLI_WIN - 1)]; PLI_V(pend_a, lane) = (pos + (uint32_t)lane) & (PLI_WIN - 1); } } }
                        else {
                            const float rcp = PLI_RCP(dist);
                            PLI_LANES(lane) { if ((uint32_t)lane < len) { const uint32_t q = (uint32_t)((float)lane * rcp + 0.004f); PLI_V(pend_v, lane) = S.win[(pos - dist + ((uint32_t)lane - q * dist)) & (PLI_WIN - 1)]; PLI_V(pend_a, lane) = (pos + (uint32_t)lane) & (PLI_WIN - 1); } }
                        }
                        pend_n = len;
                    } else if (dist >= len) {
                        PLI_LANES(lane) { for (uint32_t k = (uint32_t)lane; k < len; k += PLI_NL) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + k) & (PLI_WIN - 1)]; }
                    } else {
                        const uint32_t inv = PLI_U(S.inv[dist]);                           /* k / dist = k * inv >> 20 for k < 258 */
                        PLI_LANES(lane) { for (uint32_t k = (uint32_t)lane; k < len; k += PLI_NL) S.win[(pos + k) & (PLI_WIN - 1)] = S.win[(pos - dist + (k - ((k * inv) >> 20) * dist)) & (PLI_WIN - 1)]; }
                    }
                    const uint32_t before = pos / PLI_PIECE;
                    pos += len;
                    if (pos / PLI_PIECE != before) flush(PLI_PIECE);
                }
                if (!err) emit_run();
                done_bits = base + off;
                more = !err && !end && off > PLI_ROUND_BITS;
                };
                chain(w0, vL0, vD0, 0u);
                if (more && done_bits <= 2u * PLI_ROUND_BITS + 1u) chain(w1, vL1, vD1, PLI_ROUND_BITS + 1u);
                if (more && done_bits <= 3u * PLI_ROUND_BITS + 2u) chain(w2, vL2, vD2, 2u * (PLI_ROUND_BITS + 1u));
                const uint32_t off = done_bits;
                bp += off;
            }
            settle();
            if (err) break;
            /* back to lane 0's reader (a byte read in part: its unread bits are taken again) */
            PLI_LANE0(lane) {
                B.ip = bp >> 3; B.buf = 0; B.n = 0;
                if (bp & 7u) { pli_refill(B, S); pli_drop(B, (int)(bp & 7u)); }
            }
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
