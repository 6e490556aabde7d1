/*
 * pl_deflate_coop.h -- one deflate block encoded by a TEAM of threads (a 256-thread workgroup on the GPU).
 *
 * Same result, bit for bit, as the one-thread dfl_encode_block of pl_deflate_core.h (which stays as the plain
 * statement of the algorithm and as the cross-check in tests/c/deflate_host.cpp); what changes is who does the work:
 *
 *   parse      The lazy parse looks serial (where a token starts depends on the previous token) but the decision AT a
 *              token start p is a pure function of p: token(p) and next(p) = p + its length.  Token starts are the
 *              orbit of `begin` under next().  Every thread walks next() through its own chunk of the block as if its
 *              chunk start were a token start and records the visited positions in a bitmap; one thread then runs the
 *              true orbit across chunk borders -- it only has to walk until it lands on a position the chunk's owner
 *              visited too (the two walks coincide from there on; they merge within a few tokens, the way Huffman
 *              decoding self-synchronises) -- and patches the bitmap.  Tokens are then produced from the bitmap by
 *              all threads, compacted by a scan, histograms through shared-memory atomics.  The same machinery
 *              serves the parse by length and the optimal parse (whose chunks of the shortest-path pass go one per
 *              thread).
 *   codes      symbols are rank-sorted in parallel; the Huffman merge itself (<= 286 symbols) is one thread.
 *   bits       tokens are dealt to the threads in equal runs, a scan of their bit counts gives every run its bit
 *              offset, and each thread writes its run with atomic ORs into the zero-initialised output.
 *
 * Written once for both worlds: on the device DFL_SYNC is __syncthreads() and the shared state lives in LDS; on the
 * host (tests only) a team is a handful of pthreads with a barrier, or a single thread.
 */
#ifndef PL_DEFLATE_COOP_H
#define PL_DEFLATE_COOP_H

#include "pl_deflate_core.h"

#ifndef DFL_COOP_MAX_BLOCK
#define DFL_COOP_MAX_BLOCK DFL_DEFAULT_BLOCK_BYTES      /* bitmap capacity: input bytes per block */
#endif
#define DFL_COOP_MAX_THREADS 256u

typedef struct {
    uint32_t tid, nthreads;
    void (*sync)(void *);          /* host only: barrier of the team (NULL for a team of one) */
    void *sync_arg;
} dfl_team;

#if defined(__HIP_DEVICE_COMPILE__) && defined(DFL_PHASE_PROF)
#define DFL_PROF(k) do { if (t->tid == 0) { const unsigned long long now_ = wall_clock64(); dfl_prof_acc[k] += now_ - dfl_prof_last; dfl_prof_last = now_; } } while (0)
#else
#define DFL_PROF(k) do { } while (0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define DFL_SYNC(t) __syncthreads()
#define DFL_SHARED_ADD(p, v) atomicAdd((p), (v))
#define DFL_OUT_OR(p, v) atomicOr((p), (v))
#else
#define DFL_SYNC(t) do { if ((t)->sync) (t)->sync((t)->sync_arg); } while (0)
#define DFL_SHARED_ADD(p, v) __atomic_fetch_add((p), (v), __ATOMIC_RELAXED)
#define DFL_OUT_OR(p, v) __atomic_fetch_or((p), (v), __ATOMIC_RELAXED)
#endif

typedef struct {
    dfl_work w;
    uint16_t sorted[DFL_NUM_LL];                       /* symbols by (frequency, symbol) */
    union {                                             /* never live at the same time */
        uint32_t bitmap[DFL_COOP_MAX_BLOCK / 32];      /* parse: bit p-begin set: a token starts at p */
        uint16_t ring[DFL_DP_RING * DFL_COOP_MAX_THREADS];   /* optimal parse: the threads' cost windows, interleaved */
    } u;
    uint32_t exit_pos[DFL_COOP_MAX_THREADS];           /* where a thread's walk left its chunk */
    uint32_t part[DFL_COOP_MAX_THREADS + 1];           /* scan scratch */
    uint64_t part64[DFL_COOP_MAX_THREADS];
    uint32_t used, ntok, kind, hlit, hdist, hclen, items, header_bits;
    uint64_t extra_bits, fixed_bits;
    dfl_block_result res;
} dfl_coop;

/* the token that starts at p: the optimal parse's choice, or (choice == NULL) by length, the rule of dfl_parse_block */
DFL_HD uint32_t dfl_decide(const uint8_t *s, const uint32_t *match, uint32_t p, uint32_t end, uint32_t min_len,
                           const uint32_t *choice)
{
    if (choice) return choice[p];
    const uint32_t cur = dfl_clip(match[p], p, end, min_len);
    if (!cur) return s[p];
    const uint32_t nxt = p + 1 < end ? dfl_clip(match[p + 1], p + 1, end, min_len) : 0;
    return (nxt && DFL_TOK_LEN(nxt) > DFL_TOK_LEN(cur)) ? (uint32_t)s[p] : cur;
}

DFL_HD uint32_t dfl_token_span(uint32_t t) { return DFL_IS_MATCH(t) ? DFL_TOK_LEN(t) : 1u; }

DFL_HD void dfl_bitmap_clear(uint32_t *bm, uint32_t a, uint32_t b)         /* clear bits [a, b) */
{
    while (a < b) {
        const uint32_t wd = a >> 5, lo = a & 31u;
        const uint32_t n = (32u - lo) < (b - a) ? (32u - lo) : (b - a);
        const uint32_t mask = (n == 32u ? 0xffffffffu : ((1u << n) - 1u)) << lo;
        bm[wd] &= ~mask;
        a += n;
    }
}

/* exclusive scan of sh->part[0 .. nthreads) by thread 0; part[nthreads] = total.  Brackets itself with barriers. */
DFL_HD void dfl_team_scan(const dfl_team *t, dfl_coop *sh)
{
    DFL_SYNC(t);
    if (t->tid == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < t->nthreads; i++) { const uint32_t v = sh->part[i]; sh->part[i] = acc; acc += v; }
        sh->part[t->nthreads] = acc;
    }
    DFL_SYNC(t);
}

DFL_HD uint32_t dfl_chunk_size(uint32_t L, uint32_t nthreads)
{
    uint32_t ch = (L + nthreads - 1) / nthreads;
    ch = (ch + 31u) & ~31u;
    return ch < 32u ? 32u : ch;
}

/* cooperative parse of the block; tokens to tok[], histograms into sh->w (must be zero), returns the token count */
DFL_HD uint32_t dfl_parse_coop(const dfl_team *t, const uint8_t *s, const uint32_t *match, uint32_t begin, uint32_t end,
                               uint32_t min_len, uint32_t *tok, const uint32_t *choice, dfl_coop *sh)
{
    const uint32_t L = end - begin, CH = dfl_chunk_size(L, t->nthreads);
    const uint32_t cs = t->tid * CH < L ? t->tid * CH : L, ce = cs + CH < L ? cs + CH : L;
    /* the chunk's words of the bitmap (chunks are whole words; an empty chunk owns none) */
    const uint32_t nwords = (L + 31u) >> 5;
    const uint32_t w0 = t->tid * (CH >> 5) < nwords ? t->tid * (CH >> 5) : nwords;
    const uint32_t w1 = w0 + (CH >> 5) < nwords ? w0 + (CH >> 5) : nwords;
    for (uint32_t wd = w0; wd < w1; wd++) sh->u.bitmap[wd] = 0;
    /* 1. walk through the own chunk as if it started on a token boundary */
    uint32_t p = cs;
    while (p < ce) {
        sh->u.bitmap[p >> 5] |= 1u << (p & 31u);
        p += dfl_token_span(dfl_decide(s, match, begin + p, end, min_len, choice));
    }
    sh->exit_pos[t->tid] = p;
    DFL_SYNC(t);
    /* 2. the true orbit across the chunk borders */
    if (t->tid == 0) {
        uint32_t e = 0;
        for (uint32_t k = 0; k < t->nthreads; k++) {
            const uint32_t ks = k * CH;
            if (ks >= L) break;
            const uint32_t ke = ks + CH < L ? ks + CH : L;
            if (e >= ke) { dfl_bitmap_clear(sh->u.bitmap, ks, ke); continue; }      /* a match jumped over the chunk */
            if (e == ks) { e = sh->exit_pos[k]; continue; }                        /* the guess was right */
            dfl_bitmap_clear(sh->u.bitmap, ks, e);
            uint32_t q = e;
            for (;;) {
                if (q >= ke) { e = q; break; }                                     /* never met the owner's walk */
                if (sh->u.bitmap[q >> 5] & (1u << (q & 31u))) { e = sh->exit_pos[k]; break; }   /* merged */
                const uint32_t nq = q + dfl_token_span(dfl_decide(s, match, begin + q, end, min_len, choice));
                sh->u.bitmap[q >> 5] |= 1u << (q & 31u);
                dfl_bitmap_clear(sh->u.bitmap, q + 1, nq < ke ? nq : ke);
                q = nq;
            }
        }
    }
    DFL_SYNC(t);
    /* 3. tokens from the bitmap */
    uint32_t cnt = 0;
    for (uint32_t wd = w0; wd < w1; wd++) cnt += (uint32_t)__builtin_popcount(sh->u.bitmap[wd]);
    sh->part[t->tid] = cnt;
    dfl_team_scan(t, sh);
    uint32_t k = sh->part[t->tid];
    for (uint32_t wd = w0; wd < w1; wd++) {
        uint32_t bits = sh->u.bitmap[wd];
        while (bits) {
            const uint32_t q = (wd << 5) + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            const uint32_t tk = dfl_decide(s, match, begin + q, end, min_len, choice);
            tok[k++] = tk;
            if (DFL_IS_MATCH(tk)) {
                uint32_t sym, eb, ex;
                dfl_len_symbol(DFL_TOK_LEN(tk), &sym, &eb, &ex);
                DFL_SHARED_ADD(&sh->w.freq_ll[sym], 1u);
                dfl_dist_symbol(DFL_TOK_DIST(tk), &sym, &eb, &ex);
                DFL_SHARED_ADD(&sh->w.freq_d[sym], 1u);
            } else {
                DFL_SHARED_ADD(&sh->w.freq_ll[tk], 1u);
            }
        }
    }
    const uint32_t total = sh->part[t->nthreads];
    DFL_SYNC(t);
    return total;
}

/* dfl_build_code with the sort done by the team (rank sort); everything else on thread 0 */
DFL_HD void dfl_build_code_coop(const dfl_team *t, uint32_t *freq, uint32_t n, uint32_t limit, uint8_t *len,
                                uint16_t *code, dfl_coop *sh)
{
    dfl_work *w = &sh->w;
    if (t->tid == 0) {
        uint32_t used = 0;
        for (uint32_t i = 0; i < n; i++) { len[i] = 0; code[i] = 0; if (freq[i]) sh->sorted[used++] = (uint16_t)i; }
        for (uint32_t i = 0; used < 2 && i < n; i++)
            if (!freq[i]) { freq[i] = 1; sh->sorted[used++] = (uint16_t)i; }
        sh->used = used;
    }
    DFL_SYNC(t);
    const uint32_t used = sh->used;
    for (uint32_t i = t->tid; i < used; i += t->nthreads) {
        const uint32_t v = sh->sorted[i], fv = freq[v];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < used; j++) {
            const uint32_t u = sh->sorted[j], fu = freq[u];
            rank += (fu < fv || (fu == fv && u < v)) ? 1u : 0u;
        }
        w->order[rank] = (uint16_t)v;
    }
    DFL_SYNC(t);
    if (t->tid == 0) dfl_build_code_sorted(freq, n, used, limit, len, code, w);
    DFL_SYNC(t);
}

/* bit writer that ORs 32-bit words into zero-initialised memory, starting at any bit offset */
typedef struct { uint32_t *out; uint32_t word; uint64_t acc; uint32_t nbits; } dfl_orbits;

DFL_HD void dfl_or_init(dfl_orbits *b, uint8_t *out, uint32_t bit_offset)
{
    b->out = (uint32_t *)out; b->word = bit_offset >> 5; b->nbits = bit_offset & 31u; b->acc = 0;
}

DFL_HD void dfl_or_put(dfl_orbits *b, uint32_t value, uint32_t n)
{
    b->acc |= (uint64_t)value << b->nbits;
    b->nbits += n;
    if (b->nbits >= 32u) {
        if ((uint32_t)b->acc) DFL_OUT_OR(&b->out[b->word], (uint32_t)b->acc);
        b->word++;
        b->acc >>= 32;
        b->nbits -= 32u;
    }
}

DFL_HD void dfl_or_finish(dfl_orbits *b)
{
    if (b->nbits && (uint32_t)b->acc) DFL_OUT_OR(&b->out[b->word], (uint32_t)b->acc);
}

DFL_HD uint32_t dfl_token_bits(uint32_t tk, const dfl_work *w)
{
    if (!DFL_IS_MATCH(tk)) return w->len_ll[tk];
    uint32_t sym, eb, ex, bits;
    dfl_len_symbol(DFL_TOK_LEN(tk), &sym, &eb, &ex);
    bits = w->len_ll[sym] + eb;
    dfl_dist_symbol(DFL_TOK_DIST(tk), &sym, &eb, &ex);
    return bits + w->len_d[sym] + eb;
}

/* ---------------------------------------------------------------------------------------------------------------
 * The block.  All threads of the team call this with the same arguments; `out` must be zero-initialised, 4-byte
 * aligned and dfl_block_bound(L) long; `choice` is per-position scratch (NULL: single parse by length).  The result
 * is returned to every thread.
 * ------------------------------------------------------------------------------------------------------------- */
DFL_HD dfl_block_result dfl_encode_block_coop(const dfl_team *t, const uint8_t *s, const uint32_t *match, const uint32_t *near,
                                              const dfl_block_desc *d, const dfl_params *prm, uint32_t *tok,
                                              uint32_t *choice, uint8_t *out, dfl_coop *sh)
{
    static const uint8_t cl_order[DFL_NUM_CL] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    dfl_work *w = &sh->w;
    const uint32_t L = d->end - d->begin;
#if defined(__HIP_DEVICE_COMPILE__) && defined(DFL_PHASE_PROF)
    unsigned long long dfl_prof_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dfl_prof_last = wall_clock64();
#endif

    /* Adler-32 partial sums */
    {
        uint32_t a;
        uint64_t b;
        dfl_adler_partial(s, d->begin, d->end, t->tid, t->nthreads, &a, &b);
        sh->part[t->tid] = a;
        sh->part64[t->tid] = b;
    }
    for (uint32_t i = t->tid; i < DFL_NUM_LL; i += t->nthreads) w->freq_ll[i] = 0;
    for (uint32_t i = t->tid; i < DFL_NUM_D; i += t->nthreads) w->freq_d[i] = 0;
    for (uint32_t i = t->tid; i < DFL_NUM_CL; i += t->nthreads) w->freq_cl[i] = 0;
    DFL_SYNC(t);
    if (t->tid == 0) {
        uint32_t a = 0;
        uint64_t b = 0;
        for (uint32_t i = 0; i < t->nthreads; i++) { a += sh->part[i]; b += sh->part64[i]; }
        sh->res.adler_a = a;
        sh->res.adler_b = b;
    }
    DFL_SYNC(t);

    DFL_PROF(0);
    uint32_t ntok = dfl_parse_coop(t, s, match, d->begin, d->end, prm->min_len, tok, NULL, sh);
    DFL_PROF(1);
    if (choice && ntok < L) {
        for (int it = 0; it < DFL_DP_ITERATIONS; it++) {
            if (t->tid == 0) w->freq_ll[256] = 1;
            DFL_SYNC(t);
            dfl_build_code_coop(t, w->freq_ll, 286, 15, w->len_ll, w->code_ll, sh);
            dfl_build_code_coop(t, w->freq_d, 30, 15, w->len_d, w->code_d, sh);
            dfl_length_prices(w, t->tid, t->nthreads);
            DFL_SYNC(t);
            DFL_PROF(2);
            for (uint32_t c = t->tid; c * DFL_DP_CHUNK < L; c += t->nthreads)
                dfl_dp_chunk(s, match, near, d->begin, d->end, prm->min_len, c, w, choice, &sh->u.ring[t->tid], t->nthreads);
            for (uint32_t i = t->tid; i < DFL_NUM_LL; i += t->nthreads) w->freq_ll[i] = 0;
            for (uint32_t i = t->tid; i < DFL_NUM_D; i += t->nthreads) w->freq_d[i] = 0;
            DFL_SYNC(t);
            DFL_PROF(3);
            ntok = dfl_parse_coop(t, s, match, d->begin, d->end, prm->min_len, tok, choice, sh);
            DFL_PROF(4);
        }
    }
    DFL_PROF(5);

    if (t->tid == 0) {
        w->freq_ll[256] = 1;
        uint64_t extra_bits = 0;
        for (uint32_t i = 265; i < 285; i++) extra_bits += (uint64_t)w->freq_ll[i] * ((i - 261u) >> 2);
        for (uint32_t i = 4; i < 30; i++) extra_bits += (uint64_t)w->freq_d[i] * ((i - 2u) >> 1);
        uint64_t fixed_bits = 3 + extra_bits;
        for (uint32_t i = 0; i < DFL_NUM_LL; i++) fixed_bits += (uint64_t)w->freq_ll[i] * (i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
        for (uint32_t i = 0; i < 30; i++) fixed_bits += (uint64_t)w->freq_d[i] * 5;
        sh->extra_bits = extra_bits;
        sh->fixed_bits = fixed_bits;
    }
    DFL_SYNC(t);
    dfl_build_code_coop(t, w->freq_ll, 286, 15, w->len_ll, w->code_ll, sh);
    dfl_build_code_coop(t, w->freq_d, 30, 15, w->len_d, w->code_d, sh);
    if (t->tid == 0) {
        uint32_t hlit = 286, hdist = 30;
        while (hlit > 257 && !w->len_ll[hlit - 1]) --hlit;
        while (hdist > 1 && !w->len_d[hdist - 1]) --hdist;
        uint8_t *all = (uint8_t *)w->weight;
        for (uint32_t i = 0; i < hlit; i++) all[i] = w->len_ll[i];
        for (uint32_t i = 0; i < hdist; i++) all[hlit + i] = w->len_d[i];
        sh->items = dfl_rle_lengths(all, hlit + hdist, w);
        for (uint32_t i = 0; i < sh->items; i++) w->freq_cl[w->cl_sym[i]]++;
        sh->hlit = hlit;
        sh->hdist = hdist;
    }
    DFL_SYNC(t);
    dfl_build_code_coop(t, w->freq_cl, DFL_NUM_CL, 7, w->len_cl, w->code_cl, sh);
    if (t->tid == 0) {
        uint64_t body = 0;
        for (uint32_t i = 0; i < 286; i++) body += (uint64_t)w->freq_ll[i] * w->len_ll[i];
        for (uint32_t i = 0; i < 30; i++) body += (uint64_t)w->freq_d[i] * w->len_d[i];
        uint32_t hclen = DFL_NUM_CL;
        while (hclen > 4 && !w->len_cl[cl_order[hclen - 1]]) --hclen;
        uint64_t header = 3 + 14 + 3ull * hclen;
        for (uint32_t i = 0; i < sh->items; i++) {
            const uint32_t sy = w->cl_sym[i];
            header += w->len_cl[sy] + (sy == 16 ? 2u : (sy == 17 ? 3u : (sy == 18 ? 7u : 0u)));
        }
        const uint64_t dyn_bits = header + sh->extra_bits + body;
        const uint32_t chunks = L ? (L + 65534u) / 65535u : 1u;
        const uint64_t stored_bits = 8ull * ((uint64_t)L + 5ull * chunks), sync_bits = d->last ? 0u : 8ull * 5;
        sh->hclen = hclen;
        if (stored_bits <= sh->fixed_bits + sync_bits && stored_bits <= dyn_bits + sync_bits) sh->kind = 0;
        else if (sh->fixed_bits <= dyn_bits) sh->kind = 1;
        else sh->kind = 2;
        sh->header_bits = sh->kind == 2 ? (uint32_t)header : 3u;
        sh->res.kind = sh->kind;
        sh->res.tokens = ntok;
        if (sh->kind == 1) {
            dfl_fixed_lengths(w->len_ll, w->len_d);
            dfl_canonical(w->len_ll, DFL_NUM_LL, w->code_ll);
            dfl_canonical(w->len_d, DFL_NUM_D, w->code_d);
        }
    }
    DFL_SYNC(t);

    if (sh->kind == 0) {
        const uint32_t chunks = L ? (L + 65534u) / 65535u : 1u;
        for (uint32_t c = t->tid; c < chunks; c += t->nthreads) {
            const uint32_t off = c * 65535u, len = L - off > 65535u ? 65535u : L - off;
            uint8_t *h = out + (size_t)c * 65540u;
            h[0] = (d->last && c + 1 == chunks) ? 1 : 0; h[1] = (uint8_t)len; h[2] = (uint8_t)(len >> 8); h[3] = (uint8_t)~len; h[4] = (uint8_t)(~len >> 8);
        }
        for (uint32_t i = t->tid; i < L; i += t->nthreads) out[(size_t)(i / 65535u) * 65540u + 5u + i % 65535u] = s[d->begin + i];
        if (t->tid == 0) sh->res.bytes = L + 5u * chunks;
        DFL_SYNC(t);
        return sh->res;
    }

    DFL_PROF(6);
    /* header by thread 0, tokens by everyone */
    if (t->tid == 0) {
        dfl_orbits bw;
        dfl_or_init(&bw, out, 0);
        if (sh->kind == 1) {
            dfl_or_put(&bw, d->last | (1u << 1), 3);
        } else {
            dfl_or_put(&bw, d->last | (2u << 1), 3);
            dfl_or_put(&bw, sh->hlit - 257u, 5);
            dfl_or_put(&bw, sh->hdist - 1u, 5);
            dfl_or_put(&bw, sh->hclen - 4u, 4);
            for (uint32_t i = 0; i < sh->hclen; i++) dfl_or_put(&bw, w->len_cl[cl_order[i]], 3);
            for (uint32_t i = 0; i < sh->items; i++) {
                const uint32_t sy = w->cl_sym[i];
                dfl_or_put(&bw, w->code_cl[sy], w->len_cl[sy]);
                if (sy == 16) dfl_or_put(&bw, w->cl_arg[i], 2);
                else if (sy == 17) dfl_or_put(&bw, w->cl_arg[i], 3);
                else if (sy == 18) dfl_or_put(&bw, w->cl_arg[i], 7);
            }
        }
        dfl_or_finish(&bw);
    }
    const uint32_t per = (ntok + t->nthreads - 1) / t->nthreads;
    const uint32_t k0 = t->tid * per < ntok ? t->tid * per : ntok, k1 = k0 + per < ntok ? k0 + per : ntok;
    uint32_t bits = 0;
    for (uint32_t k = k0; k < k1; k++) bits += dfl_token_bits(tok[k], w);
    sh->part[t->tid] = bits;
    dfl_team_scan(t, sh);
    {
        dfl_orbits bw;
        dfl_or_init(&bw, out, sh->header_bits + sh->part[t->tid]);
        for (uint32_t k = k0; k < k1; k++) {
            const uint32_t tk = tok[k];
            if (DFL_IS_MATCH(tk)) {
                uint32_t sym, eb, ex;
                dfl_len_symbol(DFL_TOK_LEN(tk), &sym, &eb, &ex);
                dfl_or_put(&bw, (uint32_t)w->code_ll[sym] | (ex << w->len_ll[sym]), w->len_ll[sym] + eb);
                dfl_dist_symbol(DFL_TOK_DIST(tk), &sym, &eb, &ex);
                dfl_or_put(&bw, (uint32_t)w->code_d[sym] | (ex << w->len_d[sym]), w->len_d[sym] + eb);
            } else {
                dfl_or_put(&bw, w->code_ll[tk], w->len_ll[tk]);
            }
        }
        dfl_or_finish(&bw);
    }
    DFL_PROF(7);
    if (t->tid == 0) {
        dfl_orbits bw;
        uint32_t at = sh->header_bits + sh->part[t->nthreads];
        dfl_or_init(&bw, out, at);
        dfl_or_put(&bw, w->code_ll[256], w->len_ll[256]);
        at += w->len_ll[256];
        if (d->last) {                                             /* the stream ends here */
            dfl_or_finish(&bw);
            sh->res.bytes = (at + 7u) / 8u;
        } else {
            dfl_or_put(&bw, 0, 3);                                 /* empty stored block: the sync marker */
            at += 3u;
            const uint32_t pad = (8u - (at & 7u)) & 7u;
            dfl_or_put(&bw, 0, pad);
            dfl_or_put(&bw, 0x0000u, 16);
            dfl_or_put(&bw, 0xffffu, 16);
            dfl_or_finish(&bw);
            sh->res.bytes = (at + pad) / 8u + 4u;
        }
#if defined(__HIP_DEVICE_COMPILE__) && defined(DFL_PHASE_PROF)
        DFL_PROF(7);
        if (d->begin == 0) printf("phases (100 MHz ticks): init+adler %llu parse0 %llu | codes %llu dp %llu parse %llu | - %llu plan %llu emit %llu\n",
                                  dfl_prof_acc[0], dfl_prof_acc[1], dfl_prof_acc[2], dfl_prof_acc[3], dfl_prof_acc[4], dfl_prof_acc[5], dfl_prof_acc[6], dfl_prof_acc[7]);
#endif
    }
    DFL_SYNC(t);
    return sh->res;
}

#endif
