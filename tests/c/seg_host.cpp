/*
 * seg_host.cpp -- TEST INFRASTRUCTURE: runs the kernel bodies of the segment-parallel row engine
 * (pngloss_amd/csrc/pl_seg_core.h, the same source hipcc compiles into pl_seg.hip's kernels) on the CPU, as plain loops
 * over (workgroup, thread), so that the CPU suite can check the engine's logic bit-exactly against the oracle without a GPU.
 * Never shipped, never loaded by the product (which has no CPU path).
 *
 *   seg_host_optimize(rgba, W, H, row_filters|NULL, strength, bleed, stats[8])  -> 0, or 64 if (strength, bleed) has more
 *   chain states than the enumeration has lanes (the product then uses the one-workgroup-per-image engine)
 */
static unsigned long long seg_dbg[4][2];   /* [slot]: events, sum */
#define SEG_DEBUG_COUNT(slot, v) (seg_dbg[slot][0]++, seg_dbg[slot][1] += (v))
/* per row: which candidates went through an epoch (failed validation) or an extra start, and who won */
static unsigned seg_row_mask, seg_row_none;
static unsigned long long seg_epochs[5], seg_epoch_rows[5], seg_epoch_rows_won[5], seg_none_starts, seg_none_started_won, seg_rows;
static void seg_debug_row(int kind, unsigned failed, int winner, int start_none)
{
    if (kind == 1) { for (int f = 0; f < 5; f++) if ((failed >> f) & 1u) { seg_epochs[f]++; seg_row_mask |= 1u << f; } if (start_none) { seg_none_starts++; seg_row_none = 1; } }
    if (kind == 3) {
        seg_rows++;
        for (int f = 0; f < 5; f++) if ((seg_row_mask >> f) & 1u) { seg_epoch_rows[f]++; if (winner == f) seg_epoch_rows_won[f]++; }
        if (seg_row_none && winner == 0) seg_none_started_won++;
        seg_row_mask = 0; seg_row_none = 0;
    }
}
#define SEG_DEBUG_ROW(kind, failed, winner, start_none) seg_debug_row((kind), (failed), (winner), (start_none))
#include <cstdio>
#include <cstdlib>
#define SEG_DEBUG_BREAK(f, c, sg, est, y) do { if (getenv("SEG_HOST_VERBOSE") && atoi(getenv("SEG_HOST_VERBOSE")) > 1) fprintf(stderr, "seg_host: row %u broken off: candidate %d channel %d segment %u, entry state left %u cn %d th %d\n", (unsigned)(y), (int)(f), (int)(c), (unsigned)(sg), (unsigned)((est) & 255u), (int)(((est) >> 8) & 0xffffu) - 32768, (int)((est) >> 24) - 128); } while (0)
#include "../../pngloss_amd/csrc/pl_seg_core.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

int paeth_i(int a, int d, int l) { return seg_paeth(a, d, l); }

struct Arena {
    std::vector<std::vector<unsigned char>> bufs;
    template <class T> T *take(size_t n) { bufs.emplace_back((n ? n : 1) * sizeof(T) + 64, (unsigned char)0xA5); return (T *)bufs.back().data(); }   /* junk-filled: nothing may rely on zeroed memory */
};

} // namespace

/* what the engine does with a (strength, bleed) pair: out = { supported, seeded, chain states (exhaustive), run-in pixels, cmax, tmax, dmax, seeds of none / up } */
extern "C" int seg_host_describe(unsigned strength, long bleed, int32_t *out)
{
    static SegParams P;
    const bool ok = seg_build_params(P, (int)strength, (int)bleed);
    out[0] = ok; out[1] = P.seeded; out[2] = P.ns; out[3] = P.kin; out[4] = P.cmax; out[5] = P.tmax; out[6] = P.dmax; out[7] = P.nseed_small;
    return ok ? 0 : 64;
}

/* the seed set of the unit enumeration from seeds (round 6): out = { seed_n, seed_kin, ns, dmax }, states[i] = (delta, cn, th) of seed i */
extern "C" int seg_host_seeds(unsigned strength, long bleed, int32_t *out, int32_t *states)
{
    static SegParams P;
    if (!seg_build_params(P, (int)strength, (int)bleed)) return 64;
    out[0] = P.seed_n; out[1] = P.seed_kin; out[2] = P.ns; out[3] = P.dmax;
    for (int i = 0; i < P.seed_n; i++) {
        if (P.seed_idx[i] >= P.ns) return 65;
        const uint32_t w = P.st_pack[P.seed_idx[i]];
        states[3 * i] = (int)(w & 255u) - 128; states[3 * i + 1] = (int)((w >> 8) & 255u) - 128; states[3 * i + 2] = (int)((w >> 16) & 255u) - 128;
    }
    return 0;
}

extern "C" int seg_host_optimize(unsigned char *rgba, uint32_t W, uint32_t H, unsigned char *row_filters, unsigned strength, long bleed, uint32_t *stats)
{
    if (!W || !H) return 0;
    static SegParams P;
    if (!seg_build_params(P, (int)strength, (int)bleed, getenv("SEG_HOST_SEEDED") != nullptr)) return 64;
    if (getenv("SEG_HOST_FORCE_FILTER")) P.engine_flags = (atoi(getenv("SEG_HOST_FORCE_FILTER")) + 1) << 8;
    if (getenv("SEG_HOST_FLAGS")) P.engine_flags |= atoi(getenv("SEG_HOST_FLAGS")) & 0xfe;   /* test hooks of the chain kernel (2: slow path, 4: wide stride) */
    if (getenv("SEG_HOST_UNIT") && atoi(getenv("SEG_HOST_UNIT")) && !P.seeded && (P.ns <= SEG_NSP || atoi(getenv("SEG_HOST_UNIT")) > 1)) { P.unit = SEG_UNIT; P.tparts = SEG_TPARTS_BATCH; }
    if (getenv("SEG_HOST_TPARTS")) P.tparts = atoi(getenv("SEG_HOST_TPARTS")) == 1 ? SEG_TPARTS_BATCH : SEG_TPARTS;   /* enumeration in units, as the launcher asks for batches (seg_enum_unit_body) */
    /* classify + pack into slots (what pl_classify / pl_repack do on the device) */
    bool gray = true, opaque = true;
    for (size_t i = 0; i < (size_t)W * H; i++) { const unsigned char *p = rgba + 4 * i; gray &= p[0] == p[1] && p[1] == p[2]; opaque &= p[3] == 255; }
    const uint32_t bpp = gray ? (opaque ? 1u : 2u) : (opaque ? 3u : 4u);
    Arena A;
    uint32_t *img = A.take<uint32_t>((size_t)W * H);
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const unsigned char *p = rgba + 4 * i;
        img[i] = bpp == 4 ? ((uint32_t)p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)) : bpp == 3 ? ((uint32_t)p[0] | (p[1] << 8) | (p[2] << 16))
                 : bpp == 2 ? ((uint32_t)p[1] | (p[3] << 8)) : (uint32_t)p[1];
    }
    /* original_frequency + ranks (pl_hist / pl_rank) */
    std::vector<uint32_t> oh(5 * 256, 0), rank(5 * 256, 0);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++)
            for (uint32_t c = 0; c < bpp; c++) {
                auto at = [&](long yy, long xx) -> int { return (yy < 0 || xx < 0) ? 0 : (int)((img[(size_t)yy * W + xx] >> (8 * c)) & 255u); };
                const int hv = at(y, x), lv = at(y, (long)x - 1), av = at((long)y - 1, x), dv = at((long)y - 1, (long)x - 1);
                oh[0 * 256 + (hv & 255)]++; oh[1 * 256 + ((hv - lv) & 255)]++; oh[2 * 256 + ((hv - av) & 255)]++;
                oh[3 * 256 + ((hv - ((av + lv) >> 1)) & 255)]++; oh[4 * 256 + ((hv - paeth_i(av, dv, lv)) & 255)]++;
            }
    for (int f = 0; f < 5; f++) for (int b = 0; b < 256; b++) { uint32_t r = 0; for (int k = 0; k < 256; k++) r += oh[f * 256 + k] < oh[f * 256 + b]; rank[f * 256 + b] = r; }

    SegJob j{};
    j.img = img; j.W = W; j.H = H; j.bpp = bpp;
    j.row_filters = row_filters;
    j.row_ids = A.take<uint8_t>(H);
    j.orig_rank = rank.data();
    j.cand = A.take<uint32_t>((size_t)5 * W * 4);
    j.err0 = A.take<uint32_t>((size_t)W * 4); j.err1 = A.take<uint32_t>((size_t)W * 4);   /* (by row parity; the first control kernel zeroes what row 0 reads) */
    j.rowcopy = A.take<uint32_t>((size_t)W * 3);
    j.final_hist = A.take<uint32_t>(256); j.result = A.take<int32_t>(64); j.progress = nullptr;
    j.nseg = (W + SEG_L - 1) / SEG_L; j.ngrp = (j.nseg + SEG_GRP - 1) / SEG_GRP;
    if (W > SEG_MAX_WIDTH) return 64;
    j.ctl = A.take<SegCtl>(3); j.base = A.take<uint32_t>(3 * 5 * 256); j.H0 = A.take<uint32_t>(3 * 256); j.acc = A.take<SegAcc>(3);
    j.tables = A.take<uint32_t>(5 * SEG_TBL_WORDS);
    j.maps = A.take<uint16_t>((size_t)5 * j.nseg * 4 * P.nsp);
    j.ehash = A.take<uint32_t>(P.seeded ? (size_t)5 * j.nseg * 4 * SEG_EH_WORDS : 4);
    j.rout = A.take<uint16_t>((size_t)5 * j.nseg * 4 * SEG_NSP);
    j.rst = A.take<uint32_t>((size_t)5 * j.nseg * 4 * SEG_NSP);
    j.rck = A.take<uint32_t>((size_t)5 * j.nseg * 4 * SEG_NSP * (SEG_PARTS - 1));
    j.dnout = A.take<uint16_t>((size_t)5 * j.nseg * 4);
    j.dcnt = A.take<uint32_t>((size_t)5 * j.nseg * 4);
    j.entry = A.take<uint32_t>((size_t)5 * j.nseg * 4);
    j.segcnt = A.take<uint16_t>((size_t)5 * j.nseg * 256);
    j.grpcnt = A.take<uint32_t>((size_t)5 * j.ngrp * 256);
    j.grpleft = A.take<uint32_t>((size_t)5 * j.ngrp);
    j.firstidx = A.take<uint32_t>(5 * 4 * 2);
    j.rowmm = A.take<int32_t>(4);
    std::vector<unsigned char> smem(160 * 1024, 0x5A);
    const int ncommit = (int)((W + SEG_COMMIT_W - 1) / SEG_COMMIT_W);
    j.nbreak = 0u;
    j.self = &j; for (int k = 0; k < 3; k++) { j.v[k].magic = 0u; j.v[k].finished = 0u; j.v[k].ignore = 0u; j.vfail[k] = 0u; }
    j.ctl[2].magic = 0u; j.acc[2].failmask = 0u;   /* (seg_k_resolve does this on the device: the first attempt finds no attempt behind it) */
    int attempt = 0;
    const long max_attempts = (long)H * ((long)strength + 1) * (2 + 2 * SEG_MAX_RESTARTS * SEG_NFILT) + 1024;   /* (the product's bound: pl_host.hip) */
    /* One attempt = four launches: [control of this attempt + validation of the attempt before], enumerate, chain, replay.  The two halves of
     * the first launch run side by side on the device; here one after the other, in either order (SEG_HOST_VAL_FIRST): neither may depend on it. */
    const bool val_first = getenv("SEG_HOST_VAL_FIRST") != nullptr;
    for (;; attempt++) {
        if (attempt > max_attempts) { fprintf(stderr, "seg_host: no progress\n"); return 65; }
        const int par = attempt % 3, kv = (par + 2) % 3;
        std::vector<unsigned char> cvsm((size_t)(P.tparts == SEG_TPARTS_BATCH ? SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS_BATCH)) : SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS))), 0x5A);       /* (the launch's LDS request: the sanitizer build sees an overrun) */
        for (int half = 0; half < 2; half++) {
            if ((half == 0) != val_first) { for (int bx = 0; bx < SEG_CTL_IMG_OF(P) + 1 + ncommit; bx++) { if (P.tparts == SEG_TPARTS_BATCH) seg_ctl_body<SEG_TPARTS_BATCH>(j, P, par, bx, cvsm.data()); else seg_ctl_body<SEG_TPARTS>(j, P, par, bx, cvsm.data()); } }
            else if (P.tparts == SEG_TPARTS_BATCH) { for (int f = 0; f < SEG_NFILT; f++) for (uint32_t vg = 0; vg * SEG_VGRP_OF(SEG_TPARTS_BATCH) < j.nseg; vg++) seg_post_body<SEG_VGRP_OF(SEG_TPARTS_BATCH)>(j, P, seg_ctl_view(j, kv, f), kv, f, (int)vg, cvsm.data()); }
            else { for (int f = 0; f < SEG_NFILT; f++) for (uint32_t vg = 0; vg * SEG_VGRP_OF(SEG_TPARTS) < j.nseg; vg++) seg_post_body<SEG_VGRP_OF(SEG_TPARTS)>(j, P, seg_ctl_view(j, kv, f), kv, f, (int)vg, cvsm.data()); }
        }
        if (j.ctl[par].finished == 2u) break;
        /* the enumeration's workgroups come in two sizes; the product picks by row width, SEG_HOST_ENUM_NT pins one */
        int nt = j.nseg <= SEG_ENUM_NT_SMALL_MAX_NSEG ? 512 : 1024;
        if (getenv("SEG_HOST_ENUM_NT")) nt = atoi(getenv("SEG_HOST_ENUM_NT")) == 1024 ? 1024 : 512;
        /* the enumeration kernel is launched with exactly SEG_SM_ENUM_NT(nt) bytes of LDS: the bodies get a buffer of that size here, and the
         * sanitizer build (tests/test_seg_host.py) sees any byte they touch beyond it */
        std::vector<unsigned char> esm((size_t)SEG_SM_ENUM_NT(nt), 0x5A);
        const bool seeds1 = P.unit == 1 && !P.seeded && P.ns <= SEG_NSP && P.seed_n > 0 && getenv("SEG_HOST_SEEDS") != nullptr && atoi(getenv("SEG_HOST_SEEDS")) != 0;
        if (seeds1) {
            /* pl_seg.hip:seg_k_enum_unit<1> (round 6: small and mid-size batches): the unit enumeration's bodies segment by segment, started from seeds */
            if (getenv("SEG_HOST_SEED_KIN")) P.seed_kin = atoi(getenv("SEG_HOST_SEED_KIN"));
            std::vector<unsigned char> usm((size_t)SEG_SM_ENUM_UNIT, 0x5A);
            const uint32_t npairs = j.nseg * j.bpp;
            for (int f = 0; f < SEG_NFILT; f++) {
                const SegCtlView cv = seg_ctl_view(j, par, f);
                if (seg_is_small(P, f)) { for (uint32_t g = 0; g * SEG_UNC_SMALL_OF(1) < npairs; g++) seg_enum_unit_body<SEG_NSS, 1, SEG_UNC_SMALL_OF(1)>(j, P, cv, par, f, (int)g, usm.data()); }
                else if (seg_unit_from_seeds(j, P, cv, f, 1)) { for (uint32_t g = 0; g * SEG_UNC_SEEDS1 < npairs; g++) seg_enum_unit_body<SEG_SEED_LANES, 1, SEG_UNC_SEEDS1, true>(j, P, cv, par, f, (int)g, usm.data()); }
                else { for (uint32_t g = 0; g * SEG_UNC < npairs; g++) seg_enum_unit_body<SEG_NSP, 1, SEG_UNC>(j, P, cv, par, f, (int)g, usm.data()); }
            }
            for (int f = 0; f < SEG_NFILT; f++) seg_first_body<SEG_UNT, false>(j, P, seg_ctl_view(j, par, f), par, f, usm.data());
        } else
        if (P.unit > 1) {
            /* pl_seg.hip:seg_k_enum_unit: per candidate `per` workgroups of SEG_UNC (unit, channel) pairs, then the five walkers -- with the kernel's LDS size */
            std::vector<unsigned char> usm((size_t)SEG_SM_ENUM_UNIT, 0x5A);
            const bool seeds = getenv("SEG_HOST_SEEDS") != nullptr && atoi(getenv("SEG_HOST_SEEDS")) && P.seed_n > 0;     /* the first phase from seeds with a run-in (round 6) */
            if (seeds && getenv("SEG_HOST_SEED_KIN")) P.seed_kin = atoi(getenv("SEG_HOST_SEED_KIN"));
            const uint32_t perseed = (((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * 4 + SEG_UNC_SEEDS - 1) / SEG_UNC_SEEDS;
            const uint32_t perb = (((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * 4 + SEG_UNC - 1) / SEG_UNC, pers = (((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * 4 + SEG_UNC_SMALL - 1) / SEG_UNC_SMALL;
            for (int f = 0; f < SEG_NFILT; f++) {
                if (seg_is_small(P, f)) { for (uint32_t g = 0; g < pers; g++) if (g * SEG_UNC_SMALL < ((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * j.bpp) seg_enum_unit_body<SEG_NSS, SEG_UNIT, SEG_UNC_SMALL>(j, P, seg_ctl_view(j, par, f), par, f, (int)g, usm.data()); }
                else if (seeds && seg_unit_from_seeds(j, P, seg_ctl_view(j, par, f), f, 1)) { for (uint32_t g = 0; g < perseed; g++) if (g * SEG_UNC_SEEDS < ((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * j.bpp) seg_enum_unit_body<SEG_SEED_LANES, SEG_UNIT, SEG_UNC_SEEDS, true>(j, P, seg_ctl_view(j, par, f), par, f, (int)g, usm.data()); }
                else { for (uint32_t g = 0; g < perb; g++) if (g * SEG_UNC < ((j.nseg + SEG_UNIT - 1) / SEG_UNIT) * j.bpp) seg_enum_unit_body<SEG_NSP, SEG_UNIT, SEG_UNC>(j, P, seg_ctl_view(j, par, f), par, f, (int)g, usm.data()); }
            }
            for (int f = 0; f < SEG_NFILT; f++) seg_first_body<SEG_UNT, true>(j, P, seg_ctl_view(j, par, f), par, f, usm.data());
        } else
        if (P.seeded) {
            std::vector<unsigned char> ssm((size_t)SEG_SM_ENUM_SEEDED(nt), 0x5A);
            for (int f = 0; f < SEG_NFILT; f++)
                for (uint32_t sg = 0; sg < j.nseg; sg++) {
                    if (nt == 512) for (int ch = 0; ch < 2; ch++) seg_enum_seeded_body<512>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, ch, ssm.data());
                    else seg_enum_seeded_body<1024>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, 0, ssm.data());
                }
        } else
        for (int f = 0; f < SEG_NFILT; f++) {
            if (nt == 512) {
                if (seg_is_small(P, f)) for (uint32_t sg = 0; sg < j.nseg; sg += 4) seg_enum_small_body<512>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, esm.data());
                else for (uint32_t sg = 0; sg < j.nseg; sg++) for (int ch = 0; ch < 2; ch++) seg_enum_body<512>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, ch, esm.data());
            } else {
                if (seg_is_small(P, f)) for (uint32_t sg = 0; sg < j.nseg; sg += 8) seg_enum_small_body<1024>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, esm.data());
                else for (uint32_t sg = 0; sg < j.nseg; sg++) seg_enum_body<1024>(j, P, seg_ctl_view(j, par, f), par, f, (int)sg, 0, esm.data());
            }
        }
        if (P.unit <= 1 && !seeds1) for (int f = 0; f < SEG_NFILT; f++) { if (nt == 512) seg_first_body<512, false>(j, P, seg_ctl_view(j, par, f), par, f, esm.data()); else seg_first_body<1024, false>(j, P, seg_ctl_view(j, par, f), par, f, esm.data()); }
        if (P.seeded && j.nseg > 1) {   /* the gather kernel of seeded sets: every (filter, channel, block of segments) */
            const unsigned nblk = (j.nseg - 1 + SEG_GS - 1) / SEG_GS;
            for (int f = 0; f < SEG_NFILT; f++) for (int c = 0; c < 4; c++) for (unsigned b = 0; b < nblk; b++) seg_gather_seeded_body(j, seg_ctl_view(j, par, f), f, c, (int)b);
        }
        {   /* the chain kernel's LDS is sized by the row's segments: the same size here (the sanitizer build sees an overrun) */
            std::vector<unsigned char> csm(P.seeded ? (size_t)SEG_SM_CHAIN(j.nseg) : (size_t)SEG_SM_CHAIN_X(P.unit > 1 ? (j.nseg + P.unit - 1) / P.unit : j.nseg), 0x5A);
            for (int f = 0; f < SEG_NFILT; f++) for (int c = 0; c < 4; c++) {
                if (P.seeded) seg_chain_body<true, SEG_CHAIN_THREADS, false>(j, P, seg_ctl_view(j, par, f), par, f, c, csm.data());
                else if (P.unit > 1) seg_chain_body<false, SEG_CHAIN_THREADS_UNIT, true>(j, P, seg_ctl_view(j, par, f), par, f, c, csm.data());
                else seg_chain_body<false, SEG_CHAIN_THREADS, false>(j, P, seg_ctl_view(j, par, f), par, f, c, csm.data());
            }
            if (P.unit > 1) seg_extremes_body<SEG_CHAIN_THREADS_UNIT>(j, P, seg_ctl_view(j, par, 0), par, csm.data()); else seg_extremes_body<SEG_CHAIN_THREADS>(j, P, seg_ctl_view(j, par, 0), par, csm.data());
        }
        { std::vector<unsigned char> rsm((size_t)SEG_SM_REPLAY, 0x5A); for (int f = 0; f < SEG_NFILT; f++) for (uint32_t g = 0; g < j.ngrp; g++) { if (P.unit > 1) seg_replay_body<SEG_REPLAY_NT_BATCH>(j, P, seg_ctl_view(j, par, f), par, f, (int)g, rsm.data()); else seg_replay_body<SEG_REPLAY_NT>(j, P, seg_ctl_view(j, par, f), par, f, (int)g, rsm.data()); } }
    }
    const SegCtl &fc = j.ctl[attempt % 3];
    if (getenv("SEG_HOST_VERBOSE")) {
        fprintf(stderr, "seg_host: %llu rows; epochs per candidate (none sub up avg paeth): %llu %llu %llu %llu %llu; rows with an epoch of it: %llu %llu %llu %llu %llu, of which it won: %llu %llu %llu %llu %llu; extra starts of none %llu (won %llu)\n",
                seg_rows, seg_epochs[0], seg_epochs[1], seg_epochs[2], seg_epochs[3], seg_epochs[4], seg_epoch_rows[0], seg_epoch_rows[1], seg_epoch_rows[2], seg_epoch_rows[3], seg_epoch_rows[4],
                seg_epoch_rows_won[0], seg_epoch_rows_won[1], seg_epoch_rows_won[2], seg_epoch_rows_won[3], seg_epoch_rows_won[4], seg_none_starts, seg_none_started_won);
    }
    if (getenv("SEG_HOST_VERBOSE")) fprintf(stderr, "seg_host: replay lanes from an entry state %llu (%.1f px each), from a checkpoint %llu (%.1f px each)\n", seg_dbg[0][0], seg_dbg[0][0] ? (double)seg_dbg[0][1] / seg_dbg[0][0] : 0.0, seg_dbg[1][0], seg_dbg[1][0] ? (double)seg_dbg[1][1] / seg_dbg[1][0] : 0.0);
    if (getenv("SEG_HOST_VERBOSE")) fprintf(stderr, "seg_host: chain repairs (segments walked step by step) %llu\n", seg_dbg[2][0]);
    if (stats) { stats[0] = (uint32_t)attempt; stats[1] = fc.restarts_total; stats[2] = fc.retried; stats[3] = fc.serial_rows; stats[4] = (uint32_t)j.result[2]; stats[5] = bpp; stats[6] = (uint32_t)P.ns; stats[7] = fc.status; }
    /* unpack (pl_unpack) */
    for (size_t i = 0; i < (size_t)W * H; i++) {
        unsigned char *p = rgba + 4 * i; const uint32_t w = img[i];
        switch (bpp) {
        case 1: p[0] = p[1] = p[2] = (unsigned char)w; p[3] = 255; break;
        case 2: p[0] = p[1] = p[2] = (unsigned char)w; p[3] = (unsigned char)(w >> 8); break;
        case 3: p[0] = (unsigned char)w; p[1] = (unsigned char)(w >> 8); p[2] = (unsigned char)(w >> 16); p[3] = 255; break;
        default: p[0] = (unsigned char)w; p[1] = (unsigned char)(w >> 8); p[2] = (unsigned char)(w >> 16); p[3] = (unsigned char)(w >> 24); break;
        }
    }
    return (int)fc.status;
}
