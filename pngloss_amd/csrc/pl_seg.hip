/*
 * pl_seg.hip -- kernels and launcher of the SEGMENT-PARALLEL row engine: one image spread over the whole MI355X.
 *
 * The algorithm, its proof obligation (the validation pass) and the kernel bodies live in pl_seg_core.h, which is also compiled
 * for the CPU by tests/c/seg_host.cpp.  Here: the four gfx950 kernels of one row attempt, blockIdx.y = image of the batch,
 *
 *   seg_k_ctl     5 x 4 candidate workgroups (each a quarter of a candidate's decision tables) + 1 image-wide + W/256 commit workgroups
 *   seg_k_enum    3 x nseg x 2 workgroups of 512 lanes (a channel pair x 256 chain states; 1024 lanes = 4 channels for large batches) for the
 *                 filters that look at the left pixel, 2 x nseg/8 for none / up, 5 first-segment walkers; tables + pixel records in LDS
 *   seg_k_chain   5 x 4 workgroups, a row's dense transition tables (linked: an entry is the index of the next table's entry) and exit states in LDS (up to 149 KB of the CU's 160 KB)
 *   (seeded state sets only: seg_k_gather_seeded, 5 x 4 x nseg/4 workgroups of 256 lanes between the enumeration and the chain -- the chain's lookups by value)
 *   seg_k_replay  5 x ngrp workgroups: lane = (segment, quarter, channel), 8 steps each from the enumeration's checkpoints
 *   (the exact validation of every decision -- seg_post_body, 5 x 2 ngrp workgroups of 1024 lanes -- rides in seg_k_ctl's launch, one attempt behind)
 *
 * and no grid barrier anywhere: consecutive kernels on one stream are the grid-wide synchronisation (1.5-2 us on this machine
 * against 4-7 us for a hand-made in-kernel barrier across 8 XCDs, /opt/skills/guides/MI355X_MICROARCH.md), and every piece of
 * state is in device memory, so the host only enqueues attempts until the images report that they are finished.
 */
#include "pl_seg.h"

#include <atomic>

namespace {

/* a workgroup's view of attempt k (SegCtlView) straight from the record in device memory: every address follows from the kernel's arguments, so
 * these scalar loads travel with the loads of the record itself -- no second round trip before the workgroup knows whether it has work */
__device__ __forceinline__ SegCtlView seg_view_of(const SegJob *rec, int k, int f)
{
    typedef const __attribute__((address_space(4))) SegJob *seg_const_job;
    seg_const_job c = (seg_const_job)(uintptr_t)rec;
    SegCtlView v;
    const uint32_t fin = c->v[k].finished, magic = c->v[k].magic, ign = c->v[k].ignore, fm = c->vfail[seg_k_prev(k)];
    v.y = c->v[k].y; v.s = c->v[k].s; v.active = c->v[k].active[f]; v.start_x = c->v[k].start_x[f];
    v.finished = (fin != 0u || magic != SEG_MAGIC || (fm & ~ign) != 0u) ? 1u : 0u;
    return v;
}

__global__ void seg_k_resolve(const PlJob *jobs, SegJob *sj, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        sj[i].bpp = pl_job_bpp(jobs[sj[i].job_index]);
        sj[i].ctl[2].magic = 0u;          /* the image's first attempt (copy 0) finds no attempt behind it: no control block, */
        sj[i].acc[2].failmask = 0u;       /* ... no failed validation */
        for (int k = 0; k < 3; k++) { sj[i].v[k].magic = 0u; sj[i].v[k].finished = 0u; sj[i].v[k].ignore = 0u; sj[i].vfail[k] = 0u; }   /* (the same in the record's own copies) */
        sj[i].nbreak = 0u;
    }
}

/* First launch of attempt k: its CONTROL workgroups (blockIdx.x < nctl: decide the attempt before optimistically, commit, prepare this one) and, side by
 * side with them, the VALIDATION workgroups of the attempt before (copy kv): the proof of what is being decided arrives one launch later and
 * takes nothing off the critical path (seg_ctl_body says what happens when it fails). */
#ifndef SEG_EXPERIMENT_NO_VAL_CODE
#define SEG_EXPERIMENT_NO_VAL_CODE 0      /* (1: TIMING EXPERIMENT -- the control kernel without the validation's code in it; results unvalidated) */
#endif
#if SEG_EXPERIMENT_NO_VAL_CODE
#define SEG_CTL_BOUNDS __launch_bounds__(SEG_THREADS)
#else
#define SEG_CTL_BOUNDS __launch_bounds__(SEG_THREADS, 8)
#endif
template <int TPARTS>
__global__ SEG_CTL_BOUNDS void seg_k_ctl(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int k, unsigned nctl, unsigned max_ngrp)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    if (blockIdx.x < nctl) {
        constexpr unsigned ctl_img = SEG_NFILT * TPARTS;
        if (blockIdx.x > ctl_img && (blockIdx.x - ctl_img - 1) * SEG_COMMIT_W >= j.W) return;
        seg_ctl_body<TPARTS>(j, *P, k, (int)blockIdx.x, seg_smem);
        return;
    }
#if SEG_EXPERIMENT_NO_VAL_CODE
    return;
#endif
    /* validation groups are half replay groups (one image) or whole ones (batches in units): max_ngrp * (SEG_GRP / VGRP) workgroups per candidate */
    constexpr unsigned VGRP = SEG_VGRP_OF(TPARTS);
    const unsigned bx = blockIdx.x - nctl, per = max_ngrp * (SEG_GRP / VGRP), f = bx / per, vg = bx % per;
    if (vg * VGRP >= j.nseg) return;
    seg_post_body<(int)VGRP>(j, *P, seg_view_of(sj + blockIdx.y, seg_k_prev(k), (int)f), seg_k_prev(k), (int)f, (int)vg, seg_smem);
}

/* NT threads per workgroup: 1024 (four channels of a segment) or 512 (a channel pair), see SEG_ENUM_NT_SMALL_MAX_NSEG */
template <int NT>
__global__ __launch_bounds__(NT) void seg_k_enum(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par, unsigned max_nseg)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    /* workgroups [0, nbig * max_nseg * halves): one segment (and channel group) of a filter that looks at the left pixel; behind them:
     * NT / 128 segments of none / up each; the last five walk the epoch's first segment of one candidate each */
    const bool small_ok = P->small_ok != 0;
    const unsigned nbig = small_ok ? 3u : 5u;
    constexpr unsigned halves = 4 / (NT / SEG_NSP), small_segs = NT / (4 * SEG_NSS);
    if (blockIdx.x < nbig * max_nseg * halves) {
        const unsigned k = blockIdx.x / (max_nseg * halves), r = blockIdx.x % (max_nseg * halves), seg = r / halves, chalf = r % halves;
        const unsigned f = small_ok ? (k == 0 ? 1u : (k == 1 ? 3u : 4u)) : k;
        if (seg >= j.nseg) return;
        seg_enum_body<NT>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)f), par, (int)f, (int)seg, (int)chalf, seg_smem);
    } else if (blockIdx.x < gridDim.x - SEG_NFILT) {
        const unsigned r = blockIdx.x - nbig * max_nseg * halves, per = (max_nseg + small_segs - 1) / small_segs;
        const unsigned f = r / per ? 2u : 0u, seg0 = (r % per) * small_segs;
        if (seg0 >= j.nseg) return;
        seg_enum_small_body<NT>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)f), par, (int)f, (int)seg0, seg_smem);
    } else {
        seg_first_body<NT, false>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)(blockIdx.x - (gridDim.x - SEG_NFILT))), par, (int)(blockIdx.x - (gridDim.x - SEG_NFILT)), seg_smem);
    }
}

/* seeded state sets (SegParams::seeded): every filter through seg_enum_seeded_body, one workgroup per (filter, segment, channel group); the
 * last five walk the epoch's first segment */
template <int NT>
__global__ __launch_bounds__(NT) void seg_k_enum_seeded(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par, unsigned max_nseg)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    constexpr unsigned halves = 4 / (NT / SEG_NSP);
    if (blockIdx.x < SEG_NFILT * max_nseg * halves) {
        const unsigned f = blockIdx.x / (max_nseg * halves), r = blockIdx.x % (max_nseg * halves), seg = r / halves, chalf = r % halves;
        if (seg >= j.nseg) return;
        seg_enum_seeded_body<NT>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)f), par, (int)f, (int)seg, (int)chalf, seg_smem);
    } else {
        seg_first_body<NT, false>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)(blockIdx.x - SEG_NFILT * max_nseg * halves)), par, (int)(blockIdx.x - SEG_NFILT * max_nseg * halves), seg_smem);
    }
}

/* enumeration in UNITS (batches; SegParams::unit = SEG_UNIT): first the filters that look at the left pixel (their workgroups are the long ones: `perb`
 * workgroups of SEG_UNC (unit, channel) pairs per candidate), then none / up -- with their small state set (when it exists) segment by segment, `pers`
 * workgroups of SEG_UNC_SMALL (unit, channel) pairs --, and the five walkers of an epoch's first unit */
/* (the second bound asks for 8 waves per SIMD: the body's 100 SGPRs held it at 7 -- three workgroups of 8 waves per CU where LDS and threads allow four; with 78 + spills to
 *  vector lanes a batch of more workgroups than slots gains: 96 frames of 1080p 312 -> 301 ms, 128: 403 -> 392; 16 ... 64 frames within +-0.7 %) */
/* `seeds` (round 6): the launcher offers the start from seeds (seg_unit_from_seeds decides per image, candidate and attempt); perb is then sized for whichever of the two
 * bodies needs more workgroups (the exhaustive one: SEG_UNC pairs a workgroup against SEG_UNC_SEEDS).
 * UNIT = SEG_UNIT: batches composed in units.  UNIT = 1 (round 6): the SAME bodies segment by segment -- (segment, channel) pairs, sixteen a workgroup, each started from
 * seeds eight pixels in front of it -- for small and mid-size batches, whose attempts are bound by the enumeration's dependent path, not by its work: 8 + 32 dependent
 * steps instead of 8 + 96, a twentieth of the workgroups of seg_k_enum (one per segment and channel pair, every segment from all 253 states). */
template <int UNIT>
__global__ __launch_bounds__(SEG_UNT, 8) void seg_k_enum_unit(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par, unsigned perb, unsigned pers, int seeds)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    const bool small_ok = P->small_ok != 0;
    constexpr int NCS = SEG_UNC_SMALL_OF(UNIT);
    const unsigned nbig = small_ok ? 3u : 5u, nb = nbig * perb, ns = small_ok ? 2u * pers : 0u;
    if (blockIdx.x >= nb + ns) {
        seg_first_body<SEG_UNT, (UNIT > 1)>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)(blockIdx.x - nb - ns)), par, (int)(blockIdx.x - nb - ns), seg_smem);
        return;
    }
    const unsigned npairs = ((j.nseg + UNIT - 1) / UNIT) * j.bpp;
    if (blockIdx.x < nb) {
        const unsigned k = blockIdx.x / perb, grp = blockIdx.x % perb;
        const unsigned f = small_ok ? (k == 0 ? 1u : (k == 1 ? 3u : 4u)) : k;
        const SegCtlView cv = seg_view_of(sj + blockIdx.y, par, (int)f);
        if (seeds && seg_unit_from_seeds(j, *P, cv, (int)f, seeds)) {
            if (grp * SEG_UNC_SEEDS_OF(UNIT) >= npairs) return;
            seg_enum_unit_body<SEG_SEED_LANES, UNIT, SEG_UNC_SEEDS_OF(UNIT), true>(j, *P, cv, par, (int)f, (int)grp, seg_smem);
            return;
        }
        if (grp * SEG_UNC >= npairs) return;
        seg_enum_unit_body<SEG_NSP, UNIT, SEG_UNC>(j, *P, cv, par, (int)f, (int)grp, seg_smem);
    } else {
        const unsigned r = blockIdx.x - nb, f = r / pers ? 2u : 0u, grp = r % pers;
        if (grp * NCS >= npairs) return;
        seg_enum_unit_body<SEG_NSS, UNIT, NCS>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)f), par, (int)f, (int)grp, seg_smem);
    }
}

template <bool SEEDED, int CT, bool UNITS>
__global__ __launch_bounds__(CT) void seg_k_chain(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    if (blockIdx.x == 0) { seg_extremes_body<CT>(j, *P, seg_view_of(sj + blockIdx.y, par, 0), par, seg_smem); return; }      /* (the spare workgroup, dispatched first: the row's extremes for none's bound) */
    seg_chain_body<SEEDED, CT, UNITS>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)((blockIdx.x - 1) >> 2)), par, (int)((blockIdx.x - 1) >> 2), (int)((blockIdx.x - 1) & 3), seg_smem);
}

template <int RNT>
__global__ __launch_bounds__(RNT) void seg_k_replay(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par, unsigned max_ngrp)
{
    extern __shared__ __align__(16) unsigned char seg_smem[];
    const SegJob j = sj[blockIdx.y];
    const unsigned f = blockIdx.x / max_ngrp, grp = blockIdx.x % max_ngrp;
    if (grp >= j.ngrp) return;
    seg_replay_body<RNT>(j, *P, seg_view_of(sj + blockIdx.y, par, (int)f), par, (int)f, (int)grp, seg_smem);
}

/* The ORDER of the kernels in the code object, pinned: the one-image kernels first, in the order round 4's library had them, the kernels of batches behind
 * them.  (Measured, profiles/r05_code_layout.txt: with the batch kernels emitted in between, the control kernel -- byte for byte the same instructions --
 * took 0.45 us longer per launch, the enumeration 0.5: the headline lost 3 %.) */
template __global__ void seg_k_ctl<SEG_TPARTS>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned, unsigned);
template __global__ void seg_k_replay<SEG_REPLAY_NT>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
template __global__ void seg_k_chain<false, SEG_CHAIN_THREADS, false>(const SegJob *__restrict__, const SegParams *__restrict__, int);
template __global__ void seg_k_chain<true, SEG_CHAIN_THREADS, false>(const SegJob *__restrict__, const SegParams *__restrict__, int);
template __global__ void seg_k_enum_seeded<512>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
template __global__ void seg_k_enum_seeded<1024>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
template __global__ void seg_k_enum<512>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
template __global__ void seg_k_enum<1024>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
template __global__ void seg_k_ctl<SEG_TPARTS_BATCH>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned, unsigned);
template __global__ void seg_k_enum_unit<SEG_UNIT>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned, unsigned, int);
template __global__ void seg_k_chain<false, SEG_CHAIN_THREADS_UNIT, true>(const SegJob *__restrict__, const SegParams *__restrict__, int);
template __global__ void seg_k_replay<SEG_REPLAY_NT_BATCH>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned);
#if SEG_UNIT != 1
template __global__ void seg_k_enum_unit<1>(const SegJob *__restrict__, const SegParams *__restrict__, int, unsigned, unsigned, int);      /* (round 6; behind the pinned kernels) */
#endif

/* seeded state sets: the dense transitions of the enumerated segments, between the enumeration and the chain (seg_gather_seeded_body); 5 x 4 x nblk workgroups.
 * (behind the pinned kernels: it joins the code object at its end) */
__global__ __launch_bounds__(SEG_GT) void seg_k_gather_seeded(const SegJob *__restrict__ sj, const SegParams *__restrict__ P, int par, unsigned nblk)
{
    (void)P;
    const SegJob j = sj[blockIdx.y];
    const unsigned fc = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    if (blk * SEG_GS + 1u >= j.nseg) return;
    seg_gather_seeded_body(j, seg_view_of(sj + blockIdx.y, par, (int)(fc >> 2)), (int)(fc >> 2), (int)(fc & 3), (int)blk);
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

/* the kernels of the engine whose dynamic LDS can exceed 64 KB: opted in per device (pl_lds_optin) */
hipError_t chain_attr()
{
    static std::atomic<unsigned> done_chain{ 0 }, done_ctl{ 0 };
    static std::atomic<unsigned> done_chain_s{ 0 };
    static std::atomic<unsigned> done_chain_u{ 0 };
    hipError_t e = pl_lds_optin((const void *)seg_k_chain<false, SEG_CHAIN_THREADS, false>, SEG_SM_CHAIN(SEG_CHAIN_CAP + 1), done_chain);
    if (e == hipSuccess) e = pl_lds_optin((const void *)seg_k_chain<false, SEG_CHAIN_THREADS_UNIT, true>, SEG_SM_CHAIN(SEG_CHAIN_CAP + 1), done_chain_u);
    if (e == hipSuccess) e = pl_lds_optin((const void *)seg_k_chain<true, SEG_CHAIN_THREADS, false>, SEG_SM_CHAIN(SEG_CHAIN_CAP + 1), done_chain_s);
    static std::atomic<unsigned> done_ctl1{ 0 };
    if (e == hipSuccess && SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS)) > 65536) e = pl_lds_optin((const void *)seg_k_ctl<SEG_TPARTS>, SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS)), done_ctl);
    if (e == hipSuccess && SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS_BATCH)) > 65536) e = pl_lds_optin((const void *)seg_k_ctl<SEG_TPARTS_BATCH>, SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS_BATCH)), done_ctl1);
    return e;
}
static_assert(SEG_SM_REPLAY <= 65536 && SEG_SM_ENUM_NT(1024) <= 65536 && SEG_SM_ENUM_SEEDED(1024) <= 65536 && SEG_SM_ENUM_UNIT <= 65536, "these kernels are launched without an LDS opt-in");
static_assert((SEG_TBL_WORDS + 1024 + 512) * 4 + SEG_UNIT * SEG_L * 4 * 8 <= SEG_SM_ENUM_UNIT && (SEG_TBL_WORDS + 1024 + 512) * 4 + SEG_L * 4 * 8 <= SEG_SM_ENUM_NT(512), "seg_first_body's carve fits the enumeration kernels' LDS");

} // namespace

PlSegLayout pl_seg_layout(uint32_t width, uint32_t nsp, bool seeded)
{
    PlSegLayout l{};
    l.nseg = (width + SEG_L - 1) / SEG_L;
    l.ngrp = (l.nseg + SEG_GRP - 1) / SEG_GRP;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align256(o + (bytes ? bytes : 4)); return at; };
    l.ctl = take(3 * sizeof(SegCtl));
    l.base = take(3 * SEG_NFILT * 256 * 4);
    l.h0 = take(3 * 256 * 4);
    l.acc = take(3 * sizeof(SegAcc));
    l.err0 = take((size_t)width * 16);
    l.err1 = take((size_t)width * 16);
    l.rowcopy = take((size_t)width * 12);
    l.tables = take((size_t)SEG_NFILT * SEG_TBL_WORDS * 4);
    l.maps = take(seeded ? 0 : (size_t)SEG_NFILT * l.nseg * 4 * nsp * 2);
    l.ehash = take(seeded ? (size_t)SEG_NFILT * l.nseg * 4 * SEG_EH_WORDS * 4 : 0);
    l.rout = take((size_t)SEG_NFILT * l.nseg * 4 * SEG_NSP * 2);
    l.rst = take((size_t)SEG_NFILT * l.nseg * 4 * SEG_NSP * 4);
    l.rck = take((size_t)SEG_NFILT * l.nseg * 4 * SEG_NSP * (SEG_PARTS - 1) * 4);
    l.dnout = take((size_t)SEG_NFILT * l.nseg * 4 * 2);
    l.dcnt = take((size_t)SEG_NFILT * l.nseg * 4 * 4);
    l.entry = take((size_t)SEG_NFILT * l.nseg * 4 * 4);
    l.segcnt = take((size_t)SEG_NFILT * l.nseg * 256 * 2);
    l.grpcnt = take((size_t)SEG_NFILT * l.ngrp * 256 * 4);
    l.grpleft = take((size_t)SEG_NFILT * l.ngrp * 4);
    l.firstidx = take(SEG_NFILT * 4 * 2 * 4);
    l.rowmm = take(16);
    l.total = o;
    return l;
}

bool pl_seg_supported(const uint32_t *widths, size_t n, unsigned strength, long bleed, SegParams *params_out)
{
    if (bleed < 1 || bleed > 32767 || strength > 255) return false;
    for (size_t i = 0; i < n; i++)
        if (widths[i] > SEG_MAX_WIDTH) return false;
    return seg_build_params(*params_out, (int)strength, (int)bleed);
}

hipError_t pl_seg_launch_resolve(const PlJob *d_jobs, SegJob *d_sj, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(seg_k_resolve, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_jobs, d_sj, (unsigned)n);
    return hipGetLastError();
}

hipError_t pl_seg_launch_attempt(const PlSegBatch &b, int attempt, hipStream_t stream)
{
    if (!b.n) return hipSuccess;
    hipError_t e = chain_attr();
    if (e != hipSuccess) return e;
    const int par = attempt % 3;                               /* which copy of the control block / sums / histogram / prefix bumps the attempt writes */
    const unsigned n = (unsigned)b.n;
    {
        /* (the validation workgroups can be left out at COMPILE time only -- SEG_EXPERIMENT_NO_VAL_CODE, a timing experiment whose results are unvalidated;
         *  the shipped library has no run-time switch that changes what it computes) */
        const unsigned nctl = SEG_NFILT * b.tparts + 1 + b.max_ncommit, nval = SEG_EXPERIMENT_NO_VAL_CODE ? 0u : SEG_NFILT * b.max_ngrp * (SEG_GRP / SEG_VGRP_OF(b.tparts));
        if (b.tparts == SEG_TPARTS_BATCH) hipLaunchKernelGGL(seg_k_ctl<SEG_TPARTS_BATCH>, dim3(nctl + nval, n), dim3(SEG_THREADS), SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS_BATCH)), stream, b.d_sj, b.d_params, par, nctl, b.max_ngrp);
        else hipLaunchKernelGGL(seg_k_ctl<SEG_TPARTS>, dim3(nctl + nval, n), dim3(SEG_THREADS), SEG_SM_CTLVAL_V(SEG_VGRP_OF(SEG_TPARTS)), stream, b.d_sj, b.d_params, par, nctl, b.max_ngrp);
    }
    {
        const bool small_ok = b.small_ok;
        const unsigned nt = b.enum_nt, halves = 4 / (nt / SEG_NSP), small_segs = nt / (4 * SEG_NSS);
        const unsigned blocks = (small_ok ? 3 * b.max_nseg * halves + 2 * ((b.max_nseg + small_segs - 1) / small_segs) : SEG_NFILT * b.max_nseg * halves) + SEG_NFILT;
        const size_t enum_lds = (size_t)SEG_SM_ENUM_NT(nt);
        if (b.unit == 1 && b.seeds && !b.seeded) {
            /* (round 6) small and mid-size batches: segment by segment, from seeds, through the unit enumeration's bodies */
            const unsigned pairs = b.max_nseg * 4, nc_min = SEG_UNC_SEEDS1 < SEG_UNC ? SEG_UNC_SEEDS1 : SEG_UNC;
            const unsigned perb = (pairs + nc_min - 1) / nc_min, pers = (pairs + SEG_UNC_SMALL_OF(1) - 1) / SEG_UNC_SMALL_OF(1);
            const unsigned ublocks = (small_ok ? 3 * perb + 2 * pers : SEG_NFILT * perb) + SEG_NFILT;
            hipLaunchKernelGGL(seg_k_enum_unit<1>, dim3(ublocks, n), dim3(SEG_UNT), (size_t)SEG_SM_ENUM_UNIT, stream, b.d_sj, b.d_params, par, perb, pers, 1);
        } else
        if (b.unit > 1 && !b.seeded) {
            const unsigned pairs = ((b.max_nseg + SEG_UNIT - 1) / SEG_UNIT) * 4, nc_min = b.seeds && SEG_UNC_SEEDS < SEG_UNC ? SEG_UNC_SEEDS : SEG_UNC;      /* (workgroups for whichever body takes fewer pairs each) */
            const unsigned perb = (pairs + nc_min - 1) / nc_min, pers = (pairs + SEG_UNC_SMALL - 1) / SEG_UNC_SMALL;
            const unsigned blocks = (small_ok ? 3 * perb + 2 * pers : SEG_NFILT * perb) + SEG_NFILT;
            hipLaunchKernelGGL(seg_k_enum_unit<SEG_UNIT>, dim3(blocks, n), dim3(SEG_UNT), (size_t)SEG_SM_ENUM_UNIT, stream, b.d_sj, b.d_params, par, perb, pers, b.seeds ? 1 : 0);
        } else
        if (b.seeded) {
            const unsigned sblocks = SEG_NFILT * b.max_nseg * halves + SEG_NFILT;
            if (nt == 512) hipLaunchKernelGGL(seg_k_enum_seeded<512>, dim3(sblocks, n), dim3(512), (size_t)SEG_SM_ENUM_SEEDED(512), stream, b.d_sj, b.d_params, par, b.max_nseg);
            else hipLaunchKernelGGL(seg_k_enum_seeded<1024>, dim3(sblocks, n), dim3(1024), (size_t)SEG_SM_ENUM_SEEDED(1024), stream, b.d_sj, b.d_params, par, b.max_nseg);
        } else
        if (nt == 512) hipLaunchKernelGGL(seg_k_enum<512>, dim3(blocks, n), dim3(512), enum_lds, stream, b.d_sj, b.d_params, par, b.max_nseg);
        else hipLaunchKernelGGL(seg_k_enum<1024>, dim3(blocks, n), dim3(1024), enum_lds, stream, b.d_sj, b.d_params, par, b.max_nseg);
    }
    if (b.seeded && b.max_nseg > 1) {
        const unsigned nblk = (b.max_nseg - 1 + SEG_GS - 1) / SEG_GS;
        hipLaunchKernelGGL(seg_k_gather_seeded, dim3(SEG_NFILT * 4 * nblk, n), dim3(SEG_GT), 0, stream, b.d_sj, b.d_params, par, nblk);
    }
    if (b.seeded) hipLaunchKernelGGL((seg_k_chain<true, SEG_CHAIN_THREADS, false>), dim3(SEG_NFILT * 4 + 1, n), dim3(SEG_CHAIN_THREADS), SEG_SM_CHAIN(b.max_nseg), stream, b.d_sj, b.d_params, par);
    else if (b.unit > 1) hipLaunchKernelGGL((seg_k_chain<false, SEG_CHAIN_THREADS_UNIT, true>), dim3(SEG_NFILT * 4 + 1, n), dim3(SEG_CHAIN_THREADS_UNIT), SEG_SM_CHAIN_X((b.max_nseg + b.unit - 1) / b.unit), stream, b.d_sj, b.d_params, par);
    else hipLaunchKernelGGL((seg_k_chain<false, SEG_CHAIN_THREADS, false>), dim3(SEG_NFILT * 4 + 1, n), dim3(SEG_CHAIN_THREADS), SEG_SM_CHAIN_X(b.max_nseg), stream, b.d_sj, b.d_params, par);
    if (b.unit > 1) hipLaunchKernelGGL(seg_k_replay<SEG_REPLAY_NT_BATCH>, dim3(SEG_NFILT * b.max_ngrp, n), dim3(SEG_REPLAY_NT_BATCH), SEG_SM_REPLAY, stream, b.d_sj, b.d_params, par, b.max_ngrp);
    else hipLaunchKernelGGL(seg_k_replay<SEG_REPLAY_NT>, dim3(SEG_NFILT * b.max_ngrp, n), dim3(SEG_REPLAY_NT), SEG_SM_REPLAY, stream, b.d_sj, b.d_params, par, b.max_ngrp);
    return hipGetLastError();
}
