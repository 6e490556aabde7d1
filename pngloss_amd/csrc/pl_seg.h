/*
 * pl_seg.h -- host-side interface of the segment-parallel row engine (pl_seg.hip / pl_seg_core.h).  Internal.
 */
#ifndef PL_SEG_H
#define PL_SEG_H

#include "pl_device.h"
#include "pl_seg_core.h"

/* per-image device workspace of the engine, beyond what PlJob already has (offsets into one 256-B aligned carve) */
struct PlSegLayout { size_t ctl, base, h0, acc, err0, err1, rowcopy, tables, maps, ehash, rout, rst, rck, dnout, dcnt, entry, segcnt, grpcnt, grpleft, firstidx, rowmm, total; uint32_t nseg, ngrp; };
PlSegLayout pl_seg_layout(uint32_t width, uint32_t nsp, bool seeded);   /* nsp, seeded: SegParams::nsp / ::seeded of the (strength, bleed) pair */

/* can the engine take this batch?  (chain states of (strength, bleed) fit the lanes, every row fits the chain kernel) */
bool pl_seg_supported(const uint32_t *widths, size_t n, unsigned strength, long bleed, SegParams *params_out);

struct PlSegBatch {
    const SegJob *d_sj;       /* device: one per image */
    const SegParams *d_params;
    size_t n;
    uint32_t max_nseg, max_ngrp, max_ncommit;
    uint32_t enum_nt;         /* threads of the enumeration's workgroups: 512 or 1024 (SEG_ENUM_NT_SMALL_MAX_NSEG) */
    bool small_ok;            /* SegParams::small_ok (none / up enumerated with their own small state set) */
    bool seeded;              /* SegParams::seeded (seg_k_enum_seeded) */
    uint32_t tparts;          /* SegParams::tparts */
    uint32_t unit;            /* SegParams::unit: 1, or SEG_UNIT = enumeration in units (seg_k_enum_unit; batches) */
    bool seeds;               /* (units only) the units may start from seeds with a run-in instead of from every state (SegParams::seed_n > 0; seg_unit_from_seeds decides per image, candidate and attempt) */
};

/* fills sj[i].bpp from the class the prepare kernels detected */
hipError_t pl_seg_launch_resolve(const PlJob *d_jobs, SegJob *d_sj, size_t n, hipStream_t stream);
/* one attempt = [control + validation of the attempt before], enumerate, chain, replay (attempt: counted by the caller, any starting point that is a multiple of 3) */
hipError_t pl_seg_launch_attempt(const PlSegBatch &b, int attempt, hipStream_t stream);

#endif
