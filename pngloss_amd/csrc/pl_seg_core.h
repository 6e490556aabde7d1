/*
 * pl_seg_core.h -- the SEGMENT-PARALLEL row engine ("latency mode"): one image spread over the whole GPU.
 *
 * Replaces, like pl_engine.hip, the reference's optimize_image / optimize_state_row / optimize_state_run
 * (/root/reference/src/pngloss_image.c:159-333, /root/reference/src/optimize_state.c:114-361), but cuts the x-chain of a row
 * into segments that run concurrently.  Why that is possible (measured: oracle/frozen_study.c, profiles/r03_frozen_study.txt):
 *
 *   The only thing that couples a pixel to ALL earlier pixels of its row is the running symbol histogram
 *   (optimize_state.c:221,253).  If every decision of a row is taken against the histogram FROZEN at the start of the row,
 *   the decision differs from the reference's in < 1e-5 of the cases (sub, up, average, paeth: 0.002-0.02 mismatches per row on
 *   every image tried; none: 0.3-0.9 per row where its P and N bands compete).  Against a frozen histogram the four channels
 *   decouple, and one channel's chain is a FINITE-STATE MACHINE: its state in front of pixel x is
 *       (left = reconstructed byte of x-1,  cn = rem(diff[x-1]) + thr(diff[x-2]),  th = thr(diff[x-1]))
 *   (optimize_state.c:146,172,455,467) -- relative to the data at most 253 states for s = 19, bleed = 2.  So:
 *
 *   ENUMERATE  (seg_enum_body)    every segment of 32 pixels is run from EVERY possible entry state at once (lane = state; up to
 *                                 1024 states in chunks of 256 lanes); the trajectories merge within a few pixels, so after 4 steps
 *                                 the distinct states (a few dozen) are given dense ids and only they run on.  Out: entry state ->
 *                                 dense id, dense id -> exit state, dense id -> state at every quarter of the segment (checkpoints);
 *                                 BATCHES enumerate in units of three segments (seg_enum_unit_body), and since round 6 a unit -- or, for small batches, a
 *                                 segment -- is not started from every state but from SEEDS with a run-in of eight pixels (one per left byte within reach of
 *                                 the data): what they have become at its first pixel is its entry set; a row whose true state none of them reached is
 *                                 broken off there by the chain and finishes from every state (costs attempts, never bytes: see VALIDATE);
 *   CHAIN      (seg_chain_body)   the dense transition tables of consecutive segments are composed from the row's true start
 *                                 state: one table lookup per segment instead of 32 dependent pixel steps -- this is where the
 *                                 serial chain of W steps shrinks;
 *   REPLAY     (seg_replay_body)  every quarter of every segment is run once more from its now known state (entry state or
 *                                 checkpoint) and writes the candidate row;
 *   VALIDATE   (seg_post_body)    THE GROUND TRUTH: every decision of the candidate row is re-derived from its predecessors'
 *                                 outputs and checked against the reference's arg-max rule under the exact RUNNING histogram
 *                                 (block counts + in-segment counting).  A row that passes is, by induction over x, exactly
 *                                 the reference's row -- whatever tables, maps or states produced it.  The first decision that
 *                                 fails is evaluated exactly (seg_ctl_body), the histogram is re-frozen behind it and the rest
 *                                 of the row is speculated again ("epoch"); progress is at least one pixel per attempt, and a
 *                                 candidate that keeps failing finishes its row serially.  The same kernel computes the
 *                                 derivative error, libpng's heuristic and the entropy cost (optimize_state.c:265-342,492-562);
 *   CONTROL    (seg_ctl_body)     winner (strict <, pngloss_image.c:257), strength retry (:266-274), commit (:277-308),
 *                                 decision tables of the next row.
 *
 * One "attempt" = these steps, four launches since round 4 (the validation rides in the control kernel's launch, one attempt behind); all state lives in device memory, so the host only enqueues attempts (no data-dependent
 * host control flow, no in-kernel grid barrier: a kernel boundary is the cheapest grid-wide sync on this machine).
 *
 * This header is compiled twice: by hipcc into the kernels of pl_seg.hip, and by g++ into tests/c/seg_host.cpp, which runs the
 * very same bodies as plain loops over (block, thread) -- test infrastructure that lets the CPU suite prove the logic bit-exact
 * against the oracle without a GPU.  The product has no CPU path: only the kernels are shipped.
 */
#ifndef PL_SEG_CORE_H
#define PL_SEG_CORE_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#define PLS_HD __host__ __device__ __forceinline__
#else
#define PLS_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PLS_THREADS(tid, nt) for (int tid = (int)threadIdx.x, pls_once_ = 1; pls_once_ && tid < (int)(nt); pls_once_ = 0)
#define PLS_SYNC() __syncthreads()
#define PLS_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define PLS_ATOMIC_MIN(p, v) atomicMin((p), (v))
#define PLS_ATOMIC_OR(p, v) atomicOr((p), (v))
#define PLS_ATOMIC_CAS(p, cmp, v) atomicCAS((p), (cmp), (v))
#define PLS_ATOMIC_ADD_RET(p, v) atomicAdd((p), (v))
#define PLS_ATOMIC_EXCH(p, v) atomicExch((p), (v))
#define PLS_ATOMIC_ADD64(p, v) atomicAdd((unsigned long long *)(p), (unsigned long long)(v))
#define PLS_CLOCK() wall_clock64()
#define PLS_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define PLS_WAVE_LEADER(tid) (((tid) & 63) == 0)
__device__ __forceinline__ uint32_t pls_wave_sum_u32(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ uint64_t pls_wave_sum_u64(uint64_t v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ int pls_wave_max_i(int v) { for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; } return v; }
__device__ __forceinline__ int pls_wave_min_i(int v) { for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t < v ? t : v; } return v; }
#define PLS_ATOMIC_MAX_I(p, v) atomicMax((p), (v))
#define PLS_ATOMIC_MAX_U(p, v) atomicMax((p), (v))
#define PLS_ATOMIC_MIN_I(p, v) atomicMin((p), (v))
#define PLS_HOST_VISIBLE_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define PLS_HOST_VISIBLE_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#else
#define PLS_THREADS(tid, nt) for (int tid = 0; tid < (int)(nt); ++tid)
#define PLS_SYNC() ((void)0)
#define PLS_ATOMIC_ADD(p, v) (*(p) += (v))
#define PLS_ATOMIC_MIN(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define PLS_ATOMIC_OR(p, v) (*(p) |= (v))
inline uint32_t pls_host_cas(uint32_t *p, uint32_t cmp, uint32_t v) { const uint32_t old = *p; if (old == cmp) *p = v; return old; }
inline uint32_t pls_host_add_ret(uint32_t *p, uint32_t v) { const uint32_t old = *p; *p += v; return old; }
#define PLS_ATOMIC_CAS(p, cmp, v) pls_host_cas((p), (cmp), (v))
#define PLS_ATOMIC_ADD_RET(p, v) pls_host_add_ret((p), (v))
#define PLS_ATOMIC_EXCH(p, v) (*(p) = (v))
#define PLS_ATOMIC_ADD64(p, v) (*(p) += (v))
/* the CPU harness runs one "thread" at a time: every thread is its own wave */
#define PLS_CLOCK() 0ull
#define PLS_ATOMIC_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define PLS_WAVE_LEADER(tid) (true)
inline uint32_t pls_wave_sum_u32(uint32_t v) { return v; }
inline uint64_t pls_wave_sum_u64(uint64_t v) { return v; }
inline int pls_wave_max_i(int v) { return v; }
inline int pls_wave_min_i(int v) { return v; }
#define PLS_ATOMIC_MAX_I(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define PLS_ATOMIC_MAX_U(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define PLS_ATOMIC_MIN_I(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
#define PLS_HOST_VISIBLE_ADD(p, v) (*(p) += (v))
#define PLS_HOST_VISIBLE_STORE(p, v) (*(p) = (v))
#endif

/* Pointers with their address space named (device): shared memory -> ds_* instructions, device memory -> global_* instead of FLAT
 * (a FLAT access counts in both wait counters, so every wait for an LDS result would also wait for outstanding global traffic, and
 * LDS through FLAT is several times slower).  On the host they are plain pointers of the same size. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SEG_PLAIN_POINTERS)   /* (a translation unit without kernels of this engine sets SEG_PLAIN_POINTERS) */
#define SEG_AS_LDS __attribute__((address_space(3)))
#define SEG_AS_GLB __attribute__((address_space(1)))
#else
#define SEG_AS_LDS
#define SEG_AS_GLB
#endif
typedef SEG_AS_LDS uint32_t *seg_lds_u32;
typedef SEG_AS_LDS uint8_t *seg_lds_u8;
typedef SEG_AS_LDS uint16_t *seg_lds_u16;

#if defined(__HIPCC__)
#define PLS_UNROLL _Pragma("unroll")
#else
#define PLS_UNROLL
#endif
#define SEG_NFILT 5
#ifndef SEG_EXPERIMENT_WG_SPAN
#define SEG_EXPERIMENT_WG_SPAN 0    /* experiment build (tools/replay_clocks.sh, EXP_DEFS=-DSEG_EXPERIMENT_WG_SPAN=1): when the enumeration's workgroups start and end, relative to the launch's first */
#endif
#ifndef SEG_EXPERIMENT_REPLAY_CLOCKS
#define SEG_EXPERIMENT_REPLAY_CLOCKS 0   /* (1: an experiment build in which the REPLAY's phase clocks take the enumeration's slots of the result record; tools/replay_clocks.sh) */
#endif
#ifndef SEG_DEBUG_ROW
#define SEG_DEBUG_ROW(kind, failed, winner, start_none)
#endif
#ifndef SEG_DEBUG_REPAIR
#define SEG_DEBUG_REPAIR(f, c, sg, est, sid)
#endif
#ifndef SEG_DEBUG_BREAK
#define SEG_DEBUG_BREAK(f, c, sg, est, y)   /* (the CPU harness can list where the chain kernel broke a row off: a unit whose entry state no enumerated state reached) */
#endif
#ifndef SEG_DEBUG_STATE
#define SEG_DEBUG_STATE(f, k, sl, i, ps)   /* (the CPU harness can count the distinct states of a pair at every segment's end: how fast a unit's states keep merging) */
#endif
#ifndef SEG_DEBUG_COUNT
#define SEG_DEBUG_COUNT(slot, v)   /* (the CPU harness counts a few things the tests pin) */
#endif
#ifndef SEG_L
#define SEG_L 32                 /* pixels per segment (16 was measured: enumeration -5 us, chain +5 us, no gain) */
#endif
#ifndef SEG_UNIT
#define SEG_UNIT 3               /* segments per enumeration UNIT when the launcher asks for units (SegParams::unit; batches).  Measured (profiles/r05_unit_groups.txt): 2, 3, 4 within 3 % of each other from 16 frames of 1080p on, 3 best at 32 and 64; 8 loses below 64 frames */
#endif
#define SEG_UNIT_MIN_SEGS 680    /* the launcher enumerates in units when the batch's images have more segments than this between them (twelve frames of 1920 pixels; measured with two
                                    launch groups, profiles/r05_suite_groups.txt: 8 / 10 / 12 / 16 frames per segment 87 / 99 / 111 / 137 ms, in units 105 / 106 / 108 / 121) */
/* round 6, batches whose (strength, bleed) pair has a seed set (SegParams::seed_n): between SEG_SEEDS1_MIN_SEGS and SEG_UNIT_MIN_SEGS_SEEDS segments they are enumerated
 * segment by segment from seeds (seg_k_enum_unit<1>), beyond in units from seeds, below like one image (seg_k_enum: every segment from every state -- the shortest dependent
 * path).  Measured, 1080p frames of 60 segments, ms per batch, [from every state per segment | per segment from seeds | units from seeds]: 2 frames 48.9 | 53.6 | -,
 * 4: 60.7 | 60.7 | 92, 6: 73.1 | 65.4 | -, 8: 86.3 | 74.9 | 95.7, 11: 106 | 81.8 | 97.5, 12: - | 83.7 | 100.9, 16: - | 97.5 | 103.2, 20: - | 114 | ~108, 24: - | 132 | 113.8, 32: - | 166 | 123.5;
 * the reference's suite as one batch (182 segments, flat content whose fixed points the seeds miss): 59.4 | 66.8 | - (profiles/r06_seeds.txt). */
#define SEG_SEEDS1_MIN_SEGS 330
#define SEG_SEEDS1_MIN_SEGS_PER_IMAGE 32    /* ... of images a thousand pixels wide on average: batches of NARROW images keep the start from every state (measured on the suite's photographs, 512 .. 768 pixels wide:
                                               20 of them -- 392 segments -- 39.4 ms from every state, 47.4 segment by segment from seeds; 40: 61.6 / 57.8; profiles/r06_photo_batch.txt) */
#define SEG_UNIT_MIN_SEGS_SEEDS 1000
#if !defined(SEG_UNC) && SEG_UNIT > 6
#define SEG_UNC 7                /* (longer units: fewer pairs, so that their pixel records fit the 16 KB the workgroup's shared memory has for them) */
#endif
#ifndef SEG_UNC
#define SEG_UNC 10               /* (unit, channel) pairs per workgroup of the unit enumeration (measured 9 / 10 / 11 / 12: 175.9 / 163.5 / 166.8 / 170.4 ms for 32 frames of 1080p: ~170 distinct states = three waves, and 80 pairs of a 1920-pixel row = 8 workgroups) */
#endif
#ifndef SEG_UNC_SEEDS
#define SEG_UNC_SEEDS 16         /* (unit, channel) pairs per workgroup of the unit enumeration FROM SEEDS (SEG_SEED_LANES lanes a pair: two turns of eight pairs) */
#endif
#ifndef SEG_UNC_SEEDS1
#define SEG_UNC_SEEDS1 8         /* ... and segment by segment (seg_k_enum_unit<1>): ONE turn of eight pairs -- what those batches pay for is the dependent path (measured, 1080p frames,
                                    8 against 16 pairs: 2 / 6 / 11 / 16 frames 53.6 / 65.4 / 81.8 / 97.5 against 59.5 / 72.7 / 86.5 / 98.5 ms; 24 / 32: 132 / 166 against 128 / 160) */
#endif
#define SEG_UNC_SEEDS_OF(unit) ((unit) == 1 ? SEG_UNC_SEEDS1 : SEG_UNC_SEEDS)
#ifndef SEG_UNT
#define SEG_UNT 512              /* its threads (512 against 1024, 1080p frames: 157.7 against 161.4 ms at 32, 249.2 against 281.4 at 64, 460.6 against 511.3 at 128: the CU starts four workgroups at once instead of two) */
#endif
#define SEG_UPOOL SEG_UNT        /* ... and the most distinct states its pairs may have between them: one LANE each (round 5: it said 1024 for 512-thread workgroups -- states beyond the lanes had no one to walk them) */
#define SEG_GRP 16               /* segments per group (replay / validation workgroup) */
#define SEG_VGRP 8                /* segments per VALIDATION workgroup of ONE image (half a replay group: one decision per thread, twice the CUs) ... */
#define SEG_VGRP_UNITS 16         /* ... and of batches composed in units (the whole replay group, two decisions per thread: half the workgroups, each with the same staging round trips and barriers --
                                     what a saturated GPU pays for; measured, 1080p frames in one batch, 8 against 16: 12 / 16 frames 229 / 300 -> 228 / 299 Mpx/s, 32: 469 -> 479, 64: 569 -> 600, 128: 602 -> 648) */
#ifndef SEG_TPARTS_BATCH
#define SEG_TPARTS_BATCH 1        /* control workgroups per candidate for batches composed in units (SegParams::tparts there; SEG_TPARTS = 4 for one image) */
#endif
#define SEG_VGRP_OF(tparts) ((tparts) == SEG_TPARTS_BATCH ? SEG_VGRP_UNITS : SEG_VGRP)      /* (the launcher asks for one control workgroup per candidate exactly when it composes in units) */
#define SEG_TPARTS 4              /* control kernel: workgroups that share the build of one candidate's decision tables */
#define SEG_COMMIT_W 256           /* control kernel: pixels per commit workgroup */
#define SEG_CTL_IMG_OF(P) (SEG_NFILT * (P).tparts)      /* blockIdx.x of the image-wide workgroup; the commit workgroups follow */
#define SEG_PARTS 4               /* the replay cuts a segment into this many parts: the enumeration leaves the state at every cut (checkpoints) */
#define SEG_PL (SEG_L / SEG_PARTS)
#define SEG_NSP 256              /* lanes per channel in the enumeration; also the most DISTINCT states a segment may have after the dedupe */
#define SEG_NS_MAX 1024          /* most chain states of a (strength, bleed) pair the engine takes (enumerated in chunks of SEG_NSP) */
#define SEG_TOFF 384             /* decision tables cover v in [-384, 383]: every v a clamped band can hold (the re-centred prediction lies in orig - 127 .. orig + 128, so lo >= -383, hi <= 382) */
#define SEG_TN 768
#define SEG_TBL_WORDS (4 * SEG_TN + 128)  /* pre[2], suf[2], cls[2][256 bytes] */
#define SEG_INVALID 0xFFFFu
#define SEG_NOFAIL 0xFFFFFFFFu
#define SEG_MAGIC 0x5E6C0DE1u
#define SEG_MAX_RESTARTS 12      /* epochs per candidate and row before the rest of the row is done serially */
#define SEG_NG_BURST 16              /* group counts a lane requests in one burst (rows up to 8192 pixels); the groups of wider rows are read in a loop behind it */
#define SEG_MAX_WIDTH (1u << 20)      /* rows the engine takes (libpng's own default limit); the chain kernel walks a row in passes, nothing else depends on the width */
#define SEG_THREADS 1024
#define SEG_CHAIN_THREADS 1024       /* the chain kernel's workgroups for one image (the gather of 256 transitions x 64 ids wants the lanes) ... */
#define SEG_CHAIN_THREADS_UNIT 256   /* ... and when the row is composed in units (batches: a third of the transitions, and a CU holds two workgroups of 1024 threads but eight of 256) */
#define SEG_REPLAY_THREADS (SEG_GRP * SEG_L)   /* pixels of a replay group: the first SEG_REPLAY_THREADS threads of the workgroup take one each, SEG_GRP * SEG_PARTS * 4 of them walk */
#define SEG_REPLAY_NT_BATCH SEG_REPLAY_THREADS  /* ... of batches composed in units: a CU holds two workgroups of 1024 threads, but three of 512 with the 50 KB they ask for */
#define SEG_REPLAY_NT (2 * SEG_REPLAY_THREADS) /* threads of the replay's workgroups: the second half helps with the staging and takes the bump counts while the first takes the sums */
#define SEG_KEYLUT_MAX 8192
#define SEG_NSS 32                /* lanes per channel for none / up */
#define SEG_SMALL_SEGS 8           /* most segments per enumeration workgroup for them: 8 segments x 4 channels x SEG_NSS lanes (1024 threads) */
/* The enumeration's workgroups come in two sizes, NT = 1024 threads (4 channels x SEG_NSP lanes) or 512 (a channel pair).  With the
 * kernel's LDS request at what the bodies really use (SEG_SM_ENUM_NT: 35.5 KB at 512 threads = four workgroups per CU; the generous bound
 * of before allowed three, and rows beyond 3200 pixels then needed a second round of workgroups) the lighter ones win or tie for every
 * single image (engine ms, 512 against 1024 threads: 1920 x 2048: 106.8 / 110.0; 3200 x 2048: 119.1 / 125.3; 4096 x 2048: 119.6 / 121.9;
 * 6144 x 1024: 68.7 / 72.2; 7168 x 1024: 74.1 / 73.7; 8192 x 1024: 80.4 / 82.5) and for small batches (1920 x 1080 frames, n = 4: 84.5 / 85.3,
 * n = 8: 116.3 / 115.1, n = 32: 328.5 / 315.1): the launcher picks 512 while all images' segments together are at most this many. */
#define SEG_ENUM_NT_SMALL_MAX_NSEG 320
#define SEG_KEYS_MAX 2048
/* SEEDED enumeration (state sets beyond SEG_NS_MAX: s = 85 at bleed 1 has ~10^5 chain states): a segment is not run from every state there is
 * but from SEG_NSP seeds per channel, started SEG_KIN pixels IN FRONT of the segment -- every possible left byte with the carried terms that go with
 * it (filters that look at the left pixel), a grid of carried terms (none, up) -- and whatever they have become at the segment's first pixel is
 * the segment's entry set (measured: oracle/seed_study.c -- the reference's own state is in that set in all but 1e-4 .. 1e-3 of the boundaries; the
 * chain kernel walks such a segment step by step).  Entry states are found by value, through a small hash table per segment and channel. */
#ifndef SEG_KA_SEEDED
#define SEG_KA_SEEDED 4           /* steps every seed of the seeded enumeration takes before the distinct states go on alone */
#endif
#define SEG_KIN 32                /* most run-in pixels of the seeded enumeration (SegParams::kin: 16 .. 32 by the size of the carried terms) */
#define SEG_SEED_LANES 64         /* lanes (most seeds) per (unit, channel) pair of the unit enumeration from seeds */
#define SEG_SEED_LONG_AFTER 2u      /* rows an image has had broken off before its run-ins get half as long again */
#define SEG_SEED_KMAX 16          /* most run-in pixels there (SegParams::seed_kin) */
#define SEG_EH 512                /* slots of a segment's entry hash (per channel); a key lives in the SEG_EHW slots from its bucket's first */
#define SEG_EHW 8
#define SEG_EH_WORDS (SEG_EH + SEG_EHW - 4)
#define SEG_EH_EMPTY 0xFFFFFFFFu

/* what a row attempt decided (seg_ctl_body) */
enum { SEG_K_INIT = 0, SEG_K_RESTART, SEG_K_RETRY, SEG_K_COMMIT, SEG_K_ABORT, SEG_K_FINISHED };

/* ---- per-batch constants (strength, bleed): built on the host by seg_build_params ------------------------------------- */
struct SegParams {
    int32_t strength, bleed;
    int32_t ns;                    /* number of chain states */
    int32_t dmax, cmax, tmax;      /* |delta| <= dmax, |cn| <= cmax, |th| <= tmax */
    int32_t keyn;                  /* entries of keylut */
    int32_t engine_flags;          /* bits 8..: debugging aid, 1 + the candidate that wins every row */
    uint32_t lut_a[512];           /* [diff+256] -> rem (int16) | thr << 16      of the Sierra split (optimize_state.c:445-467) */
    uint32_t lut_b[512];           /* [diff+256] -> t | f << 8 | v << 16 | h << 24 (int8 each): the next-rows terms          */
    uint32_t st_pack[SEG_NS_MAX];  /* state i -> (delta+128) | (cn+128) << 8 | (th+128) << 16 */
    int32_t nsp;                   /* ns rounded up to a multiple of 64: stride of a segment's entry map */
    uint16_t keylut[SEG_KEYLUT_MAX]; /* ((delta+dmax) * (2cmax+1) + cn+cmax) * (2tmax+1) + th+tmax -> state or SEG_INVALID */
    /* filters whose prediction ignores the left pixel (none, up): the state is (cn, th) alone */
    uint8_t rt_max[256];           /* [D] -> max |rem(d)| + max |thr(d)| over |d| <= D (capped at 255) */
    int32_t ns_small, small_ok;
    uint32_t idx0_big, idx0_small; /* entry index of the state a fresh row starts with (nothing carried, boundary record all zero) */
    uint32_t st_small[SEG_NSS];    /* state i -> (cn+128) | (th+128) << 8 */
    uint16_t keylut_small[SEG_KEYS_MAX];   /* (cn+cmax) * (2tmax+1) + th+tmax -> state or SEG_INVALID */
    /* seeded enumeration (more chain states than SEG_NS_MAX) */
    int32_t seeded;                /* 1: segments are enumerated from seeds with a run-in (seg_enum_seeded_body), entry states are looked up by value */
    int32_t kin;                   /* run-in pixels */
    int32_t nseed_small;           /* seeds of none / up */
    uint32_t seed_small[SEG_NSP];  /* (cn+128) | (th+128) << 8 */
    /* enumeration in UNITS (batches, exhaustive state sets only): 1 = every segment is enumerated on its own (one image: the shortest dependent path);
     * SEG_UNIT = a run of SEG_UNIT segments is enumerated as one (seg_enum_unit_body): from every state only at its first pixel, its distinct states
     * run on through all its segments.  Set by the launcher (seg_build_params leaves 1). */
    int32_t unit;
    /* control kernel: workgroups that share the build of one candidate's decision tables: SEG_TPARTS (one image: the build is on the critical path of
     * every row), 1 for batches (a quarter of the control workgroups: what a batch pays for is workgroups, not the length of one) */
    int32_t tparts;
    /* enumeration in units FROM SEEDS (round 6; batches, exhaustive state sets of at most 255 states; seg_enum_unit_body<.., SEEDS = true>): a unit is not started from
     * every state at its first pixel but from seed_n seeds seed_kin pixels IN FRONT of it -- the states in which nothing was carried into the boundary pixel, one per diff
     * of that pixel -- and what they have become at the unit's first pixel is its entry set; every other entry index maps to "none".  0: no such set (too many seeds). */
    int32_t seed_n, seed_kin;
    uint16_t seed_idx[SEG_SEED_LANES];   /* indices into the exhaustive state list */
};

/* ---- per-image control block, double buffered by attempt parity ---------------------------------------------------------- */
struct SegCtl {
    uint32_t y, s, status;
    uint32_t finished;               /* 0: rows left; 1: the last row is committed, its validation still out (seg_ctl_body decides OPTIMISTICALLY, see there); 2: finished for good */
    uint32_t retried, restarts_total, serial_rows, attempts, dropped_none;
    uint32_t magic;                  /* SEG_MAGIC once a control kernel has written this block: the attempt that finds none behind it is the image's first (the launcher resets the word per batch) */
    uint32_t none_eager;             /* rows for which candidate none is run straight away (its bound did not rule it out lately) */
    uint32_t ignore;                 /* bit f: the decision this block came from does not depend on candidate f's row of the attempt before being valid (none, ruled out by its bound) */
    uint32_t active[SEG_NFILT];      /* the candidate still has unvalidated pixels (or sums) to produce in this attempt */
    uint32_t start_x[SEG_NFILT];     /* pixels [0, start_x) of the candidate row are validated */
    uint32_t state[SEG_NFILT][4];    /* chain state in front of start_x: left | (cn+128) << 8 | (th+128) << 16 */
    uint32_t restarts[SEG_NFILT];    /* epochs of this candidate in this row */
    uint64_t cost[SEG_NFILT];        /* final row cost of a finished candidate (~0 = rejected) */
};

/* the fields of the control block a workgroup branches on, in one burst of loads (read one by one where they are needed, each
 * `if` waits for its own round trip to device memory) */
struct SegCtlView { uint32_t finished, y, s, active, start_x; };
struct SegJob;
PLS_HD SegCtlView seg_ctl_view(const SegJob &j, int k, int f);

struct SegAcc {
    uint64_t derr[SEG_NFILT];           /* derivative error of the candidate row (seg_row_sums) */
    uint32_t hs[SEG_NFILT][SEG_NFILT];  /* sums of libpng's heuristic over the candidate row */
    uint32_t fail[SEG_NFILT];        /* smallest failing decision index x*4+c, or SEG_NOFAIL */
    uint32_t failmask;               /* bit f: candidate f's row failed validation (set by the validation workgroups, one launch behind the attempt) */
    uint32_t lb_valid;               /* workgroups that contributed to none_lb (must reach ngrp) */
    uint64_t none_lb;                /* lower bound of candidate none's row cost (seg_row_sums) */
};

static_assert(sizeof(SegCtl) / 4 <= 128 && sizeof(SegAcc) / 4 <= 128, "the control kernel copies both with 128 lanes each");
static_assert((SEG_NFILT + 1) * 256 + (sizeof(SegCtl) + 7) / 8 * 2 + (sizeof(SegAcc) + 7) / 8 * 2 + 56 <= 4 * SEG_TN, "they live in the table staging area, below the classes (and the decision behind them)");

/* What the workgroups of an attempt branch on, kept INSIDE the job record (three deep, like the control blocks): the record is the first thing
 * every workgroup loads, so these fields arrive with it -- one round trip to device memory less at the head of every kernel than reading them
 * from the control block behind the record's pointer.  Written by the control workgroups (and the fail masks by the validation) next to the
 * control block's own fields.  The kernels build a workgroup's SegCtlView from the record IN MEMORY (pl_seg.hip:seg_view_of: scalar loads whose
 * addresses need nothing but the kernel's arguments, in flight together with the record itself) and hand it to the body; the CPU harness calls
 * seg_ctl_view. */
struct SegViewRec { uint32_t finished, y, s, ignore, magic, pad_; uint32_t active[SEG_NFILT], start_x[SEG_NFILT]; };
struct SegJob {
    SEG_AS_GLB uint32_t *img;            /* slots image (pl_device.h) */
    SEG_AS_GLB uint8_t *row_filters;     /* or null */
    SEG_AS_GLB uint8_t *row_ids;
    uint32_t W, H;
    uint32_t bpp;             /* resolved by the launcher kernel (seg_resolve) / the host harness */
    uint32_t job_index;       /* which image of the batch (a mixed batch gives this engine a subset) */
    const SEG_AS_GLB uint32_t *orig_rank;/* [5][256] */
    SEG_AS_GLB uint32_t *cand;           /* [5][W][4]: byte | (diff16 & 0xffff) << 8 | bin << 24 */
    SEG_AS_GLB uint32_t *err0, *err1;    /* [2][W][2]: 4 x int16; by the parity of the row they belong to (seg_e0 / seg_e1) */
    SEG_AS_GLB uint32_t *rowcopy;        /* [3][W]: the ORIGINAL pixels of rows y-1, y, y+1 (row r in copy r % 3): the image row itself is committed in place while
                                            the validation of that very row is still running, and a row attempt that is repeated wants its originals back */
    SEG_AS_GLB uint32_t *final_hist;     /* [256] */
    SEG_AS_GLB int32_t *result;          /* [64] */
    SEG_AS_GLB uint32_t *progress;       /* or null: host-visible word that receives the number of finished rows (-v display) */
    SEG_AS_GLB uint32_t *done_counter;   /* or null: host-visible word, +1 when this image is finished (the host stops enqueueing attempts) */
    SEG_AS_GLB uint32_t *attempt_word;   /* or null: host-visible word that receives the number of the attempt being started (launch throttle) */
    SEG_AS_GLB uint32_t *break_word;     /* or null: host-visible word of the image's launch group, +1 for every row the chain kernel breaks off (the launch thread takes a group whose rows keep breaking
                                            off the start from seeds: pl_host.hip:seg_worker_main) */
    SEG_AS_GLB SegCtl *ctl;              /* [3]: by attempt % 3, like base, H0, acc */
    SEG_AS_GLB uint32_t *base;           /* [3][5][256] bumps of the validated prefix [0, start_x) */
    SEG_AS_GLB uint32_t *H0;             /* [3][256] committed histogram */
    SEG_AS_GLB SegAcc *acc;              /* [3] */
    SEG_AS_GLB uint32_t *tables;         /* [5][SEG_TBL_WORDS] */
    SEG_AS_GLB uint16_t *maps;           /* [5][nseg][4][nsp]: entry index of a segment -> dense id of its state after the dedupe (0xffff: none) */
    SEG_AS_GLB uint32_t *ehash;          /* (seeded) [5][nseg][4][SEG_EH_WORDS]: entry hash of a segment: key << 8 | dense id of the entry state, or SEG_EH_EMPTY */
    SEG_AS_GLB uint16_t *rout;           /* [5][nseg][4][SEG_NSP]: dense id -> exit index of the segment (0xffff: left what the tables cover) */
    SEG_AS_GLB uint32_t *rst;            /* [5][nseg][4][SEG_NSP]: dense id -> exit state of the segment, packed (0xffffffff: none) */
    SEG_AS_GLB uint32_t *rck;            /* [5][nseg][4][SEG_NSP][SEG_PARTS-1]: dense id -> state in front of part 1, 2, .. of the segment (0xffffffff: none) */
    SEG_AS_GLB uint16_t *dnout;          /* [5][nseg][4]: the dense id the row's true path has in every segment (chain kernel; 0xffff: unknown) */
    SEG_AS_GLB uint32_t *dcnt;           /* [5][nseg][4]: distinct states of the segment */
    SEG_AS_GLB uint32_t *entry;          /* [5][nseg][4] */
    SEG_AS_GLB uint16_t *segcnt;         /* [5][nseg][256] */
    SEG_AS_GLB uint32_t *grpcnt;         /* [5][ngrp][256] */
    SEG_AS_GLB uint32_t *grpleft;        /* [5][ngrp]: the new bytes (one per channel) the row sums of a group took for the pixel in front of it, when that pixel was another workgroup's (seg_row_sums; checked by the validation) */
    SEG_AS_GLB uint32_t *firstidx;       /* [5][4][2]: exit index of the epoch's first (partial) segment | packed state when it has none */
    SEG_AS_GLB int32_t *rowmm;           /* [2][2]: max and min of orig + incoming error over the row (seg_extremes_body, the chain launch's spare workgroup; by row parity: seg_rowmm) */
    uint32_t nseg, ngrp;
    SEG_AS_GLB SegJob *self;             /* where this record lives (device memory): the writers of the fields below */
    SegViewRec v[3];                     /* by attempt % 3 */
    uint32_t vfail[3];                   /* bit f: candidate f's row of that attempt failed validation (= SegAcc::failmask) */
    uint32_t nbreak;                     /* rows the chain kernel broke off because a unit's entry state was in no enumerated set (exhaustive state sets; seg_unit_from_seeds reads it) */
};

/* what belongs to row y (see SegJob) */
PLS_HD SEG_AS_GLB uint32_t *seg_row_orig(const SegJob &j, uint32_t y) { return j.rowcopy + (size_t)(y % 3u) * j.W; }
PLS_HD SEG_AS_GLB uint32_t *seg_e0(const SegJob &j, uint32_t y) { return j.err0 + (size_t)(y & 1u) * 2u * j.W; }
PLS_HD SEG_AS_GLB uint32_t *seg_e1(const SegJob &j, uint32_t y) { return j.err1 + (size_t)(y & 1u) * 2u * j.W; }
PLS_HD SEG_AS_GLB int32_t *seg_rowmm(const SegJob &j, uint32_t y) { return j.rowmm + (size_t)(y & 1u) * 2u; }
/* attempt a keeps its control block, sums, histogram and prefix bumps in copy a % 3: the one before, the one two before */
PLS_HD int seg_k_prev(int k) { return k == 0 ? 2 : k - 1; }
PLS_HD int seg_k_prev2(int k) { return k == 2 ? 0 : k + 1; }

/* The fields of attempt k's control block that a workgroup of that attempt branches on.  `finished` also stands for "nothing to do":
 * the attempt is VOID -- a candidate row of the attempt before it failed validation after that row's decision had been taken (the
 * validation runs one launch behind, seg_ctl_body), and the decision depended on it -- or there is no such attempt (the launch in front
 * of an image's first). */
PLS_HD SegCtlView seg_ctl_view(const SegJob &j, int k, int f)
{
    SegCtlView v;
    const SegViewRec &V = j.v[k];
    v.y = V.y; v.s = V.s; v.active = V.active[f]; v.start_x = V.start_x[f];
    v.finished = (V.finished != 0u || V.magic != SEG_MAGIC || (j.vfail[seg_k_prev(k)] & ~V.ignore) != 0u) ? 1u : 0u;
    return v;
}

/* ---- small pure helpers -------------------------------------------------------------------------------------------------- */
PLS_HD int seg_sext8(int v) { return (int)(int8_t)(uint8_t)(v & 0xff); }
PLS_HD int seg_sext16(int v) { return (int)(int16_t)(uint16_t)(v & 0xffff); }
PLS_HD int seg_min(int a, int b) { return a < b ? a : b; }
PLS_HD int seg_max(int a, int b) { return a > b ? a : b; }
PLS_HD int seg_abs(int a) { return a < 0 ? -a : a; }
PLS_HD int seg_paeth(int above, int diag, int left)
{
    const int p = above - diag, pd = left - diag;
    const int pl = seg_abs(p), pa = seg_abs(pd), pg = seg_abs(p + pd);
    return (pl <= pa && pl <= pg) ? left : (pa <= pg ? above : diag);
}
PLS_HD int seg_predict(int f, int above, int diag, int left)
{
    switch (f) {
    case 1: return left;
    case 2: return above;
    case 3: return (above + left) >> 1;
    case 4: return seg_paeth(above, diag, left);
    default: return 0;
    }
}
/* which error plane a channel uses / which channel feeds a plane (color_delta.c:4-41, optimize_state.c:167-171) */
PLS_HD int seg_plane_of_channel(uint32_t bpp, int c) { return (bpp == 2 && c == 1) ? 3 : c; }
PLS_HD int seg_channel_of_plane(uint32_t bpp, int p)
{
    if (bpp == 2) return p == 0 ? 0 : (p == 3 ? 1 : -1);
    return p < (int)bpp ? p : -1;
}
PLS_HD int seg_err_plane(const uint32_t *e2, int p) { return seg_sext16((int)(p < 2 ? (e2[0] >> (16 * p)) : (e2[1] >> (16 * (p - 2))))); }

/* Sierra split (optimize_state.c:397-401,445-467): truncating divisions */
struct SegSplit { int t, h, f, v, rem; };
PLS_HD SegSplit seg_split_slow(int diff16, int bleed)
{
    SegSplit s;
    int d = diff16 / bleed;
    s.t = d / 16; d -= 4 * s.t;
    s.h = d / 8; d -= 2 * s.h;
    s.f = (d * 2) / 9; d -= 2 * s.f;
    s.v = d / 2; d -= s.v;
    s.rem = d;
    return s;
}
PLS_HD void seg_rem_thr(const uint32_t *lut_a, int bleed, int diff, int &rem, int &thr)
{
    if (diff >= -256 && diff <= 255) { const uint32_t e = lut_a[diff + 256]; rem = seg_sext16((int)e); thr = (int)e >> 16; }
    else { const SegSplit s = seg_split_slow(diff, bleed); rem = s.rem; thr = s.h; }
}
PLS_HD uint32_t seg_terms(const uint32_t *lut_b, int bleed, int diff)
{
    if (diff >= -256 && diff <= 255) return lut_b[diff + 256];
    const SegSplit s = seg_split_slow(diff, bleed);
    return ((uint32_t)s.t & 255u) | (((uint32_t)s.f & 255u) << 8) | (((uint32_t)s.v & 255u) << 16) | ((uint32_t)s.h << 24);
}

PLS_HD uint32_t seg_terms_lds(seg_lds_u32 lut_b, int bleed, int diff)
{
    if (diff >= -256 && diff <= 255) return lut_b[diff + 256];
    const SegSplit s = seg_split_slow(diff, bleed);
    return ((uint32_t)s.t & 255u) | (((uint32_t)s.f & 255u) << 8) | (((uint32_t)s.v & 255u) << 16) | ((uint32_t)s.h << 24);
}

/* candidate word: byte | diff16 << 8 | bin << 24 */
PLS_HD uint32_t seg_cand_pack(int back, int diff, int bin) { return (uint32_t)(back & 255) | (((uint32_t)diff & 0xffffu) << 8) | ((uint32_t)(bin & 255) << 24); }
PLS_HD int seg_cand_byte(uint32_t w) { return (int)(w & 255u); }
PLS_HD int seg_cand_diff(uint32_t w) { return seg_sext16((int)(w >> 8)); }
PLS_HD int seg_cand_bin(uint32_t w) { return (int)(w >> 24); }

struct alignas(16) SegVec16 { uint32_t a, b, c, d; };

/* chain state */
struct SegState { int left, cn, th; };
PLS_HD uint32_t seg_state_pack(const SegState &s) { return (uint32_t)(s.left & 255) | ((uint32_t)((s.cn + 32768) & 0xffff) << 8) | ((uint32_t)((s.th + 128) & 255) << 24); }
PLS_HD SegState seg_state_unpack(uint32_t w) { SegState s; s.left = (int)(w & 255u); s.cn = (int)((w >> 8) & 0xffffu) - 32768; s.th = (int)(w >> 24) - 128; return s; }

/* per pixel and channel data of the chain: orig | above << 8 | diag << 16 | tr << 24, e0 */
struct SegPix { uint32_t w; int e0; };
PLS_HD SegPix seg_pix_make(int orig, int above, int diag, int tr, int e0) { SegPix p; p.w = (uint32_t)orig | ((uint32_t)above << 8) | ((uint32_t)diag << 16) | ((uint32_t)tr << 24); p.e0 = e0; return p; }

/* pixel x, channel c of row y from the image / error row (slots layout).  row = original row y, nab = optimised row y-1 or null */
PLS_HD SegPix seg_pix_load(const uint32_t *row, const uint32_t *nab, const uint32_t *err0, uint32_t bpp, uint32_t x, int c)
{
    const uint32_t o = row[x];
    const uint32_t a = nab ? nab[x] : 0u, d = (nab && x) ? nab[x - 1] : 0u;
    const int tr = ((bpp & 1u) == 0u && (uint32_t)c == bpp - 1u && ((o >> (8u * (bpp - 1u))) & 255u) == 0u) ? 1 : 0;
    return seg_pix_make((int)((o >> (8 * c)) & 255u), (int)((a >> (8 * c)) & 255u), (int)((d >> (8 * c)) & 255u), tr,
                        seg_err_plane(err0 + 2 * (size_t)x, seg_plane_of_channel(bpp, c)));
}

/* strength geometry: s, q = s + 1 and (device) the float reciprocal that makes trunc(a / q) == (int)(a * rq) for |a| < 2^17
 * (exhaustively checked in tests/test_host_logic.py::test_float_reciprocal_division) */
struct SegGeo { int s, q, fmax; float rq; };
PLS_HD SegGeo seg_geo(int s)
{
    SegGeo g; g.s = s; g.q = s + 1;
    union { float f; uint32_t u; } r; r.f = 1.0f / (float)(s + 1); r.u += 2u; g.rq = r.f;
    g.fmax = 1 << 20;                                           /* (the tables cover every clamped band whatever filt is; a band that misses lo..hi altogether needs no lookup) */
    return g;
}
PLS_HD int seg_div_q(int a, const SegGeo &g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)((float)a * g.rq);
#else
    return a / g.q;
#endif
}

/* the four channel records of pixel x (zero records beyond the row / for unused channels): every load of the pixel is issued once */
/* the same in two steps -- the loads (registers), then the split into per-channel records -- so that a block can put all its
 * requests in front of all its stores */
struct SegPixRaw { uint32_t o, a, d, e0, e1; };
PLS_HD SegPixRaw seg_pix_raw_zero() { SegPixRaw r; r.o = r.a = r.d = r.e0 = r.e1 = 0u; return r; }
PLS_HD SegPixRaw seg_pix_fetch(const uint32_t *row, const uint32_t *nab, const uint32_t *err0, uint32_t x, uint32_t W)
{
    SegPixRaw r = seg_pix_raw_zero();
    if (x < W) {
        r.o = row[x];
        if (nab) { r.a = nab[x]; r.d = x ? nab[x - 1] : 0u; }
        r.e0 = err0[2 * (size_t)x]; r.e1 = err0[2 * (size_t)x + 1];
    }
    return r;
}
PLS_HD void seg_pix_split4(SegPix *dst, const SegPixRaw &r, uint32_t bpp, uint32_t x, uint32_t W)
{
    const uint32_t e[2] = { r.e0, r.e1 };
    const bool alpha0 = x < W && (bpp & 1u) == 0u && ((r.o >> (8u * (bpp - 1u))) & 255u) == 0u;
    for (int c = 0; c < 4; c++) {
        if ((uint32_t)c < bpp && x < W)
            dst[c] = seg_pix_make((int)((r.o >> (8 * c)) & 255u), (int)((r.a >> (8 * c)) & 255u), (int)((r.d >> (8 * c)) & 255u), (alpha0 && (uint32_t)c == bpp - 1u) ? 1 : 0,
                                  seg_err_plane(e, seg_plane_of_channel(bpp, c)));
        else dst[c] = seg_pix_make(0, 0, 0, 0, 0);
    }
}
PLS_HD void seg_pix_load4(SegPix *dst, const uint32_t *row, const uint32_t *nab, const uint32_t *err0, uint32_t bpp, uint32_t x, uint32_t W)
{
    uint32_t o = 0, a = 0, d = 0, e[2] = { 0, 0 };
    if (x < W) {
        o = row[x];
        if (nab) { a = nab[x]; d = x ? nab[x - 1] : 0u; }
        e[0] = err0[2 * (size_t)x]; e[1] = err0[2 * (size_t)x + 1];
    }
    const bool alpha0 = x < W && (bpp & 1u) == 0u && ((o >> (8u * (bpp - 1u))) & 255u) == 0u;
    for (int c = 0; c < 4; c++) {
        if ((uint32_t)c < bpp && x < W)
            dst[c] = seg_pix_make((int)((o >> (8 * c)) & 255u), (int)((a >> (8 * c)) & 255u), (int)((d >> (8 * c)) & 255u), (alpha0 && (uint32_t)c == bpp - 1u) ? 1 : 0,
                                  seg_err_plane(e, seg_plane_of_channel(bpp, c)));
        else dst[c] = seg_pix_make(0, 0, 0, 0, 0);
    }
}

/* band geometry of one lookup (optimize_state.c:186-210): clamped candidate range [v0, v1] (single value when v0 == v1) */
struct SegBand { int v0, v1, bandlo, neg, t; };
PLS_HD SegBand seg_band(int filt, int lo, const SegGeo &g)
{
    const int s = g.s, q = g.q;
    SegBand b;
    b.neg = filt < 0;
    const int af = b.neg ? -filt : filt;
    const int t = seg_div_q(af, g);
    b.t = t;
    b.bandlo = b.neg ? -(t * q) - s : t * q;
    const int bandhi = b.bandlo + s, hi = lo + 255;
    b.v0 = seg_max(b.bandlo, lo);
    b.v1 = seg_min(bandhi, hi);
    if (b.v0 > b.v1) { const int v = bandhi < lo ? lo : hi; b.v0 = b.v1 = v; }
    return b;
}

/* the reference's choice inside [v0, v1] (optimize_state.c:212-244) = lexicographic arg-max of (H[v], O[v], v == osym, -v), by scanning.
 * H: symbol frequencies (256), extra: added on top (may be null), rank: order/equality preserving rank of original_frequency */
PLS_HD int seg_argmax_scan(const uint32_t *H, const uint32_t *extra, const uint32_t *rank, int v0, int v1, int osym)
{
    int best = v0;
    uint32_t bh = H[v0 & 255] + (extra ? extra[v0 & 255] : 0u), br = rank[v0 & 255];
    int bflag = v0 == osym;
    for (int v = v0 + 1; v <= v1; v++) {
        const uint32_t h = H[v & 255] + (extra ? extra[v & 255] : 0u), r = rank[v & 255];
        const int fl = v == osym;
        const bool better = h != bh ? h > bh : (r != br ? r > br : fl > bflag);
        if (better) { best = v; bh = h; br = r; bflag = fl; }
    }
    return best;
}

/* One step of one channel's chain against a frozen histogram, decision by scanning (replay, chain walk, exact pixel).
 * Returns the candidate word; st becomes the state in front of the next pixel. */
PLS_HD uint32_t seg_step_scan(int f, const SegPix &p, SegState &st, const uint32_t *H, const uint32_t *extra, const uint32_t *rank,
                              const SegGeo &g, const uint32_t *lut_a, int bleed)
{
    const int orig = (int)(p.w & 255u), above = (int)((p.w >> 8) & 255u), diag = (int)((p.w >> 16) & 255u);
    const int pred = seg_predict(f, above, diag, st.left);
    int back, diff, bin;
    if (p.w >> 24) { back = 0; diff = 0; bin = (0 - pred) & 255; }             /* optimize_state.c:158-164 */
    else {
        const int osym = seg_sext8(orig - pred), lo = osym - orig;
        const int filt = osym + seg_sext16(p.e0 + st.cn);
        const SegBand b = seg_band(filt, lo, g);
        const int v = seg_argmax_scan(H, extra, rank, b.v0, b.v1, osym);
        back = v - lo; diff = seg_sext16(filt - v); bin = v & 255;
    }
    int rem, thr;
    seg_rem_thr(lut_a, bleed, diff, rem, thr);
    st.left = back; st.cn = rem + st.th; st.th = thr;
    return seg_cand_pack(back, diff, bin);
}

/* ---- decision tables of one candidate filter (enumeration only) --------------------------------------------------------------
 * pre[sgn][v + SEG_TOFF] = leader of [bandlo(v), v], suf[sgn][v + SEG_TOFF] = leader of [v, bandhi(v)] in the band system of that
 * sign (sgn 0: filt >= 0, bands [tq, tq+s]; sgn 1: filt < 0, bands [-tq-s, -tq]); leader = arg-max of (H, rank, -v).
 * entry = (L + 1024) | cls[L & 255] << 16, cls = dense class of (H, rank): equal class <=> equal (H, rank), which is what decides
 * whether the original symbol takes the place of the leader (optimize_state.c:236-243).  Layout: pre[0] pre[1] suf[0] suf[1] cls. */
/* shared-memory pointers: naming the address space makes every access a ds_* instruction on the device (a generic pointer that
 * the compiler cannot trace back to LDS becomes a FLAT access: slower, and it ties LDS waits to outstanding global stores) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) const uint32_t *seg_lds_cu32;
typedef __attribute__((address_space(3))) const uint8_t *seg_lds_cu8;
#define SEG_LDS_CU32(p) ((seg_lds_cu32)(p))
#define SEG_LDS_CU8(p) ((seg_lds_cu8)(p))
#else
typedef const uint32_t *seg_lds_cu32;
typedef const uint8_t *seg_lds_cu8;
#define SEG_LDS_CU32(p) ((seg_lds_cu32)(p))
#define SEG_LDS_CU8(p) ((seg_lds_cu8)(p))
#endif

template <int F> PLS_HD int seg_predict_t(int above, int diag, int left)
{
    if (F == 1) return left;
    if (F == 2) return above;
    if (F == 3) return (above + left) >> 1;
    if (F == 4) {
        const int p = above - diag, pd = left - diag;
        const int pl = seg_abs(p), pa = seg_abs(pd), pg = seg_abs(p + pd);
        return (pl <= pa && pl <= pg) ? left : (pa <= pg ? above : diag);
    }
    return 0;
}

/* One step of one channel's chain against the frozen decision tables, without a branch: the hot loop of the enumeration, the replay
 * and the chain walk.  tw = the candidate's tables (pre[2] | suf[2] | cls) in shared memory, lut = the split table.
 * bad accumulates "this lane left what the tables cover" (band beyond +-SEG_TOFF, |diff| > 255): its results are void then and the
 * caller falls back (enumeration: the map entry is SEG_INVALID; replay / walk: seg_step_scan).  Returns the candidate word. */
/* 24-bit multiply (full rate on the device; the 32-bit v_mul_lo_u32 is quarter rate) */
PLS_HD uint32_t seg_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
PLS_HD int seg_mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
PLS_HD bool seg_bad(int acc) { return ((uint32_t)acc >> 11) != 0u; }      /* what seg_step_fast accumulates: did any step leave the split table? */
template <int F, bool TRX>
PLS_HD uint32_t seg_step_fast(const SegPix &p, SegState &st, int &bad, seg_lds_cu32 tw, seg_lds_cu8 cls, const SegGeo &g, seg_lds_cu32 lut)
{
    /* TRX: the range holds a fully transparent pixel (optimize_state.c:158-164): only then the forced-alpha selects are compiled in.
     * bad accumulates (OR) the split table's offset of every step: seg_bad() tells whether one of them lay outside the table */
    const int orig = (int)(p.w & 255u), above = (int)((p.w >> 8) & 255u), diag = (int)((p.w >> 16) & 255u);
    const int pred = seg_predict_t<F>(above, diag, st.left);
    const int osym = seg_sext8(orig - pred), lo = osym - orig, hi = lo + 255;
    const int filt = osym + seg_sext16(p.e0 + st.cn);
    const int sgn = filt >> 31;                                 /* -1 for filt < 0 (arithmetic shift): selects the sign's tables by masking */
    /* the band of filt (optimize_state.c:186-193): [tq, tq + s] with tq = trunc(filt / q) * q for filt >= 0, [tq - s, tq] for filt < 0 -- the
     * truncating division SIGNED (the float reciprocal is exact for |filt| < 2^17, and symmetric), so that no |filt|, no negation and no select
     * lie on the step's dependent path */
    const int tq = seg_mul24(seg_div_q(filt, g), g.q);
    const int bandlo = tq - (sgn & g.s), bandhi = bandlo + g.s;
    const int v0 = seg_max(bandlo, lo), v1 = seg_min(bandhi, hi);
    const bool degen = v0 > v1;                                /* the whole band lies outside [lo, hi]: the clamp leaves lo (band below: v0 == lo) or hi */
    const int vd = seg_min(v0, hi);
    const bool usesuf = v0 > bandlo;
    int key = usesuf ? v0 : v1;
#if !defined(__HIP_DEVICE_COMPILE__)
    key = seg_min(seg_max(key, -SEG_TOFF), SEG_TOFF - 1);      /* (only a degenerate band can point beyond the tables, and its entry is not used: on the device the read
                                                                   beyond the workgroup's shared memory returns zero -- one instruction less per step) */
#endif
    const uint32_t e = tw[(usesuf ? 2 * SEG_TN : 0) + (sgn & SEG_TN) + key + SEG_TOFF];
    const uint32_t c_os = (uint32_t)cls[(osym & 255) | (sgn & 256)];
    const int L = (int)(e & 0xffffu) - 1024;
    const bool tie = osym >= v0 && osym <= v1 && c_os == ((e >> 16) & 255u);
    int v = tie ? osym : L;
    v = degen ? vd : v;
    int back = v - lo, d32 = filt - v, bin = v & 255;
    if (TRX) {
        const bool tr = (p.w >> 24) != 0;
        back = tr ? 0 : back; d32 = tr ? 0 : d32; bin = tr ? ((0 - pred) & 255) : bin;
    }
    /* the split table covers diff in [-256, 255]: its byte offset is (4 * diff + 1024) & 0x7fc, and every bit of 4 * diff + 1024 above that, in any
     * step, says that the lane left the table (seg_bad): one OR per step instead of an absolute value, a subtraction and a maximum */
    const int t4 = (int)((uint32_t)d32 << 2) + 1024;
    bad |= t4;
    const uint32_t le = *(seg_lds_cu32)((seg_lds_cu8)lut + (t4 & 0x7fc));
    st.left = back; st.cn = seg_sext16((int)le) + st.th; st.th = (int)le >> 16;
    const int diff = seg_sext16(d32);
    return seg_cand_pack(back, diff, bin);
}

/* state <-> index, relative to the data of the boundary pixel b (the last pixel in front of the segment):
 * left = orig_b + e0_b + delta (0 for a forced transparent alpha, whose delta is 0 by definition) */
PLS_HD bool seg_state_decode(const SegParams &P, int i, const SegPix &b, SegState &st)
{
    if (i >= P.ns) return false;
    const uint32_t w = P.st_pack[i];
    const int delta = (int)(w & 255u) - 128;
    st.cn = (int)((w >> 8) & 255u) - 128;
    st.th = (int)((w >> 16) & 255u) - 128;
    if (b.w >> 24) { if (delta != 0) return false; st.left = 0; return true; }
    st.left = (int)(b.w & 255u) + b.e0 + delta;
    return st.left >= 0 && st.left <= 255;
}
PLS_HD uint32_t seg_state_encode(const SegParams &P, const SegPix &b, const SegState &st)
{
    const int delta = (b.w >> 24) ? 0 : st.left - ((int)(b.w & 255u) + b.e0);
    if ((b.w >> 24) && st.left != 0) return SEG_INVALID;
    if (delta < -P.dmax || delta > P.dmax || st.cn < -P.cmax || st.cn > P.cmax || st.th < -P.tmax || st.th > P.tmax) return SEG_INVALID;
    const int key = ((delta + P.dmax) * (2 * P.cmax + 1) + st.cn + P.cmax) * (2 * P.tmax + 1) + st.th + P.tmax;
    return key < P.keyn ? (uint32_t)P.keylut[key] : SEG_INVALID;
}

/* none / up: the state is (cn, th) */
PLS_HD bool seg_is_small(const SegParams &P, int f) { return P.small_ok && (f == 0 || f == 2); }
/* segments per enumeration unit of candidate f (SegParams::unit).  (none / up with their small state set segment by segment -- short workgroups between the
 * long ones -- was measured: 106.7 against 100.5 us for 32 frames of 1080p, and their chains then compose three times as many tables: every candidate in units) */
PLS_HD uint32_t seg_unit_of(const SegParams &P, int f) { (void)f; return (P.unit > 1 && !P.seeded) ? (uint32_t)P.unit : 1u; }
PLS_HD bool seg_small_decode(const SegParams &P, int i, SegState &st)
{
    if (i >= P.ns_small) return false;
    const uint32_t w = P.st_small[i];
    st.left = 0; st.cn = (int)(w & 255u) - 128; st.th = (int)((w >> 8) & 255u) - 128;
    return true;
}
PLS_HD uint32_t seg_small_encode(const SegParams &P, const SegState &st)
{
    if (st.cn < -P.cmax || st.cn > P.cmax || st.th < -P.tmax || st.th > P.tmax) return SEG_INVALID;
    return (uint32_t)P.keylut_small[(st.cn + P.cmax) * (2 * P.tmax + 1) + st.th + P.tmax];
}
/* either kind, by filter */
PLS_HD bool seg_any_decode(const SegParams &P, int f, int i, const SegPix &b, SegState &st) { return seg_is_small(P, f) ? seg_small_decode(P, i, st) : seg_state_decode(P, i, b, st); }
PLS_HD uint32_t seg_any_encode(const SegParams &P, int f, const SegPix &b, const SegState &st) { return seg_is_small(P, f) ? seg_small_encode(P, st) : seg_state_encode(P, b, st); }

/* seeded enumeration: entry states are found BY VALUE.  Key of a chain state: left | (cn + 128) << 8 | (th + 128) << 16 (24 bits; ~0: the state
 * has none); a key lives in one of the SEG_EHW slots from the first slot of its bucket (128 buckets of 4 slots, windows overlap), so a lookup
 * is two aligned 16-byte loads.  Slot word in device memory: key << 8 | dense id. */
#define SEG_NOKEY 0xFFFFFFFFu
PLS_HD uint32_t seg_eh_key(int left, int cn, int th)
{
    if (cn < -128 || cn > 126 || th < -128 || th > 126) return SEG_NOKEY;
    return (uint32_t)(left & 255) | ((uint32_t)(cn + 128) << 8) | ((uint32_t)(th + 128) << 16);
}
PLS_HD uint32_t seg_eh_key_of_packed(uint32_t ps)
{
    if (ps == 0xFFFFFFFFu) return SEG_NOKEY;
    const SegState st = seg_state_unpack(ps);
    return seg_eh_key(st.left, st.cn, st.th);
}
PLS_HD SegState seg_eh_state(uint32_t key) { SegState st; st.left = (int)(key & 255u); st.cn = (int)((key >> 8) & 255u) - 128; st.th = (int)((key >> 16) & 255u) - 128; return st; }
PLS_HD uint32_t seg_eh_base(uint32_t key) { return ((key * 0x9E3779B1u) >> 25) * 4u; }
/* dense id of the state with this key among the window's eight slot words, or SEG_INVALID */
PLS_HD uint32_t seg_eh_match(uint32_t key, const SegVec16 &a, const SegVec16 &b)
{
    uint32_t id = SEG_INVALID;
    if (key == SEG_NOKEY) return id;
    id = (a.a >> 8) == key ? (a.a & 255u) : id; id = (a.b >> 8) == key ? (a.b & 255u) : id; id = (a.c >> 8) == key ? (a.c & 255u) : id; id = (a.d >> 8) == key ? (a.d & 255u) : id;
    id = (b.a >> 8) == key ? (b.a & 255u) : id; id = (b.b >> 8) == key ? (b.b & 255u) : id; id = (b.c >> 8) == key ? (b.c & 255u) : id; id = (b.d >> 8) == key ? (b.d & 255u) : id;
    return id;
}
/* (one lookup on its own: the chain's repair path, the host harness) eh: the table of one segment and channel */
PLS_HD uint32_t seg_eh_lookup(const SEG_AS_GLB uint32_t *eh, uint32_t key)
{
    if (key == SEG_NOKEY) return SEG_INVALID;
    const uint32_t base = seg_eh_base(key);
    uint32_t id = SEG_INVALID;
    for (int q = 0; q < SEG_EHW; q++) { const uint32_t w = eh[base + q]; if (w != SEG_EH_EMPTY && (w >> 8) == key) id = w & 255u; }
    return id;
}

/* ---- host: the constants of a (strength, bleed) pair ------------------------------------------------------------------------
 * seg_build_exhaustive: the whole chain-state set (delta, cn, th), when it has at most SEG_NS_MAX members (false otherwise);
 * seg_build_params: that, or -- for larger sets, or when force_seeded asks for it (tests) -- the constants of the seeded enumeration. */
inline bool seg_build_exhaustive(SegParams &P, int strength, int bleed)
{
    if (strength > 127) return false;
    const int tmax = P.tmax;
    if (P.dmax > 127 || P.cmax > 127) return false;
    P.keyn = (2 * P.dmax + 1) * (2 * P.cmax + 1) * (2 * P.tmax + 1);
    if (P.keyn > SEG_KEYLUT_MAX) return false;
    for (int k = 0; k < P.keyn; k++) P.keylut[k] = (uint16_t)SEG_INVALID;
    int ns = 0;
    /* every (diff of the boundary pixel, carry it received, thr of the pixel before it) gives one state */
    for (int diff = -strength; diff <= strength; diff++) {
        const SegSplit sp = seg_split_slow(diff, bleed);
        for (int carry = -P.cmax; carry <= P.cmax; carry++)
            for (int thp = -tmax; thp <= tmax; thp++) {
                const int delta = carry - diff, cn = sp.rem + thp, th = sp.h;
                if (seg_abs(delta) > P.dmax || seg_abs(cn) > P.cmax) continue;
                const int key = ((delta + P.dmax) * (2 * P.cmax + 1) + cn + P.cmax) * (2 * P.tmax + 1) + th + P.tmax;
                if (P.keylut[key] != (uint16_t)SEG_INVALID) continue;
                if (ns >= SEG_NS_MAX) return false;
                P.keylut[key] = (uint16_t)ns;
                P.st_pack[ns] = (uint32_t)(delta + 128) | ((uint32_t)(cn + 128) << 8) | ((uint32_t)(th + 128) << 16);
                ns++;
            }
    }
    P.ns = ns; P.nsp = (ns + 63) / 64 * 64;
    {
        const int kn = (2 * P.cmax + 1) * (2 * P.tmax + 1);
        int n2 = 0; bool ok = kn <= SEG_KEYS_MAX;
        for (int k = 0; k < SEG_KEYS_MAX; k++) P.keylut_small[k] = (uint16_t)SEG_INVALID;
        for (int diff = -strength; diff <= strength && ok; diff++) {
            const SegSplit sp = seg_split_slow(diff, bleed);
            for (int thp = -tmax; thp <= tmax; thp++) {
                const int cn = sp.rem + thp, th = sp.h;
                const int key = (cn + P.cmax) * (2 * P.tmax + 1) + th + P.tmax;
                if (P.keylut_small[key] != (uint16_t)SEG_INVALID) continue;
                if (n2 >= SEG_NSS) { ok = false; break; }
                P.keylut_small[key] = (uint16_t)n2;
                P.st_small[n2] = (uint32_t)(cn + 128) | ((uint32_t)(th + 128) << 8);
                n2++;
            }
        }
        P.ns_small = ok ? n2 : 0; P.small_ok = ok ? 1 : 0;
    }
    {
        /* the seeds of the unit enumeration from seeds: nothing carried into the boundary pixel (carry 0, thr of the pixel before 0), one state per diff of that pixel.
         * Measured on the reference's own trajectories (oracle/seed_study.c, profiles/r06_seed_study.txt): after a run-in of 8 pixels the true state at a boundary is among
         * what they have become in all but 0 .. 2e-3 of the boundaries of photographic content; flat regions hold fixed points they miss (the launcher falls back). */
        int n = 0; bool ok = ns <= 255;
        for (int delta = -P.dmax; delta <= P.dmax && ok; delta++) {
            /* one seed per left byte within reach: the diff of the boundary pixel that explains it with the least carried into that pixel (none for |delta| <= s) */
            const int diff = seg_max(-strength, seg_min(strength, -delta));
            const SegSplit sp = seg_split_slow(diff, bleed);
            const int cn = sp.rem, th = sp.h;
            if (seg_abs(delta) > P.dmax || seg_abs(cn) > P.cmax || seg_abs(th) > P.tmax) continue;
            const uint16_t idx = P.keylut[((delta + P.dmax) * (2 * P.cmax + 1) + cn + P.cmax) * (2 * P.tmax + 1) + th + P.tmax];
            if (idx == (uint16_t)SEG_INVALID) continue;
            bool dup = false;
            for (int q = 0; q < n; q++) dup |= P.seed_idx[q] == idx;
            if (dup) continue;
            if (n >= SEG_SEED_LANES) { ok = false; break; }
            P.seed_idx[n++] = idx;
        }
        P.seed_n = ok ? n : 0; P.seed_kin = 8;
    }
    {
        const SegState z = { 0, 0, 0 };
        const SegPix b0 = seg_pix_make(0, 0, 0, 0, 0);
        P.idx0_big = seg_state_encode(P, b0, z);
        P.idx0_small = P.small_ok ? seg_small_encode(P, z) : (uint32_t)SEG_INVALID;
    }
    return true;
}
inline bool seg_build_params(SegParams &P, int strength, int bleed, bool force_seeded = false)
{
    memset(&P, 0, sizeof P);
    if (strength < 0 || strength > 255 || bleed < 1 || bleed > 32767) return false;
    P.strength = strength; P.bleed = bleed; P.unit = 1; P.tparts = SEG_TPARTS;
    for (int d = -256; d <= 255; d++) {
        const SegSplit s = seg_split_slow(d, bleed);
        P.lut_a[d + 256] = ((uint32_t)s.rem & 0xffffu) | ((uint32_t)s.h << 16);
        P.lut_b[d + 256] = ((uint32_t)s.t & 255u) | (((uint32_t)s.f & 255u) << 8) | (((uint32_t)s.v & 255u) << 16) | ((uint32_t)s.h << 24);
    }
    {
        int rm_ = 0, tm_ = 0;
        for (int D = 0; D < 256; D++) {
            const SegSplit a = seg_split_slow(D, bleed), b = seg_split_slow(-D, bleed);
            rm_ = seg_max(rm_, seg_max(seg_abs(a.rem), seg_abs(b.rem))); tm_ = seg_max(tm_, seg_max(seg_abs(a.h), seg_abs(b.h)));
            P.rt_max[D] = (uint8_t)seg_min(255, rm_ + tm_);
        }
    }
    int rmax = 0, tmax = 0;
    for (int d = -strength; d <= strength; d++) { const SegSplit s = seg_split_slow(d, bleed); if (seg_abs(s.rem) > rmax) rmax = seg_abs(s.rem); if (seg_abs(s.h) > tmax) tmax = seg_abs(s.h); }
    P.cmax = rmax + tmax; P.tmax = tmax; P.dmax = strength + P.cmax;
    if (!force_seeded && seg_build_exhaustive(P, strength, bleed)) return true;
    /* seeded: the filters that look at the left pixel take their seeds from the data (seg_enum_seeded_body); none / up, whose state is
     * (cn, th): every cn, and th on the finest grid that keeps the set within the lanes */
    /* run-in: the carried terms must have contracted when the seeds reach the segment -- the larger they can be, the longer it takes.
     * Measured on 8192-pixel rows (a miss costs the chain kernel ~16 us, a run-in pixel ~0.4 us of every enumeration workgroup): s = 85 at
     * bleed 2 and s = 40 at bleed 1 (cmax 12) are fastest with 16 pixels, s = 85 at bleed 1 (cmax 23) with 24 (profiles/r04_seg_coverage.txt) */
    /* (round 5: since the run-in runs in two stages -- only the distinct states through most of it -- a longer run-in costs less, and at bleed 1, where the carried
     *  terms contract slowest, 20 pixels beat 16: s = 40 b = 1 55.8 -> 70.6 Mpixels/s on an 8192-pixel strip (three segments a row walked step by step before);
     *  s = 85 b = 2, the same bounds at bleed 2, stays fastest at 16: profiles/r05_seeded_runin.txt) */
    P.seeded = 1; P.kin = P.cmax <= 12 ? (bleed == 1 ? 20 : 16) : (P.cmax <= 24 ? 24 : SEG_KIN);
    P.ns = 0; P.nsp = 64; P.keyn = 0; P.ns_small = 0; P.small_ok = 0; P.idx0_big = P.idx0_small = (uint32_t)SEG_INVALID;
    if (P.cmax > 127 || tmax > 127 || 2 * P.cmax + 1 > SEG_NSP) return false;     /* (s = 255 at bleed 1: cmax 66, tmax 24) */
    int tstep = 1;
    while ((2 * P.cmax + 1) * (2 * (tmax / tstep) + 1) > SEG_NSP) tstep++;
    int n = 0;
    for (int cn = -P.cmax; cn <= P.cmax; cn++)
        for (int th = -(tmax / tstep) * tstep; th <= tmax; th += tstep) P.seed_small[n++] = (uint32_t)(cn + 128) | ((uint32_t)(th + 128) << 8);
    P.nseed_small = n;
    return true;
}

/* =========================================================================================================================
 * Kernel bodies.  smem: the workgroup's shared memory (device: dynamic LDS; host harness: a scratch buffer).
 * par = attempt & 1: which copy of the control block this attempt reads.
 * ========================================================================================================================= */

/* shared-memory budgets (bytes) */
#define SEG_SM_ENUM (SEG_TBL_WORDS * 4 + 2048 + SEG_SMALL_SEGS * SEG_L * 4 * 8 + 64 + 4 * 512 * 4 + 4 * 512 * 2 + 4 * SEG_NSP * 4 + 4 * SEG_NSP * 2 + SEG_THREADS * 2 + SEG_THREADS * 4 + 64)
/* what the enumeration kernel's bodies really carve out for NT threads (seg_enum_body is the largest: tables, pixels, split table, hash table,
 * dense ids, distinct states, exits, one slot and one key per lane): 35.5 KB at 512 threads = four workgroups per CU, 38.5 KB at 1024 */
#define SEG_SM_ENUM_NT(nt) (SEG_TBL_WORDS * 4 + (SEG_L + 1) * 4 * 8 + 2048 + 32 + 4 * SEG_HT * 4 + 4 * SEG_HT * 2 + 4 * SEG_NSP * 4 + 4 * SEG_NSP * 2 + (nt) * 2 + (nt) * 4 + 128)
#define SEG_SM_REPLAY (4096 + SEG_TBL_WORDS * 4 + SEG_GRP * SEG_L * 4 * 8 + SEG_GRP * 256 + SEG_GRP * SEG_PARTS * 4 * 8 + SEG_GRP * SEG_L * 16 + (SEG_GRP * SEG_L + 3) * 4 + 64)
#ifndef SEG_WATCH_OF
#define SEG_WATCH_OF(V) ((V) > 8 ? 2 : 8)   /* slots of exact prefix counts a validation workgroup has for the bins its undecided decisions name (a bin without a slot is counted by scanning: exact, slower, rare -- 0.01 % of the decisions are undecided at all); two for whole replay groups: with 51 KB instead of 64 a CU takes a third enumeration workgroup next to such a workgroup */
#endif
#define SEG_SM_POST_V(V) ((768 + ((V) + 1) * 256 + ((V) * SEG_L + 2) * 4 + 64 + 512 + 2 * ((V) * SEG_L + 2) + 2 * (V) * SEG_L + 32 + 64 + (V) * SEG_L + (V) * (SEG_L + 1) + SEG_WATCH_OF(V) * ((V) * (SEG_L + 1) + 8) + 2 * 20 * 4 + 512 + 64) * 4)   /* what seg_post_body carves out, in its order (SEG_WATCH_OF(V) slots, SEG_NBAND = 20 bands): 36.9 KB for half replay groups, 51.0 KB for whole ones (round 5: 49 KB before for half groups -- the launch that carries it was short of CUs with that much free next to the enumeration of another launch group) */
#define SEG_SM_POST SEG_SM_POST_V(SEG_VGRP)
#define SEG_SM_CTLVAL_V(V) (SEG_SM_CTL > SEG_SM_POST_V(V) ? SEG_SM_CTL : SEG_SM_POST_V(V))
#define SEG_SM_CTLVAL SEG_SM_CTLVAL_V(SEG_VGRP)   /* the first launch of an attempt carries control and validation workgroups */
#define SEG_SM_CTL ((8 + 512 + (sizeof(SegCtl) + 7) / 8 * 2 + (sizeof(SegAcc) + 7) / 8 * 2 + 56 + SEG_NFILT * (SEG_COMMIT_W + 4) * 4 + SEG_COMMIT_W * 4 + (SEG_NFILT + 1) * 256 + 16) * 4)   /* what a commit workgroup carves out (seg_ctl_commit: split table, control block + sums, decision, five tiles, err1 of both parities, spec): 33.8 KB; a candidate workgroup needs 4 x 256 + SEG_TBL_WORDS words */
static_assert((1024 + SEG_TBL_WORDS) * 4 <= SEG_SM_CTL, "a candidate workgroup of the control kernel (histograms + table staging) fits the commit workgroups' request");

/* run `n` steps of filter f from pixel record px[0] (stride pstride records per pixel); returns bad > 0 when the lane left the tables */
template <int F, bool TRX>
PLS_HD int seg_run_fast(const SegPix *px, int pstride, int n, SegState &st, seg_lds_cu32 tw, seg_lds_cu8 cls, const SegGeo &g, seg_lds_cu32 lut)
{
    int bad = 0;
    SegPix p = px[0];
    for (int k = 0; k < n; k++) {
        const SegPix pn = px[(k + 1 < n ? k + 1 : k) * pstride];     /* the next record is on its way while this step runs */
        (void)seg_step_fast<F, TRX>(p, st, bad, tw, cls, g, lut);
        p = pn;
    }
    return seg_bad(bad) ? 1 : 0;
}
template <bool TRX>
PLS_HD int seg_run_fast_t(int f, const SegPix *px, int pstride, int n, SegState &st, seg_lds_cu32 tw, seg_lds_cu8 cls, const SegGeo &g, seg_lds_cu32 lut)
{
    switch (f) {
    case 1: return seg_run_fast<1, TRX>(px, pstride, n, st, tw, cls, g, lut);
    case 2: return seg_run_fast<2, TRX>(px, pstride, n, st, tw, cls, g, lut);
    case 3: return seg_run_fast<3, TRX>(px, pstride, n, st, tw, cls, g, lut);
    case 4: return seg_run_fast<4, TRX>(px, pstride, n, st, tw, cls, g, lut);
    default: return seg_run_fast<0, TRX>(px, pstride, n, st, tw, cls, g, lut);
    }
}
/* trx: some pixel of the range is fully transparent (uniform over the workgroup) */
PLS_HD int seg_run_fast_f(int f, bool trx, const SegPix *px, int pstride, int n, SegState &st, seg_lds_cu32 tw, seg_lds_cu8 cls, const SegGeo &g, seg_lds_cu32 lut)
{
    return trx ? seg_run_fast_t<true>(f, px, pstride, n, st, tw, cls, g, lut) : seg_run_fast_t<false>(f, px, pstride, n, st, tw, cls, g, lut);
}

/* ---- ENUMERATE, filters that look at the left pixel: task (f, seg), SEG_THREADS lanes = 4 channels x SEG_NSP states --------
 * Trajectories merge quickly (the carried error terms contract within a few pixels; what stays apart are the ~2s+1 possible left
 * bytes): after SEG_K1 steps the lanes of a channel hold far fewer DISTINCT states than lanes.  They are deduplicated through a small
 * hash table in shared memory (exact: full keys are compared), the distinct ones are packed into the first lanes of the channel and only
 * those run the remaining steps -- whole waves fall idle -- then every lane picks up the result of its representative. */
#define SEG_K1 4                 /* steps before the dedupe when the state set comes in several chunks of SEG_NSP lanes */
#ifndef SEG_K1_ONE_CHUNK
#define SEG_K1_ONE_CHUNK 4       /* ... when it fits one chunk.  Measured on the headline frame (profiles/r04_enum_variants.txt), enumeration us for 1 / 2 / 3 / 4 / 6
                                    steps before the dedupe: one wave per channel behind it 20.6 (and the chain 15.9: too many distinct states) / 18.6 / 19.2 / 19.7 / 20.0;
                                    both channels packed into the first lanes (as shipped) - / 17.4 / 18.5 / 16.7 / 17.3 (5 steps: 16.9) -- after two steps the two
                                    channels' ~35 + 35 distinct states no longer fit one wave, after four their ~17 + 17 do */
#endif
PLS_HD int seg_k1(int ns) { return ns <= SEG_NSP ? SEG_K1_ONE_CHUNK : SEG_K1; }
#define SEG_HT 512
template <int NT>
PLS_HD void seg_enum_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int seg, int chalf, unsigned char *smem)
{
    if (cv.finished || cv.active != 1) return;
    const uint32_t W = j.W, bpp = j.bpp;
    constexpr int NCH = NT / SEG_NSP;                        /* channels of this workgroup: c0 .. c0 + NCH - 1 */
    const int c0 = chalf * NCH;
    if ((uint32_t)c0 >= bpp) return;
    const uint32_t x0 = (uint32_t)seg * SEG_L;
    if (x0 >= W) return;                                      /* (the last segment has no successor, but the replay wants its checkpoints) */
    if (cv.start_x && x0 <= cv.start_x) return;       /* an epoch that starts inside the row: its first (partial) segment is walked by
                                                                 seg_first_body.  A fresh row starts from the known state in front of pixel 0,
                                                                 which has an index like any other (boundary record = zeros): segment 0 is
                                                                 enumerated with the rest */
    uint32_t *tw = (uint32_t *)smem;
    SegPix *px = (SegPix *)(smem + SEG_TBL_WORDS * 4);        /* [(SEG_L + 1)][4]: slot 0 = boundary pixel x0-1 */
    uint32_t *lut = (uint32_t *)(px + (SEG_L + 1) * 4);       /* the split table */
    uint32_t *trflag = lut + 512;                             /* [0] some pixel of the segment is fully transparent, [1..4] distinct states per channel */
    uint32_t *ht = trflag + 8;                                /* [4][SEG_HT] hash table: packed state or ~0 */
    uint16_t *dense = (uint16_t *)(ht + 4 * SEG_HT);          /* [4][SEG_HT] slot -> rank among the distinct states */
    uint32_t *uniq = (uint32_t *)(dense + 4 * SEG_HT);        /* [4][SEG_NSP] the distinct states, packed */
    uint16_t *res = (uint16_t *)(uniq + 4 * SEG_NSP);         /* [4][SEG_NSP] exit index of each distinct state */
    uint16_t *lslot = res + 4 * SEG_NSP;                      /* [NT] the hash slot of every lane's state (or 0xffff) */
    const uint32_t y = cv.y;
    const SEG_AS_GLB uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    const bool span = SEG_EXPERIMENT_WG_SPAN && (P.engine_flags & 1) != 0 && (cv.y & 3u) == 0u;
    const bool prof = !SEG_EXPERIMENT_WG_SPAN && (P.engine_flags & 1) != 0;
    unsigned long long te[5] = { 0, 0, 0, 0, 0 };
    if (prof || span) te[0] = PLS_CLOCK();
    if (span && f == 1 && seg == 0 && chalf == 0) { PLS_THREADS(tid, NT) { if (tid == 0) PLS_ATOMIC_EXCH((uint32_t *)&j.result[47], (uint32_t)te[0]); } }   /* (the launch's first workgroup) */
    PLS_THREADS(tid, NT) {
        if (tid < 8) trflag[tid] = 0u;
        for (int i = tid; i < 4 * SEG_HT; i += NT) ht[i] = 0xffffffffu;
    }
    PLS_SYNC();
    PLS_THREADS(tid, NT) {
        /* requests first, stores behind them (one round trip) */
        constexpr int NTW = (SEG_TBL_WORDS + NT - 1) / NT;
        uint32_t vt[NTW], vl = 0;
        SegPixRaw vp = seg_pix_raw_zero();
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; vt[q] = i < SEG_TBL_WORDS ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (tid < 512) vl = P.lut_a[tid];
        if (tid < SEG_L + 1) vp = seg_pix_fetch(row, nab, e0g, x0 - 1 + (uint32_t)tid, W);
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
        if (tid < 512) lut[tid] = vl;
        if (tid < SEG_L + 1) {
            seg_pix_split4(px + tid * 4, vp, bpp, x0 - 1 + (uint32_t)tid, W);
            if ((bpp & 1u) == 0u && (px[tid * 4 + (bpp - 1u)].w >> 24)) PLS_ATOMIC_OR(trflag, 1u);
        }
    }
    PLS_SYNC();
    const bool trx = trflag[0] != 0u;
    if (prof) te[1] = PLS_CLOCK();
    /* -- the first SEG_K1 steps from every state, SEG_NSP states per channel at a time; neighbouring lanes mostly end in the same state,
     *    so only the first lane of a run of equal keys ("head") goes to the hash table with an atomic, the others look their key up
     *    afterwards.  Every entry index gets the DENSE id of its state: that is the segment's entry map -- */
    uint32_t *keys = (uint32_t *)(lslot + NT);       /* [NT] state key of every lane after SEG_K1 steps, ~0 = none */
    const int K1 = seg_k1(P.ns);
    SEG_AS_GLB uint16_t *dmap = j.maps + (((size_t)f * j.nseg + seg) * 4) * (size_t)P.nsp;
    for (int i0 = 0; i0 < P.ns; i0 += SEG_NSP) {
        PLS_THREADS(tid, NT) {
            const int lc = tid / SEG_NSP, c = c0 + lc, i = i0 + tid % SEG_NSP;
            uint32_t key = 0xffffffffu;
            if ((uint32_t)c < bpp && i < P.ns) {
                SegState st;
                if (seg_state_decode(P, i, px[c], st)) {
                    const int bad = seg_run_fast_f(f, trx, px + 4 + c, 4, K1, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                    if (!bad && st.cn >= -128 && st.cn <= 127) key = (uint32_t)(st.left & 255) | ((uint32_t)(st.cn & 255) << 8) | ((uint32_t)(st.th & 255) << 16);
                }
            }
            keys[tid] = key;
        }
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            const int c = tid / SEG_NSP, i = tid % SEG_NSP;          /* (local channel: tables of this workgroup) */
            const uint32_t key = keys[tid];
            if (key != 0xffffffffu && (i == 0 || keys[tid - 1] != key)) {
                uint32_t h = (key * 0x9E3779B1u) >> 23;                            /* 9 bits */
                for (int probe = 0; probe < SEG_HT; probe++) {
                    const uint32_t old = PLS_ATOMIC_CAS(&ht[c * SEG_HT + h], 0xffffffffu, key);
                    if (old == 0xffffffffu) {                                     /* the representative of a new state */
                        const uint32_t d = PLS_ATOMIC_ADD_RET(&trflag[1 + c], 1u);
                        dense[c * SEG_HT + h] = (uint16_t)(d < SEG_NSP ? d : 0xffffu);          /* more distinct states than lanes: the surplus has no id */
                        if (d < SEG_NSP) uniq[c * SEG_NSP + d] = key;
                        break;
                    }
                    if (old == key) break;
                    h = (h + 1) & (SEG_HT - 1);
                }
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            const int c = tid / SEG_NSP, gc = c0 + c, i = i0 + tid % SEG_NSP;
            const uint32_t key = keys[tid];
            uint32_t d = 0xffffu;
            if (key != 0xffffffffu) {
                uint32_t h = (key * 0x9E3779B1u) >> 23;
                for (int probe = 0; probe < SEG_HT; probe++) {
                    const uint32_t at = ht[c * SEG_HT + h];
                    if (at == key) { d = dense[c * SEG_HT + h]; break; }
                    if (at == 0xffffffffu) break;                                 /* (the table was full when its head came) */
                    h = (h + 1) & (SEG_HT - 1);
                }
            }
            if ((uint32_t)gc < bpp && i < P.ns) dmap[(size_t)gc * P.nsp + i] = (uint16_t)d;
        }
        PLS_SYNC();
    }
    if (prof) te[2] = PLS_CLOCK();
    /* -- the remaining steps, distinct states only (packed into the first lanes of each channel): dense id -> exit index -- */
    PLS_THREADS(tid, NT) {
        /* the distinct states of the workgroup's channels PACKED into its first lanes, channel behind channel: a dozen states per channel are one wave
         * for all of them, and what this phase costs is the instruction issue of the waves that have a live lane (the workgroups of a CU share its
         * four SIMDs: waves x steps is the currency, the other waves end here) */
        uint32_t Dc[NCH];
        for (int k = 0; k < NCH; k++) Dc[k] = trflag[1 + k] < SEG_NSP ? trflag[1 + k] : SEG_NSP;
        int lc = 0, i = tid;
        uint32_t D = Dc[0];
        PLS_UNROLL
        for (int k = 0; k + 1 < NCH; k++) if (lc == k && (uint32_t)i >= Dc[k]) { i -= (int)Dc[k]; lc = k + 1; D = Dc[k + 1]; }
        const int c = c0 + lc;
        if (tid < NCH && (uint32_t)(c0 + tid) < bpp) j.dcnt[((size_t)f * j.nseg + seg) * 4 + c0 + tid] = trflag[1 + tid] < SEG_NSP ? trflag[1 + tid] : SEG_NSP;
        if ((uint32_t)c < bpp && (uint32_t)i < D) {
            const uint32_t key = uniq[lc * SEG_NSP + i];
            SegState st;
            st.left = (int)(key & 255u); st.cn = seg_sext8((int)(key >> 8)); st.th = seg_sext8((int)(key >> 16));
            uint32_t out = SEG_INVALID;
            /* to the end of part 0, then part by part: the state at every cut is a checkpoint the replay starts a lane from */
            const size_t slot = (((size_t)f * j.nseg + seg) * 4 + c) * SEG_NSP + i;
            int bad = seg_run_fast_f(f, trx, px + (1 + K1) * 4 + c, 4, SEG_PL - K1, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
            for (int part = 1; part < SEG_PARTS; part++) {
                j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                bad |= seg_run_fast_f(f, trx, px + (1 + part * SEG_PL) * 4 + c, 4, SEG_PL, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
            }
            if (!bad) out = seg_state_encode(P, px[SEG_L * 4 + c], st);
            j.rout[slot] = (uint16_t)out;
            j.rst[slot] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
        }
        if (prof && tid == 0 && !SEG_EXPERIMENT_REPLAY_CLOCKS) {
            te[3] = PLS_CLOCK(); te[4] = te[3];
            for (int q = 0; q < 4; q++) { PLS_ATOMIC_MAX(&j.result[24 + q], (int32_t)(te[q + 1] - te[q])); PLS_ATOMIC_ADD((uint32_t *)&j.result[28 + q], (uint32_t)(te[q + 1] - te[q])); }
            PLS_ATOMIC_ADD((uint32_t *)&j.result[32], 1u);
            PLS_ATOMIC_ADD((uint32_t *)&j.result[33], trflag[1] + trflag[2] + trflag[3] + trflag[4]);
        }
    }
    if (span) {
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            if (tid == 0) {
                /* start and end of this workgroup against the start of the launch's first; slots of the phase clocks: "load" = start, "first steps" = end, "remaining" = duration */
                const uint32_t ref = PLS_ATOMIC_ADD_RET((uint32_t *)&j.result[47], 0u);
                const int32_t ds = (int32_t)((uint32_t)te[0] - ref), de = (int32_t)((uint32_t)PLS_CLOCK() - ref);
                if (ds >= 0 && de < 4000) {
#if defined(__HIP_DEVICE_COMPILE__)
                    const int32_t simd = (int32_t)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      /* HW_ID.SIMD_ID of the workgroup's wave 0 */
#else
                    const int32_t simd = 0;
#endif
                    const int32_t v[4] = { ds, de, de - ds, simd * 100 };
                    for (int q = 0; q < 4; q++) { PLS_ATOMIC_MAX(&j.result[24 + q], v[q]); PLS_ATOMIC_ADD((uint32_t *)&j.result[28 + q], (uint32_t)v[q]); }
                    PLS_ATOMIC_ADD((uint32_t *)&j.result[32], 1u);
                }
            }
        }
    }
}

/* ---- ENUMERATE, SEEDED (state sets beyond SEG_NS_MAX; every filter): task (f, seg), NT lanes = NT / SEG_NSP channels x SEG_NSP seeds ----
 * The seeds start SEG_KIN pixels in front of the segment (at pixel 0, where the state is known, when the row begins within that reach): a
 * filter that looks at the left pixel gets every left byte within reach of the data, each with the carried terms that go with it when nothing
 * was carried INTO the boundary pixel; none / up get a grid of carried terms (SegParams::seed_small).  What the seeds have become at the
 * segment's first pixel -- a few dozen distinct states: the carried terms contract, the left bytes fall into the phases of the band system --
 * is the segment's ENTRY SET: hashed by value (exported: the chain kernel looks the exit states of the segment in front up in it), given dense
 * ids, and run through the segment like the distinct states of seg_enum_body.  A state of the reference's own path that is in no entry set
 * (1e-4 .. 1e-3 of the boundaries, oracle/seed_study.c) costs time, not correctness: the chain kernel walks that segment step by step. */
#define SEG_SM_ENUM_SEEDED(nt) (SEG_TBL_WORDS * 4 + (SEG_KIN + SEG_L + 1) * 4 * 8 + 2048 + 32 + 4 * SEG_EH_WORDS * 4 + 4 * SEG_EH_WORDS * 2 + 4 * SEG_NSP * 4 + (nt) * 4 + 128)
template <int NT>
PLS_HD void seg_enum_seeded_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int seg, int chalf, unsigned char *smem)
{
    if (cv.finished || cv.active != 1) return;
    const uint32_t W = j.W, bpp = j.bpp;
    constexpr int NCH = NT / SEG_NSP;
    const int c0 = chalf * NCH;
    if ((uint32_t)c0 >= bpp) return;
    const uint32_t x0 = (uint32_t)seg * SEG_L;
    if (x0 >= W) return;
    if (cv.start_x && x0 <= cv.start_x) return;               /* (the epoch's first, partial segment is walked: seg_first_body) */
    const uint32_t kin = (uint32_t)P.kin;
    const uint32_t xs = x0 >= kin ? x0 - kin : 0u;            /* first pixel of the run-in */
    const int nrun = (int)(x0 - xs);
    uint32_t *tw = (uint32_t *)smem;
    SegPix *px = (SegPix *)(smem + SEG_TBL_WORDS * 4);        /* [(SEG_KIN + SEG_L + 1)][4]: slot 0 = boundary pixel xs-1 */
    uint32_t *lut = (uint32_t *)(px + (SEG_KIN + SEG_L + 1) * 4);
    uint32_t *trflag = lut + 512;                             /* [0] some pixel of the window is fully transparent, [1..4] distinct entry states per channel */
    uint32_t *ht = trflag + 8;                                /* [4][SEG_EH_WORDS] key or ~0 */
    uint16_t *dense = (uint16_t *)(ht + 4 * SEG_EH_WORDS);    /* [4][SEG_EH_WORDS] slot -> dense id */
    uint32_t *uniq = (uint32_t *)(dense + 4 * SEG_EH_WORDS);  /* [4][SEG_NSP] the entry states (keys) by dense id */
    uint32_t *keys = uniq + 4 * SEG_NSP;                      /* [NT] */
    const uint32_t y = cv.y;
    const SEG_AS_GLB uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    const bool prof = (P.engine_flags & 1) != 0;
    unsigned long long te[5] = { 0, 0, 0, 0, 0 };
    if (prof) te[0] = PLS_CLOCK();
    PLS_THREADS(tid, NT) {
        if (tid < 8) trflag[tid] = 0u;
        for (int i = tid; i < 4 * SEG_EH_WORDS; i += NT) ht[i] = SEG_NOKEY;
    }
    PLS_SYNC();
    const int npx = nrun + SEG_L + 1;
    PLS_THREADS(tid, NT) {
        constexpr int NTW = (SEG_TBL_WORDS + NT - 1) / NT;
        uint32_t vt[NTW], vl = 0;
        SegPixRaw vp = seg_pix_raw_zero();
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; vt[q] = i < SEG_TBL_WORDS ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (tid < 512) vl = P.lut_a[tid];
        if (tid < npx) vp = seg_pix_fetch(row, nab, e0g, xs - 1 + (uint32_t)tid, W);      /* (xs = 0: slot 0 lies in front of the row -- zeros) */
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
        if (tid < 512) lut[tid] = vl;
        if (tid < npx) {
            seg_pix_split4(px + tid * 4, vp, bpp, xs - 1 + (uint32_t)tid, W);
            if ((bpp & 1u) == 0u && (px[tid * 4 + (bpp - 1u)].w >> 24)) PLS_ATOMIC_OR(trflag, 1u);
        }
    }
    PLS_SYNC();
    const bool trx = trflag[0] != 0u;
    if (prof) te[1] = PLS_CLOCK();
    /* -- the seeds, through the run-in.  A workgroup of a channel PAIR does it in two stages (round 5): SEG_KA steps from every seed, then only the DISTINCT states
     *    that are left (a few dozen of the 256 per channel: the seeds differ in the left byte, and the left bytes fall into the bands' phases at once) through the
     *    rest of the run-in, both channels packed into the first lanes -- two waves instead of eight for most of the run-in's steps.  The distinct states are found
     *    like the entry set below, in the half of the hash arrays a channel pair leaves unused (channels 2, 3); a state the window has no room for is dropped: a seed
     *    less, which costs coverage, never correctness.  The entry set is the same SET of states either way. -- */
    constexpr int SEG_KA = SEG_KA_SEEDED;
    const bool two_stage = NCH == 2 && xs != 0u && nrun > SEG_KA + 2;
    PLS_THREADS(tid, NT) {
        const int lc = tid / SEG_NSP, c = c0 + lc, i = tid % SEG_NSP;
        uint32_t key = SEG_NOKEY;
        if ((uint32_t)c < bpp) {
            SegState st = { 0, 0, 0 };
            bool ok;
            if (xs == 0u) ok = i == 0;                                               /* the row's own start */
            else if (f == 0 || f == 2) {
                ok = i < P.nseed_small;
                const uint32_t w = P.seed_small[ok ? i : 0];
                st.cn = (int)(w & 255u) - 128; st.th = (int)((w >> 8) & 255u) - 128;
            } else {
                const SegPix b = px[c];
                if (b.w >> 24) { ok = i <= 2 * P.tmax; st.cn = i - P.tmax; }         /* a forced transparent alpha in front: byte 0, difference 0 */
                else {
                    const int D = (int)(b.w & 255u) + b.e0 - i;                     /* the boundary pixel's difference if nothing was carried into it */
                    ok = D >= -P.dmax && D <= P.dmax;
                    int rem = 0, thr = 0;
                    seg_rem_thr(lut, P.bleed, ok ? D : 0, rem, thr);
                    st.left = i; st.cn = rem; st.th = thr;
                }
            }
            if (ok) {
                const int nst = two_stage ? SEG_KA : nrun;
                const int bad = nst ? seg_run_fast_f(f, trx, px + 4 + c, 4, nst, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut)) : 0;
                if (!bad) key = seg_eh_key(st.left, st.cn, st.th);
            }
        }
        keys[tid] = key;
    }
    PLS_SYNC();
    if (two_stage) {
        /* the distinct states after SEG_KA steps: list uniq[2 + lc][], count trflag[3 + lc] */
        PLS_THREADS(tid, NT) {
            const int lc = tid / SEG_NSP, i = tid % SEG_NSP;
            const uint32_t key = keys[tid];
            if (lc < 2 && key != SEG_NOKEY && (i == 0 || keys[tid - 1] != key)) {
                const uint32_t base = seg_eh_base(key);
                for (int q = 0; q < SEG_EHW; q++) {
                    const uint32_t old = PLS_ATOMIC_CAS(&ht[(2 + lc) * SEG_EH_WORDS + base + q], SEG_NOKEY, key);
                    if (old == SEG_NOKEY) {
                        const uint32_t d = PLS_ATOMIC_ADD_RET(&trflag[3 + lc], 1u);
                        if (d < SEG_NSP) uniq[(2 + lc) * SEG_NSP + d] = key;
                        break;
                    }
                    if (old == key) break;
                }
            }
        }
        PLS_SYNC();
        /* ... through the rest of the run-in, one lane each */
        const uint32_t DA0 = trflag[3] < SEG_NSP ? trflag[3] : SEG_NSP, DA1 = trflag[4] < SEG_NSP ? trflag[4] : SEG_NSP;
        PLS_THREADS(tid, NT) {
            uint32_t key = SEG_NOKEY;
            if ((uint32_t)tid < DA0 + DA1) {
                const int lc = (uint32_t)tid < DA0 ? 0 : 1, c = c0 + lc;
                const uint32_t i = lc ? (uint32_t)tid - DA0 : (uint32_t)tid;
                SegState st = seg_eh_state(uniq[(2 + lc) * SEG_NSP + i]);
                const int bad = seg_run_fast_f(f, trx, px + (1 + SEG_KA) * 4 + c, 4, nrun - SEG_KA, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                if (!bad) key = seg_eh_key(st.left, st.cn, st.th);
            }
            keys[tid] = key;
        }
        PLS_SYNC();
        /* -- the entry set (as below), from the packed lanes -- */
        PLS_THREADS(tid, NT) {
            if ((uint32_t)tid < DA0 + DA1) {
                const int c = (uint32_t)tid < DA0 ? 0 : 1;
                const uint32_t i = c ? (uint32_t)tid - DA0 : (uint32_t)tid;
                const uint32_t key = keys[tid];
                if (key != SEG_NOKEY && (i == 0 || keys[tid - 1] != key)) {
                    const uint32_t base = seg_eh_base(key);
                    for (int q = 0; q < SEG_EHW; q++) {
                        const uint32_t old = PLS_ATOMIC_CAS(&ht[c * SEG_EH_WORDS + base + q], SEG_NOKEY, key);
                        if (old == SEG_NOKEY) {
                            const uint32_t d = PLS_ATOMIC_ADD_RET(&trflag[1 + c], 1u);
                            dense[c * SEG_EH_WORDS + base + q] = (uint16_t)(d < SEG_NSP ? d : 0xffffu);
                            if (d < SEG_NSP) uniq[c * SEG_NSP + d] = key;
                            break;
                        }
                        if (old == key) break;
                    }
                }
            }
        }
        PLS_SYNC();
    } else {
    /* -- the entry set: distinct states, hashed by value (only the first lane of a run of equal neighbours inserts) -- */
    PLS_THREADS(tid, NT) {
        const int c = tid / SEG_NSP, i = tid % SEG_NSP;
        const uint32_t key = keys[tid];
        if (key != SEG_NOKEY && (i == 0 || keys[tid - 1] != key)) {
            const uint32_t base = seg_eh_base(key);
            for (int q = 0; q < SEG_EHW; q++) {
                const uint32_t old = PLS_ATOMIC_CAS(&ht[c * SEG_EH_WORDS + base + q], SEG_NOKEY, key);
                if (old == SEG_NOKEY) {
                    const uint32_t d = PLS_ATOMIC_ADD_RET(&trflag[1 + c], 1u);
                    dense[c * SEG_EH_WORDS + base + q] = (uint16_t)(d < SEG_NSP ? d : 0xffffu);
                    if (d < SEG_NSP) uniq[c * SEG_NSP + d] = key;
                    break;
                }
                if (old == key) break;
            }                                                                    /* (a full window: the state gets no id) */
        }
    }
    PLS_SYNC();
    }
    if (prof) te[2] = PLS_CLOCK();
    PLS_THREADS(tid, NT) {
        for (int i = tid; i < NCH * SEG_EH_WORDS; i += NT) {
            const int lc = i / SEG_EH_WORDS;
            if ((uint32_t)(c0 + lc) >= bpp) continue;
            const uint32_t k = ht[i], d = dense[i];
            j.ehash[(((size_t)f * j.nseg + seg) * 4 + c0) * SEG_EH_WORDS + i] = (k == SEG_NOKEY || d >= SEG_NSP) ? SEG_EH_EMPTY : ((k << 8) | d);
        }
    }
    /* -- the segment itself from every entry state: dense id -> checkpoints, exit state -- */
    PLS_THREADS(tid, NT) {
        /* (the entry states of the workgroup's channels packed into its first lanes, like the distinct states of seg_enum_body) */
        uint32_t Dc[NCH];
        for (int k = 0; k < NCH; k++) Dc[k] = trflag[1 + k] < SEG_NSP ? trflag[1 + k] : SEG_NSP;
        int lc = 0, i = tid;
        uint32_t D = Dc[0];
        PLS_UNROLL
        for (int k = 0; k + 1 < NCH; k++) if (lc == k && (uint32_t)i >= Dc[k]) { i -= (int)Dc[k]; lc = k + 1; D = Dc[k + 1]; }
        const int c = c0 + lc;
        if (tid < NCH && (uint32_t)(c0 + tid) < bpp) j.dcnt[((size_t)f * j.nseg + seg) * 4 + c0 + tid] = trflag[1 + tid] < SEG_NSP ? trflag[1 + tid] : SEG_NSP;
        if ((uint32_t)c < bpp && (uint32_t)i < D) {
            SegState st = seg_eh_state(uniq[lc * SEG_NSP + i]);
            const size_t slot = (((size_t)f * j.nseg + seg) * 4 + c) * SEG_NSP + i;
            const SegPix *ps = px + (nrun + 1) * 4 + c;
            int bad = 0;
            for (int part = 0; part < SEG_PARTS; part++) {
                if (part) j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                bad |= seg_run_fast_f(f, trx, ps + part * SEG_PL * 4, 4, SEG_PL, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
            }
            j.rst[slot] = bad ? 0xFFFFFFFFu : seg_state_pack(st);             /* (rout of a seeded set: the successor's id, written by seg_gather_seeded_body) */
        }
        if (prof && tid == 0 && !SEG_EXPERIMENT_REPLAY_CLOCKS) {
            te[3] = PLS_CLOCK(); te[4] = te[3];
            for (int q = 0; q < 4; q++) { PLS_ATOMIC_MAX(&j.result[24 + q], (int32_t)(te[q + 1] - te[q])); PLS_ATOMIC_ADD((uint32_t *)&j.result[28 + q], (uint32_t)(te[q + 1] - te[q])); }
            PLS_ATOMIC_ADD((uint32_t *)&j.result[32], 1u);
            PLS_ATOMIC_ADD((uint32_t *)&j.result[33], two_stage ? 2u * (trflag[1] + trflag[2]) : trflag[1] + trflag[2] + trflag[3] + trflag[4]);
        }
    }
}

/* ---- ENUMERATE, none / up (state = (cn, th), SEG_NSS lanes per channel): task (f, SEG_SMALL_SEGS segments from seg0) -------- */
template <int NT>
PLS_HD void seg_enum_small_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int seg0, unsigned char *smem)
{
    if (cv.finished || cv.active != 1) return;
    const uint32_t W = j.W, bpp = j.bpp;
    constexpr int NSEGS = NT / (4 * SEG_NSS);                 /* segments of this workgroup */
    if ((uint32_t)(seg0 + NSEGS) * SEG_L <= cv.start_x) return;
    uint32_t *tw = (uint32_t *)smem;
    uint32_t *lut = tw + SEG_TBL_WORDS;
    SegPix *px = (SegPix *)(lut + 512);                        /* [NSEGS][SEG_L][4] */
    const uint32_t y = cv.y;
    const uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    uint32_t *trflag = (uint32_t *)(px + NSEGS * SEG_L * 4);
    PLS_THREADS(tid, NT) { if (tid == 0) *trflag = 0u; }
    PLS_SYNC();
    PLS_THREADS(tid, NT) {
        constexpr int NTW = (SEG_TBL_WORDS + NT - 1) / NT;
        uint32_t vt[NTW], vl = 0;
        SegPixRaw vp = seg_pix_raw_zero();
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; vt[q] = i < SEG_TBL_WORDS ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (tid < 512) vl = P.lut_a[tid];
        if (tid < NSEGS * SEG_L) vp = seg_pix_fetch(row, nab, e0g, (uint32_t)seg0 * SEG_L + (uint32_t)tid, W);
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
        if (tid < 512) lut[tid] = vl;
        if (tid < NSEGS * SEG_L) {
            seg_pix_split4(px + tid * 4, vp, bpp, (uint32_t)seg0 * SEG_L + (uint32_t)tid, W);
            if ((bpp & 1u) == 0u && (px[tid * 4 + (bpp - 1u)].w >> 24)) PLS_ATOMIC_OR(trflag, 1u);
        }
    }
    PLS_SYNC();
    const bool trx = *trflag != 0u;
    PLS_THREADS(tid, NT) {
        const int sl = tid / (4 * SEG_NSS), c = (tid / SEG_NSS) & 3, i = tid % SEG_NSS;
        const uint32_t seg = (uint32_t)seg0 + (uint32_t)sl, x0 = seg * SEG_L;
        if (seg < j.nseg && x0 < W && (x0 > cv.start_x || cv.start_x == 0) && (uint32_t)c < bpp && i < P.ns_small) {
            SegState st;
            uint32_t out = SEG_INVALID;
            const size_t slot = (((size_t)f * j.nseg + seg) * 4 + c) * SEG_NSP + i;
            if (seg_small_decode(P, i, st)) {
                int bad = 0;
                for (int part = 0; part < SEG_PARTS; part++) {
                    if (part) j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                    bad |= seg_run_fast_f(f, trx, px + (sl * SEG_L + part * SEG_PL) * 4 + c, 4, SEG_PL, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                }
                if (!bad) out = seg_small_encode(P, st);
            } else {
                for (int part = 1; part < SEG_PARTS; part++) j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = 0xFFFFFFFFu;
            }
            /* no dedupe for the handful of (cn, th) states: the dense id of an entry index is the index itself */
            j.maps[(((size_t)f * j.nseg + seg) * 4 + c) * (size_t)P.nsp + i] = (uint16_t)i;
            j.rout[(((size_t)f * j.nseg + seg) * 4 + c) * SEG_NSP + i] = (uint16_t)out;
            j.rst[(((size_t)f * j.nseg + seg) * 4 + c) * SEG_NSP + i] = out == SEG_INVALID ? 0xFFFFFFFFu : seg_state_pack(st);
            if (i == 0) j.dcnt[((size_t)f * j.nseg + seg) * 4 + c] = (uint32_t)P.ns_small;
        }
    }
}

/* ---- ENUMERATE IN UNITS (batches; exhaustive state sets): task (f, group of SEG_UNC pairs (unit, channel)) --------------------------------------
 * What a batch pays for is not the length of the dependent path but the instructions its waves issue and the workgroups a CU can hold.  So:
 *   - a UNIT is SEG_UNIT consecutive segments.  It is run from EVERY state only at its first pixel (SEG_K1 steps, LANES states per pair at a time,
 *     then the dedupe of seg_enum_body); the distinct states that are left (a dozen or two) run on through ALL its segments, leaving the same
 *     records per segment as seg_enum_body -- checkpoints at every part, the exit state of every segment -- with the unit's dense id in every
 *     one of them, and the exit INDEX (the next unit's entry) only at its end.  The 253-state start is paid once per unit, not once per segment;
 *   - a workgroup takes SEG_UNC pairs (unit, channel) of ONE candidate, from the row's linear list unit * bpp + channel -- in turns of as many
 *     pairs as its 1024 lanes hold states for -- and then packs the distinct states of ALL its pairs into its first lanes: ~200 lanes = four
 *     waves that share one copy of the candidate's tables (12.8 KB) instead of one or two half-empty waves per 35 KB workgroup;
 *   - none / up (LANES = 32: their state is (cn, th)) go through the same body, dedupe included: ~3 distinct states per pair instead of 17 lanes,
 *     twenty pairs a workgroup.
 * The chain kernel composes UNITS (seg_chain_body reads SegParams::unit); replay and validation do not know the difference. */
#ifndef SEG_UNC_SMALL
#define SEG_UNC_SMALL (SEG_UNIT <= 3 ? 20 : (SEG_UNIT <= 4 ? 15 : (SEG_UNIT <= 6 ? 10 : 7)))          /* (unit, channel) pairs per workgroup for none / up with their small state set */
#endif
#define SEG_UNC_SMALL_OF(unit) ((unit) == SEG_UNIT ? SEG_UNC_SMALL : (SEG_UNC_SMALL < 20 ? SEG_UNC_SMALL : 20))      /* ... and segment by segment (round 6: seg_k_enum_unit<1>) */
#define SEG_UNPX (SEG_UNC_SMALL * (SEG_UNIT * SEG_L + 1))   /* pixel records of a workgroup's pairs */
/* LDS of the unit enumeration: tables, split table, the pool, the first records of every pair, bookkeeping -- and ONE region that holds the first phase's
 * scratch (hash tables, per-turn lists, keys: 10 KB) and then, for the second phase, the pairs' full pixel records (15.5 KB for the twenty pairs of none / up):
 * 35.8 KB = four workgroups per CU.  (The first version carved both side by side: 52 KB, three per CU, and the CUs' slots, not their issue, set the kernel's
 * time; the second was sized for 1024 threads: 39.9 KB.  Sixteen / twelve pairs of none / up = 32.5 / 28.6 KB = FIVE per CU were measured: 464 / 451 Mpx/s
 * for 32 frames against 465 -- the slots are not what it is short of; the 4 KB less are worth 1.3 %, though: the other launch groups' control, validation and
 * replay workgroups want 37 - 50 KB next to two or three of these.) */
#define SEG_UN_K1MAX 4
#define SEG_UN_SCRATCH (2 * SEG_UNT * 4 + 2 * SEG_UNT * 2 + SEG_UNT * 4 + SEG_UNT * 4)        /* hash tables (a turn's pairs x twice their lanes), dense ids, the turn's lists, one key a thread */
#define SEG_UN_PHASE2 (SEG_UNPX * 8 + SEG_UPOOL * 4)      /* the pairs' records and the second list of distinct states (behind the unit's first segment) */
#define SEG_UN_SEEDX (SEG_UN_SCRATCH + (SEG_UNC_SEEDS > SEG_UNC_SEEDS1 ? SEG_UNC_SEEDS : SEG_UNC_SEEDS1) * (SEG_SEED_KMAX + 1) * 8 + (SEG_UNT / SEG_SEED_LANES) * 512)   /* (from seeds: the run-in's records and a turn's staged entry maps behind the scratch) */
#define SEG_UN_REGION ((SEG_UN_SCRATCH > SEG_UN_PHASE2 ? SEG_UN_SCRATCH : SEG_UN_PHASE2) > SEG_UN_SEEDX ? (SEG_UN_SCRATCH > SEG_UN_PHASE2 ? SEG_UN_SCRATCH : SEG_UN_PHASE2) : SEG_UN_SEEDX)
#define SEG_SM_ENUM_UNIT (SEG_TBL_WORDS * 4 + 2048 + SEG_UPOOL * 4 + SEG_UNC_SMALL * (SEG_UN_K1MAX + 1) * 8 + 512 + SEG_UN_REGION)
/* set bits among bits [a, b) of a bit array */
PLS_HD uint32_t seg_bits_count(const uint32_t *bits, uint32_t a, uint32_t b)
{
    uint32_t n = 0;
    for (uint32_t w = a >> 5; w <= (b ? (b - 1u) >> 5 : 0u) && a < b; w++) {
        uint32_t m = bits[w];
        if (w == (a >> 5)) m &= 0xFFFFFFFFu << (a & 31u);
        if (w == ((b - 1u) >> 5) && (b & 31u)) m &= 0xFFFFFFFFu >> (32u - (b & 31u));
        n += (uint32_t)__builtin_popcount(m);
    }
    return n;
}
/* SEEDS (round 6): the first phase does not run every state from the unit's first pixel but P.seed_n seeds from P.seed_kin pixels in front of it; the states they have
 * become at the unit's first pixel -- encoded relative to the boundary pixel like any exit state -- are the unit's entry set: dense ids in the order of the dedupe, the
 * pair's entry map holds an id for their indices and "none" everywhere else (a row whose true state is none of them is broken off there by the chain kernel and resumed
 * in an epoch: costs attempts, never correctness -- the validation is the ground truth either way).  LANES = SEG_SEED_LANES lanes a pair: eight pairs a turn. */
template <int LANES, int UNIT, int NC, bool SEEDS = false>
PLS_HD void seg_enum_unit_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int grp, unsigned char *smem)
{
    if (cv.finished || cv.active != 1) return;
    constexpr int NT = SEG_UNT, HT = 2 * LANES;
    constexpr int CPR = NT / LANES < NC ? NT / LANES : NC;      /* pairs per turn of the first phase */
    constexpr int ROUNDS = (NC + CPR - 1) / CPR;
    constexpr uint32_t E = UNIT, UL = E * SEG_L, NPX = UL + 1;
    static_assert(CPR * HT <= 2 * SEG_UNT && CPR * LANES <= SEG_UNT && NC <= 24 && NC * (UNIT * SEG_L + 1) <= SEG_UNPX, "hash tables, per-turn lists, pixel records and the bookkeeping words are sized for this");
    const uint32_t W = j.W, bpp = j.bpp, nseg = j.nseg, nunit = (nseg + E - 1) / E, ncombo = nunit * bpp;
    const uint32_t q0 = (uint32_t)grp * NC;
    if (q0 >= ncombo) return;
    const uint32_t sx = cv.start_x;
    /* an epoch that starts inside the row: the unit that holds its first pixel (and everything in front) is walked by seg_first_body */
    { const uint32_t ql = seg_umin(q0 + NC, ncombo) - 1u; if (sx && (ql / bpp) * UL <= sx) return; }
    constexpr uint32_t NP1 = SEEDS ? SEG_SEED_KMAX + 1 : SEG_UN_K1MAX + 1;   /* records of a pair the first phase reads: the boundary pixel and SEG_K1 more (SEEDS: the pixel in front of the run-in and the run-in) */
    uint32_t *tw = (uint32_t *)smem;
    uint32_t *lut = tw + SEG_TBL_WORDS;
    uint32_t *pool = lut + 512;                               /* [SEG_UPOOL] the distinct states of all pairs, pair behind pair */
    SegPix *px1_std = (SegPix *)(pool + SEG_UPOOL);           /* [NC][NP1]: one channel's first records of a pair, slot 0 = the boundary pixel in front of the unit */
    uint32_t *misc = (uint32_t *)(px1_std + SEG_UNC_SMALL * (SEG_UN_K1MAX + 1)); /* [0] transparent pixel seen, [8 + kk] distinct states of the turn's pair kk, [32 + k] first pool slot of pair k (.. [32 + NC] = total),
                                                                 [64 + k] the same for the SECOND list (.. [64 + NC] = its total), [96 .. 127] one bit per lane: it represents a state of the second list */
    /* first phase: */
    uint32_t *ht = misc + 128;                                /* [CPR][HT] key or ~0 (this turn) */
    uint16_t *dense = (uint16_t *)(ht + 2 * SEG_UNT);         /* [CPR][HT] slot -> dense id */
    uint32_t *uniqr = (uint32_t *)(dense + 2 * SEG_UNT);      /* [CPR][LANES] this turn's distinct states by dense id */
    uint32_t *keys = uniqr + SEG_UNT;                         /* [NT] */
    /* SEEDS: the run-in's records and the turn's entry maps, behind the first phase's scratch (the region is as large as the second phase's records) */
    SegPix *px1 = SEEDS ? (SegPix *)(keys + SEG_UNT) : px1_std;
    uint16_t *mapl = (uint16_t *)(px1 + NC * NP1);            /* (SEEDS) [CPR][256]: entry index -> dense id of the turn's pairs, staged here and stored coalesced */
    static_assert(!SEEDS || (SEG_UN_SCRATCH + NC * (int)NP1 * 8 + CPR * 512 <= SEG_UN_REGION), "the run-in's records and the staged maps fit behind the scratch");
    static_assert(!SEEDS || LANES == SEG_SEED_LANES, "one lane per seed");
    /* the run-in: P.seed_kin pixels (8) -- and half as many again for an image that has had rows broken off (SegJob::nbreak): real photographs lose the true state at a boundary in
     * ~1e-3 of the cases after 8 pixels and a quarter of that after 12 (oracle/seed_study.c on the suite's lena: sub 17 -> 4 of 23 040, paeth 5 -> 2), the generator's frames next to never
     * either way, and the longer run-in costs every workgroup 1 - 3 % */
    const int KR = SEEDS ? seg_min(seg_max(j.nbreak >= SEG_SEED_LONG_AFTER ? P.seed_kin + P.seed_kin / 2 : P.seed_kin, 1), SEG_SEED_KMAX) : 0;
    /* second phase, in the same place: */
    SegPix *px = (SegPix *)(misc + 128);                      /* [NC][NPX]: all records of a pair */
    uint32_t *pool2 = (uint32_t *)(px + SEG_UNPX);            /* [SEG_UPOOL] the states that are still distinct behind the unit's first segment, at the pairs' places of the first list */
    static_assert(SEG_K1 <= SEG_UN_K1MAX && SEG_K1_ONE_CHUNK <= SEG_UN_K1MAX, "the first phase's records");
    const uint32_t y = cv.y;
    const SEG_AS_GLB uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    const int nstates = seg_is_small(P, f) ? P.ns_small : P.ns;
    /* phase clocks (PNGLOSS_HIP_SEGPROF; the workgroups of the filters that look at the left pixel): the enumeration's slots of the result record --
     * "load" = staging, "first steps + dedupe" = the first phase's turns, "remaining steps" = the unit's first segment, "map" = second dedupe + the rest of the unit */
    const bool prof = (P.engine_flags & 1) != 0 && LANES == SEG_NSP;
    unsigned long long te[5] = { 0, 0, 0, 0, 0 };
    if (prof) te[0] = PLS_CLOCK();
    PLS_THREADS(tid, NT) { if (tid < 128) misc[tid] = 0u; }
    PLS_SYNC();
    PLS_THREADS(tid, NT) {
        constexpr int NTW = (SEG_TBL_WORDS + NT - 1) / NT;
        static_assert(NC * (SEG_UN_K1MAX + 1) <= NT, "one first-phase record per thread");
        uint32_t vt[NTW], vl = 0;
        SegPix vp = seg_pix_make(0, 0, 0, 0, 0);
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; vt[q] = i < SEG_TBL_WORDS ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (tid < 512) vl = P.lut_a[tid];
        {
            const uint32_t k = (uint32_t)tid / NP1, pq = (uint32_t)tid % NP1, cq = q0 + k;
            if (k < (uint32_t)NC && cq < ncombo) {
                const uint32_t u = cq / bpp, c = cq % bpp, x = SEEDS ? u * UL + pq - 1u - (uint32_t)KR : u * UL + pq - 1u;          /* (pq = 0 in front of the row: wraps beyond W -- a zero record; SEEDS: slot 0 = the pixel in front of the run-in, slot KR = the boundary pixel) */
                if (x < W && (!SEEDS || pq <= (uint32_t)KR)) vp = seg_pix_load(row, nab, e0g, bpp, x, (int)c);
            }
        }
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
        if (tid < 512) lut[tid] = vl;
        if ((uint32_t)tid < (uint32_t)NC * NP1) { px1[tid] = vp; if (vp.w >> 24) PLS_ATOMIC_OR(&misc[0], 1u); }
    }
    PLS_SYNC();
    const bool trx1 = misc[0] != 0u;
    if (prof) te[1] = PLS_CLOCK();
    const int K1 = SEEDS ? 0 : seg_k1(nstates);
    /* -- first phase, a turn of CPR pairs at a time: SEG_K1 steps from every state, the dedupe, the pair's entry map (entry index -> dense id) -- */
    for (int r = 0; r < ROUNDS; r++) {
        PLS_THREADS(tid, NT) {
            for (int i = tid; i < CPR * HT; i += NT) ht[i] = 0xffffffffu;
            if (tid < CPR) misc[8 + tid] = 0u;
            if (SEEDS) for (int i = tid; i < CPR * 128; i += NT) ((uint32_t *)mapl)[i] = 0xffffffffu;
        }
        PLS_SYNC();
        if (SEEDS) {
            /* the seeds through the run-in; key = the state at the unit's first pixel | its entry index << 24 (ns <= 255: no key is all ones) */
            PLS_THREADS(tid, NT) {
                const int kk = tid / LANES, k = r * CPR + kk, i = tid % LANES;
                const uint32_t cq = q0 + (uint32_t)k;
                uint32_t key = 0xffffffffu;
                if (kk < CPR && k < NC && cq < ncombo && !(sx && (cq / bpp) * UL <= sx)) {
                    const SegPix *pk = px1 + (size_t)k * NP1;
                    if (cq / bpp == 0u) { if (i == 0) key = P.idx0_big << 24; }        /* the row's first unit: its one entry state is known (nothing carried, no left pixel) */
                    else if (i < P.seed_n) {
                        SegState st;
                        if (seg_state_decode(P, (int)P.seed_idx[i], pk[0], st)) {
                            const int bad = seg_run_fast_f(f, trx1, pk + 1, 1, KR, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                            const uint32_t idx = bad ? (uint32_t)SEG_INVALID : seg_state_encode(P, pk[KR], st);
                            if (idx != (uint32_t)SEG_INVALID) key = (uint32_t)(st.left & 255) | ((uint32_t)(st.cn & 255) << 8) | ((uint32_t)(st.th & 255) << 16) | (idx << 24);
                        }
                    }
                }
                keys[tid] = key;
            }
            PLS_SYNC();
            PLS_THREADS(tid, NT) {
                const int kk = tid / LANES, i = tid % LANES;
                const uint32_t key = keys[tid];
                if (kk < CPR && key != 0xffffffffu && (i == 0 || keys[tid - 1] != key)) {
                    uint32_t h = ((key * 0x9E3779B1u) >> 16) & (uint32_t)(HT - 1);
                    for (int probe = 0; probe < HT; probe++) {
                        const uint32_t old = PLS_ATOMIC_CAS(&ht[kk * HT + h], 0xffffffffu, key);
                        if (old == 0xffffffffu) {
                            const uint32_t d = PLS_ATOMIC_ADD_RET(&misc[8 + kk], 1u);
                            dense[kk * HT + h] = (uint16_t)d;
                            uniqr[kk * LANES + d] = key;
                            mapl[kk * 256 + (key >> 24)] = (uint16_t)d;
                            break;
                        }
                        if (old == key) break;
                        h = (h + 1) & (uint32_t)(HT - 1);
                    }
                }
            }
            PLS_SYNC();
            PLS_THREADS(tid, NT) {
                /* the turn's entry maps, whole (every index the seeds did not reach: none) */
                for (int e = tid; e < CPR * 128; e += NT) {
                    const int kk = e >> 7, w = e & 127, k = r * CPR + kk;
                    const uint32_t cq = q0 + (uint32_t)k;
                    if (k < NC && cq < ncombo && !(sx && (cq / bpp) * UL <= sx) && 2 * w < P.nsp) {
                        const uint32_t u = cq / bpp, c = cq % bpp;
                        ((SEG_AS_GLB uint32_t *)(j.maps + (((size_t)f * nseg + (size_t)u * E) * 4 + c) * (size_t)P.nsp))[w] = ((const uint32_t *)mapl)[e];
                    }
                }
            }
            PLS_SYNC();
        } else
        for (int i0 = 0; i0 < nstates; i0 += LANES) {
            PLS_THREADS(tid, NT) {
                const int kk = tid / LANES, k = r * CPR + kk, i = i0 + tid % LANES;
                const uint32_t cq = q0 + (uint32_t)k;
                uint32_t key = 0xffffffffu;
                if (kk < CPR && k < NC && cq < ncombo && i < nstates && !(sx && (cq / bpp) * UL <= sx)) {
                    const SegPix *pk = px1 + (size_t)k * NP1;
                    SegState st;
                    if (seg_any_decode(P, f, i, pk[0], st)) {
                        const int bad = seg_run_fast_f(f, trx1, pk + 1, 1, K1, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                        if (!bad && st.cn >= -128 && st.cn <= 127) key = (uint32_t)(st.left & 255) | ((uint32_t)(st.cn & 255) << 8) | ((uint32_t)(st.th & 255) << 16);
                    }
                }
                keys[tid] = key;
            }
            PLS_SYNC();
            PLS_THREADS(tid, NT) {
                const int kk = tid / LANES, i = tid % LANES;
                const uint32_t key = keys[tid];
                if (kk < CPR && key != 0xffffffffu && (i == 0 || keys[tid - 1] != key)) {
                    uint32_t h = ((key * 0x9E3779B1u) >> 16) & (uint32_t)(HT - 1);
                    for (int probe = 0; probe < HT; probe++) {
                        const uint32_t old = PLS_ATOMIC_CAS(&ht[kk * HT + h], 0xffffffffu, key);
                        if (old == 0xffffffffu) {
                            const uint32_t d = PLS_ATOMIC_ADD_RET(&misc[8 + kk], 1u);
                            dense[kk * HT + h] = (uint16_t)(d < (uint32_t)LANES ? d : 0xffffu);
                            if (d < (uint32_t)LANES) uniqr[kk * LANES + d] = key;
                            break;
                        }
                        if (old == key) break;
                        h = (h + 1) & (uint32_t)(HT - 1);
                    }
                }
            }
            PLS_SYNC();
            PLS_THREADS(tid, NT) {
                const int kk = tid / LANES, k = r * CPR + kk, i = i0 + tid % LANES;
                const uint32_t cq = q0 + (uint32_t)k;
                if (kk < CPR && k < NC && cq < ncombo && i < nstates && !(sx && (cq / bpp) * UL <= sx)) {
                    const uint32_t key = keys[tid];
                    uint32_t d = 0xffffu;
                    if (key != 0xffffffffu) {
                        uint32_t h = ((key * 0x9E3779B1u) >> 16) & (uint32_t)(HT - 1);
                        for (int probe = 0; probe < HT; probe++) {
                            const uint32_t at = ht[kk * HT + h];
                            if (at == key) { d = dense[kk * HT + h]; break; }
                            if (at == 0xffffffffu) break;
                            h = (h + 1) & (uint32_t)(HT - 1);
                        }
                    }
                    const uint32_t u = cq / bpp, c = cq % bpp;
                    j.maps[(((size_t)f * nseg + (size_t)u * E) * 4 + c) * (size_t)P.nsp + i] = (uint16_t)d;
                }
            }
            PLS_SYNC();
        }
        /* the turn's lists into the pool, pair behind pair (a pair whose states the pool has no room for keeps what fits: the ids beyond get no lane,
         * dcnt says so, and the chain treats them like any state that left the tables) */
        PLS_THREADS(tid, NT) {
            if (tid == 0) {
                for (int kk = 0; kk < CPR; kk++) {
                    const int k = r * CPR + kk;
                    if (k >= NC) break;
                    uint32_t D = misc[8 + kk] < (uint32_t)LANES ? misc[8 + kk] : (uint32_t)LANES;
                    const uint32_t b0 = misc[32 + k];
                    if (b0 + D > SEG_UPOOL) D = SEG_UPOOL - b0;
                    misc[8 + kk] = D;
                    misc[32 + k + 1] = b0 + D;
                }
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            const int kk = tid / LANES, k = r * CPR + kk, i = tid % LANES;
            if (kk < CPR && k < NC && (uint32_t)i < misc[8 + kk]) pool[misc[32 + k] + (uint32_t)i] = uniqr[kk * LANES + i];
        }
        PLS_SYNC();
    }
    if (prof) te[2] = PLS_CLOCK();
    /* -- the pairs' full records, where the first phase's scratch was (nobody reads that any more: a barrier lies behind its last use) -- */
    PLS_THREADS(tid, NT) {
        constexpr int NPI = (NC * (int)NPX + NT - 1) / NT;
        SegPix vp[NPI];
        PLS_UNROLL
        for (int q = 0; q < NPI; q++) {
            const uint32_t t = (uint32_t)tid + (uint32_t)q * NT, k = t / NPX, pq = t % NPX, cq = q0 + k;
            vp[q] = seg_pix_make(0, 0, 0, 0, 0);
            if (k < (uint32_t)NC && cq < ncombo && misc[32 + k + 1] != misc[32 + k]) {            /* (a pair without a lane needs no records) */
                const uint32_t u = cq / bpp, c = cq % bpp, x = u * UL + pq - 1u;
                if (x < W) vp[q] = seg_pix_load(row, nab, e0g, bpp, x, (int)c);
            }
        }
        PLS_UNROLL
        for (int q = 0; q < NPI; q++) {
            const uint32_t t = (uint32_t)tid + (uint32_t)q * NT;
            if (t < (uint32_t)NC * NPX) { px[t] = vp[q]; if (vp[q].w >> 24) PLS_ATOMIC_OR(&misc[0], 1u); }
        }
    }
    PLS_SYNC();
    const bool trx = misc[0] != 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    /* From here on only the first misc[32 + NC] lanes have work.  The waves behind them END here: a workgroup of 16 waves keeps 16 of the CU's 32 wave slots
     * as long as its waves sit in the barriers below -- with them gone the CU takes the next workgroup (measured: profiles/r05_unit_groups.txt).  A barrier
     * counts the waves that have not ended, so the ones that stay are not held up.  Wave 0 always stays: its first NC lanes write the pairs' dcnt below -- also when
     * EVERY state of every pair left the tables in the first phase (pool total 0): the chain kernel must find "no distinct states" there, not an earlier row's count. */
    if (threadIdx.x >= 64u && (threadIdx.x & ~63u) >= misc[32 + NC]) return;
#endif
    /* -- second phase: the distinct states of all pairs, one lane each, through the unit's FIRST segment -- */
    PLS_THREADS(tid, NT) {
        const uint32_t total = misc[32 + NC];
        if (tid < NC) {
            const uint32_t cq = q0 + (uint32_t)tid;
            if (cq < ncombo && !(sx && (cq / bpp) * UL <= sx)) {
                const uint32_t u = cq / bpp, c = cq % bpp, D = misc[32 + tid + 1] - misc[32 + tid];
                for (uint32_t sg = u * E; sg < seg_umin(u * E + E, nseg); sg++) j.dcnt[((size_t)f * nseg + sg) * 4 + c] = D;
            }
        }
        if ((uint32_t)tid < total) {
            int k = 0;
            PLS_UNROLL
            for (int q = 1; q < NC; q++) k += (uint32_t)tid >= misc[32 + q] ? 1 : 0;
            const uint32_t i = (uint32_t)tid - misc[32 + k], cq = q0 + (uint32_t)k, u = cq / bpp, c = cq % bpp;
            const uint32_t key = pool[tid];
            SegState st;
            st.left = (int)(key & 255u); st.cn = seg_sext8((int)(key >> 8)); st.th = seg_sext8((int)(key >> 16));
            const SegPix *pk = px + (size_t)k * NPX;
            const uint32_t sg0 = u * E, nsg = seg_umin(E, nseg - sg0);
            const size_t slot = (((size_t)f * nseg + sg0) * 4 + c) * SEG_NSP + i;
            int bad = 0;
            for (int part = 0; part < SEG_PARTS; part++) {
                if (part) j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                const int skip = part == 0 ? K1 : 0;
                bad |= seg_run_fast_f(f, trx, pk + 1 + part * SEG_PL + skip, 1, SEG_PL - skip, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
            }
            const uint32_t ps0 = bad ? 0xFFFFFFFFu : seg_state_pack(st);
            j.rst[slot] = ps0;
            SEG_DEBUG_STATE(f, (long)cv.y * 100000 + (long)cq, 0, i, ps0);
            if (nsg == 1u) j.rout[slot] = (uint16_t)(bad ? (uint32_t)SEG_INVALID : seg_any_encode(P, f, pk[SEG_L], st));
            pool[tid] = nsg > 1u ? ps0 : 0xFFFFFFFFu;         /* (the lane's start key is used up: its place takes the state it has reached) */
        }
    }
    if (prof) { PLS_SYNC(); te[3] = PLS_CLOCK(); }
    if (E > 1u) {
        /* -- the states keep merging (measured on the CPU harness, profiles/r05_state_merging.txt: of the ~17 states a pair has four pixels into its unit,
         *    ONE is left at the first segment's end for none / up / average, one to two for paeth, 2 .. 18 for sub).  So the lanes are deduplicated once
         *    more there: a lane whose state no earlier lane of its pair has represents it (one bit a lane), the representatives get the ids of a SECOND
         *    list, rout of the unit's first segment takes an id of the first list to its id in the second, and only the second list runs on. -- */
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            const uint32_t total = misc[32 + NC];
            if ((uint32_t)tid < total) {
                const uint32_t ps = pool[tid];
                if (ps != 0xFFFFFFFFu) {
                    int k = 0;
                    PLS_UNROLL
                    for (int q = 1; q < NC; q++) k += (uint32_t)tid >= misc[32 + q] ? 1 : 0;
                    uint32_t rep = (uint32_t)tid;
                    for (uint32_t jj = misc[32 + k]; jj < (uint32_t)tid; jj++) if (pool[jj] == ps) { rep = jj; break; }
                    if (rep == (uint32_t)tid) PLS_ATOMIC_OR(&misc[96 + (tid >> 5)], 1u << (tid & 31));
                }
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            const uint32_t total = misc[32 + NC];
            if ((uint32_t)tid < total) {
                int k = 0;
                PLS_UNROLL
                for (int q = 1; q < NC; q++) k += (uint32_t)tid >= misc[32 + q] ? 1 : 0;
                const uint32_t b0 = misc[32 + k], i = (uint32_t)tid - b0, cq = q0 + (uint32_t)k, u = cq / bpp, c = cq % bpp;
                const uint32_t sg0 = u * E, nsg = seg_umin(E, nseg - sg0);
                if (nsg > 1u) {
                    const uint32_t ps = pool[tid];
                    uint32_t dB = SEG_INVALID;
                    if (ps != 0xFFFFFFFFu) {
                        uint32_t rep = (uint32_t)tid;
                        for (uint32_t jj = b0; jj < (uint32_t)tid; jj++) if (pool[jj] == ps) { rep = jj; break; }
                        dB = seg_bits_count(misc + 96, b0, rep);
                        if (rep == (uint32_t)tid) pool2[b0 + dB] = ps;
                    }
                    j.rout[(((size_t)f * nseg + sg0) * 4 + c) * SEG_NSP + i] = (uint16_t)dB;
                }
                if (i == 0u) misc[64 + k] = nsg > 1u ? seg_bits_count(misc + 96, b0, misc[32 + k + 1]) : 0u;
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            if (tid == 0) { uint32_t run = 0; for (int k = 0; k < NC; k++) { const uint32_t cnt2 = misc[64 + k]; misc[64 + k] = run; run += cnt2; } misc[64 + NC] = run; }
        }
        PLS_SYNC();
        /* -- third phase: the second list through the rest of the unit -- */
        PLS_THREADS(tid, NT) {
            const uint32_t total2 = misc[64 + NC];
            if ((uint32_t)tid < total2) {
                int k = 0;
                PLS_UNROLL
                for (int q = 1; q < NC; q++) k += (uint32_t)tid >= misc[64 + q] ? 1 : 0;
                const uint32_t i = (uint32_t)tid - misc[64 + k], cq = q0 + (uint32_t)k, u = cq / bpp, c = cq % bpp;
                SegState st = seg_state_unpack(pool2[misc[32 + k] + i]);
                const SegPix *pk = px + (size_t)k * NPX;
                const uint32_t sg0 = u * E, nsg = seg_umin(E, nseg - sg0);
                int bad = 0;
                for (uint32_t sl = 1; sl < nsg; sl++) {
                    const size_t slot = (((size_t)f * nseg + sg0 + sl) * 4 + c) * SEG_NSP + i;
                    for (int part = 0; part < SEG_PARTS; part++) {
                        if (part) j.rck[slot * (SEG_PARTS - 1) + (part - 1)] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                        bad |= seg_run_fast_f(f, trx, pk + 1 + sl * SEG_L + part * SEG_PL, 1, SEG_PL, st, SEG_LDS_CU32(tw), SEG_LDS_CU8(tw + 4 * SEG_TN), G, SEG_LDS_CU32(lut));
                    }
                    j.rst[slot] = bad ? 0xFFFFFFFFu : seg_state_pack(st);
                    SEG_DEBUG_STATE(f, (long)cv.y * 100000 + (long)cq, sl, i, bad ? 0xFFFFFFFFu : seg_state_pack(st));
                    if (sl + 1 == nsg) j.rout[slot] = (uint16_t)(bad ? (uint32_t)SEG_INVALID : seg_any_encode(P, f, pk[(sl + 1) * SEG_L], st));
                }
            }
        }
    }
    if (prof) {
        PLS_SYNC();
        PLS_THREADS(tid, NT) {
            if (tid == 0) {
                te[4] = PLS_CLOCK();
                for (int q = 0; q < 4; q++) { PLS_ATOMIC_MAX(&j.result[24 + q], (int32_t)(te[q + 1] - te[q])); PLS_ATOMIC_ADD((uint32_t *)&j.result[28 + q], (uint32_t)(te[q + 1] - te[q])); }
                PLS_ATOMIC_ADD((uint32_t *)&j.result[32], 1u);
                PLS_ATOMIC_ADD((uint32_t *)&j.result[33], 4u * misc[32 + NC] / (uint32_t)NC);
            }
        }
    }
}

/* Does candidate f's enumeration of this attempt start its units FROM SEEDS (seg_enum_unit_body<.., SEEDS = true>) -- when the launcher offers it (seeds != 0: a batch
 * composed in units whose (strength, bleed) pair has a seed set) -- or from every state?  From every state (1) in an EPOCH: a row that was broken off because its true state
 * sits in a cycle the seeds do not reach (a periodic pattern, a flat region) would be broken off again at the next unit, and again -- two attempts a unit; the exhaustive
 * start finishes the row in one go; and (2) for the rest of an image whose rows keep breaking (more than one row in sixteen, and eight to begin with): flat, few-coloured
 * content is full of such fixed points (oracle/seed_study.c on the suite's dice / tux: one boundary in a hundred).  Uniform over the workgroup, deterministic (nbreak is
 * written by the chain kernels of earlier attempts only), and either way the validation is the ground truth. */
PLS_HD bool seg_unit_from_seeds(const SegJob &j, const SegParams &P, const SegCtlView &cv, int f, int seeds)
{
    return seeds != 0 && P.seed_n > 0 && !seg_is_small(P, f) && cv.start_x == 0u && j.nbreak * 16u <= cv.y + 128u;
}

/* frozen histogram of candidate f in this attempt: H0 + base[f] */
PLS_HD void seg_load_frozen(const SegJob &j, int par, int f, uint32_t *Hf, uint32_t *rank, int tid, int nt)
{
    for (int b = tid; b < 256; b += nt) {
        Hf[b] = j.H0[par * 256 + b] + j.base[((size_t)par * SEG_NFILT + f) * 256 + b];
        rank[b] = j.orig_rank[f * 256 + b];
    }
}

/* bump counters of ONE segment, four bins a word (a segment has at most SEG_L * 4 = 128 decisions: a byte holds its count of any bin) -- a quarter of the
 * shared memory of a word per bin, which is what lets a CU hold three replay workgroups instead of two */
PLS_HD void seg_cnt_add(uint32_t *cnt, int bin, uint32_t v) { PLS_ATOMIC_ADD(&cnt[bin >> 2], v << (8 * (bin & 3))); }
PLS_HD uint32_t seg_cnt_get(const uint32_t *cnt, int bin) { return (cnt[bin >> 2] >> (8 * (bin & 3))) & 255u; }
static_assert(SEG_L * 4 <= 255, "a segment's bump count of a bin fits a byte");
/* pixels [xa, xe) of one channel from state st: the table steps, and if a lane leaves what the tables cover, once more by scanning.
 * out: candidate words (stride 4 words per pixel) or null; cnt: the segment's bump counters (seg_cnt_add: 64 words) or null. */
PLS_HD void seg_walk(int f, const SegPix *px, int pstride, uint32_t xa, uint32_t xe, SegState &st, seg_lds_cu32 tw, seg_lds_cu32 lut, const uint32_t *Hf,
                     const uint32_t *rank, const SegGeo &G, const uint32_t *lut_g, int bleed, uint32_t *out, uint32_t *cnt)
{
    const SegState st0 = st;
    int bad = 0;
    const seg_lds_cu8 cls = SEG_LDS_CU8(tw + 4 * SEG_TN);
    for (uint32_t x = xa; x < xe; x++) {
        uint32_t w;
        const SegPix &p = px[(size_t)(x - xa) * pstride];
        switch (f) {
        case 1: w = seg_step_fast<1, true>(p, st, bad, tw, cls, G, lut); break;
        case 2: w = seg_step_fast<2, true>(p, st, bad, tw, cls, G, lut); break;
        case 3: w = seg_step_fast<3, true>(p, st, bad, tw, cls, G, lut); break;
        case 4: w = seg_step_fast<4, true>(p, st, bad, tw, cls, G, lut); break;
        default: w = seg_step_fast<0, true>(p, st, bad, tw, cls, G, lut); break;
        }
        if (out) out[(size_t)(x - xa) * 4] = w;
        if (cnt) seg_cnt_add(cnt, seg_cand_bin(w), 1u);
    }
    if (seg_bad(bad)) {
        /* (rare) take the bumps back and do the range again by scanning */
        if (cnt && out) for (uint32_t x = xa; x < xe; x++) seg_cnt_add(cnt, seg_cand_bin(out[(size_t)(x - xa) * 4]), 0u - 1u);
        st = st0;
        for (uint32_t x = xa; x < xe; x++) {
            const uint32_t w = seg_step_scan(f, px[(size_t)(x - xa) * pstride], st, Hf, nullptr, rank, G, lut_g, bleed);
            if (out) out[(size_t)(x - xa) * 4] = w;
            if (cnt) seg_cnt_add(cnt, seg_cand_bin(w), 1u);
        }
    }
}

/* ---- FIRST SEGMENT: task (f): the epoch's first (partial) segment has a known entry state, so it is not enumerated but walked,
 * lane = channel, next to the enumeration workgroups (same kernel); out: the index the chain starts from ------------------------ */
template <int NT, bool UNITS>
PLS_HD void seg_first_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, unsigned char *smem)
{
    const SegCtl &ctl = j.ctl[par];
    if (cv.finished || cv.active != 1) return;
    const uint32_t W = j.W, bpp = j.bpp, nseg = j.nseg;
    const uint32_t sx = cv.start_x;
    if (sx >= W || sx == 0) return;                           /* a fresh row needs no walk: its segment 0 is enumerated */
    const uint32_t first = sx / SEG_L;
    if (first + 1 >= nseg) return;                            /* no segment behind it */
    /* enumeration in units (SegParams::unit > 1): the walk goes on to the end of the UNIT that holds the epoch's first pixel -- the enumeration starts
     * with the unit behind it -- and leaves the entry state of every segment it crosses (the replay walks those from there: no checkpoints) */
    const uint32_t E = UNITS ? seg_unit_of(P, f) : 1u, ulast = seg_umin((first / E) * E + E - 1u, nseg - 1u);   /* (!UNITS: the per-segment enumeration's kernels, where E = 1 folds all of this away) */
    const uint32_t wend = seg_umin(ulast, nseg - 2u);         /* last segment walked: the row's last segment has nothing behind it */
    const uint32_t npix = (wend - first + 1u) * SEG_L;
    uint32_t *tw = (uint32_t *)smem;
    uint32_t *lut = tw + SEG_TBL_WORDS;
    uint32_t *Hf = lut + 512, *rank = Hf + 256;
    SegPix *px = (SegPix *)(rank + 256);                      /* [npix][4]: up to a unit's pixels */
    const unsigned long long tf0 = (P.engine_flags & 1) ? PLS_CLOCK() : 0ull;
    const uint32_t y = cv.y;
    const uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    PLS_THREADS(tid, NT) {
        constexpr int NTW = (SEG_TBL_WORDS + NT - 1) / NT;
        uint32_t vt[NTW], vl = 0, vh = 0, vb = 0, vr = 0;
        SegPixRaw vp = seg_pix_raw_zero();
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; vt[q] = i < SEG_TBL_WORDS ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (tid < 512) vl = P.lut_a[tid];
        if (tid < 256) { const int b = tid; vh = j.H0[par * 256 + b]; vb = j.base[((size_t)par * SEG_NFILT + f) * 256 + b]; vr = j.orig_rank[f * 256 + b]; }
        static_assert(256 + SEG_UNIT * SEG_L <= NT, "a unit's pixels are staged by the threads behind the first 256");
        if (tid >= 256 && (uint32_t)(tid - 256) < npix) vp = seg_pix_fetch(row, nab, e0g, first * SEG_L + (uint32_t)(tid - 256), W);
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * NT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
        if (tid < 512) lut[tid] = vl;
        if (tid < 256) { Hf[tid] = vh + vb; rank[tid] = vr; }
        if (tid >= 256 && (uint32_t)(tid - 256) < npix) seg_pix_split4(px + (tid - 256) * 4, vp, bpp, first * SEG_L + (uint32_t)(tid - 256), W);
    }
    PLS_SYNC();
    PLS_THREADS(tid, NT) {
        if (tid < 4 && (uint32_t)tid < bpp) {
            const int c = tid;
            SegState st = seg_state_unpack(ctl.state[f][c]);
            for (uint32_t sg = first; sg <= wend; sg++) {
                const uint32_t xa = sg == first ? sx : sg * SEG_L, xe = (sg + 1u) * SEG_L;         /* xe < W: there is a segment behind */
                seg_walk(f, px + (xa - first * SEG_L) * 4 + c, 4, xa, xe, st, SEG_LDS_CU32(tw), SEG_LDS_CU32(lut), Hf, rank, G, lut, P.bleed, nullptr, nullptr);
                if (sg + 1u <= ulast) {
                    /* the next segment belongs to the walked unit: its entry state for the replay (no dense id: no checkpoints) */
                    j.entry[((size_t)f * nseg + sg + 1u) * 4 + c] = seg_state_pack(st);
                    j.dnout[((size_t)f * nseg + sg + 1u) * 4 + c] = (uint16_t)SEG_INVALID;
                }
            }
            /* (when the walked unit is the row's last there is no enumerated unit behind it and the chain does not read this) */
            const uint32_t idx = seg_any_encode(P, f, px[(npix - 1u) * 4 + c], st);
            j.firstidx[(f * 4 + c) * 2] = idx;
            j.firstidx[(f * 4 + c) * 2 + 1] = seg_state_pack(st);
            if ((P.engine_flags & 1) && tid == 0) { const unsigned long long t1 = PLS_CLOCK(); PLS_ATOMIC_MAX(&j.result[34], (int32_t)(t1 - tf0)); PLS_ATOMIC_ADD((uint32_t *)&j.result[35], (uint32_t)(t1 - tf0)); PLS_ATOMIC_ADD((uint32_t *)&j.result[36], 1u); }
        }
    }
}

/* ---- GATHER (seeded sets): task (f, c, block of SEG_GS segments) -- the dense transitions, looked up by many workgroups ------------
 * For every dense id d of segment sg: the id, in segment sg + 1's entry set, of d's exit state (rst -> key -> two 16-byte loads of the entry hash),
 * left under rout[sg][d] (SEG_INVALID: no such id, no exit state, or the next segment's entry set does not hold it).  This was the head of the chain kernel:
 * 32 k lookups of two dependent loads on each of its 20 workgroups, 15 us of its 22 (profiles/r05_seeded_runin.txt); on 1280 workgroups of
 * 256 threads it is one round of loads, and the chain reads two bytes per transition instead. */
#ifndef SEG_GS
#define SEG_GS 4
#endif
#define SEG_GT 256
static_assert(SEG_GT == SEG_NSP && SEG_NSP % 8 == 0, "one lane of the gather kernel per dense id; the chain reads a table row in pieces of eight ids");
PLS_HD void seg_gather_seeded_body(const SegJob &j, const SegCtlView &cv, int f, int c, int blk)
{
    if (cv.finished || cv.active != 1 || (uint32_t)c >= j.bpp) return;
    const uint32_t W = j.W, nseg = j.nseg, sx = cv.start_x;
    if (sx >= W) return;
    const uint32_t first = sx / SEG_L;
    if (first + 1 >= nseg) return;
    /* transitions sg -> sg + 1 of the enumerated segments (behind a walked first segment: from the one after it), this block's share */
    const uint32_t s0 = sx ? first + 1 : 0u, b0 = (uint32_t)blk * SEG_GS;
    const uint32_t lo = s0 > b0 ? s0 : b0, hi = seg_umin(nseg - 1u, b0 + SEG_GS);
    if (lo >= hi) return;
    const SEG_AS_GLB uint32_t *ehash = j.ehash + ((size_t)f * nseg * 4 + c) * SEG_EH_WORDS;    /* + sg * 4 * SEG_EH_WORDS */
    SEG_AS_GLB uint16_t *rout = j.rout + ((size_t)f * nseg * 4 + c) * SEG_NSP;                 /* + sg * 4 * SEG_NSP */
    const SEG_AS_GLB uint32_t *rst = j.rst + ((size_t)f * nseg * 4 + c) * SEG_NSP;
    const SEG_AS_GLB uint32_t *dcnt = j.dcnt + (size_t)f * nseg * 4 + c;                       /* + sg * 4 */
    const uint32_t rstep32 = 4u * SEG_NSP, estep32 = 4u * SEG_EH_WORDS;
    PLS_THREADS(tid, SEG_GT) {
        const uint32_t d = (uint32_t)tid;                       /* (SEG_GT = SEG_NSP: one lane per dense id, its SEG_GS segments in flight together) */
        uint32_t ps[SEG_GS], key[SEG_GS];
        bool live[SEG_GS];                                      /* (a lane beyond the segment's distinct states writes SEG_INVALID: the chain reads whole rows) */
        SegVec16 w0[SEG_GS], w1[SEG_GS];
        PLS_UNROLL
        for (int q = 0; q < SEG_GS; q++) {
            const uint32_t sg = seg_umin(lo + (uint32_t)q, hi - 1u);
            live[q] = lo + (uint32_t)q < hi && d < dcnt[sg * 4u];
            ps[q] = live[q] ? rst[sg * rstep32 + d] : 0xFFFFFFFFu;
        }
        PLS_UNROLL
        for (int q = 0; q < SEG_GS; q++) {
            const uint32_t sg = seg_umin(lo + (uint32_t)q, hi - 1u);
            key[q] = seg_eh_key_of_packed(ps[q]);
            const SEG_AS_GLB SegVec16 *w = (const SEG_AS_GLB SegVec16 *)(ehash + (sg + 1u) * estep32 + (key[q] == SEG_NOKEY ? 0u : seg_eh_base(key[q])));
            if (live[q]) { w0[q] = w[0]; w1[q] = w[1]; } else { w0[q] = SegVec16{ 0, 0, 0, 0 }; w1[q] = w0[q]; }
        }
        PLS_UNROLL
        for (int q = 0; q < SEG_GS; q++) {
            const uint32_t sg = lo + (uint32_t)q;
            if (sg < hi) rout[sg * rstep32 + d] = (uint16_t)(live[q] ? seg_eh_match(key[q], w0[q], w1[q]) : (uint32_t)SEG_INVALID);
        }
    }
}

/* ---- CHAIN: task (f, c): compose the segments from the epoch's start state ----------------------------------------------
 * The enumeration left, per segment: a way from an entry STATE to its dense id (exhaustive sets: maps, by entry index -- the exit index of
 * segment k IS the entry index of segment k+1; seeded sets: ehash, by value), dense id -> exit index (rout) and exit state (rst).  The chain
 * only needs the DENSE transition tables  T_k[d] = id in segment k+1 of the exit of id d of segment k: a few dozen entries per segment
 * whatever the number of chain states, gathered here in parallel.  Composed in blocks of SEG_CBLK segments (every block's composed table for
 * all ids in parallel, then every block walks from the start id across the composed tables to its own head and through its segments):
 * 16 + nblk + 16 dependent lookups instead of nseg.  The entry state of segment k+1 is the exit state rst_k[d_k]: nothing to decode.
 *
 * The tables hold, instead of the next segment's id d', the INDEX of that id's entry in the next table, ((k + 1) << sh) + d': a step of
 * the composition is one load feeding the next load's address -- no compare, no select, no shift-and-add on the dependent path.  "No
 * successor" is the index of a cell that contains itself (row ntr + 1, which no segment owns).
 *
 * A row is taken in PASSES of as many segments as the tables have rows at the stride in use (64 entries per segment for nearly every row of
 * an exhaustive set: 17 distinct states per segment on average; 128 or 256 for seeded sets and the odd wide segment): rows of any width, and
 * a pass can start anywhere -- which is how the rare segment whose entry state the enumeration did not cover is handled: the pass ends in
 * front of it, ONE lane walks it step by step from its known entry state (table steps; exact), looks the exit up in the next segment's
 * entry set, and the passes go on from there. */
#define SEG_CBLK 16
#define SEG_CR_SH 6
#define SEG_CR_MAX (1 << SEG_CR_SH)                             /* the usual table stride; the exit states are staged in shared memory at this stride */
#define SEG_NOSTATE 0xFFFFFFFFu
#define SEG_CQ 8                                                /* gather items in flight per thread */
#define SEG_CHAIN_CAP 256                                       /* most transitions of a pass (strides 64 and 128) */
#define SEG_CHAIN_CAP8 224                                      /* ... at stride 256 */
#define SEG_CHAIN_POS (SEG_CHAIN_CAP + 32)                      /* dn / entL entries */
#define SEG_CHAIN_TROWS(n) (((n) < SEG_CHAIN_CAP8 ? (n) : SEG_CHAIN_CAP8) + SEG_CBLK + 2)
#define SEG_CHAIN_TBYTES(n) ((size_t)SEG_CHAIN_TROWS(n) * 512)     /* T (and R behind it at the usual stride) for a row of n segments: the widest stride sets the size */
#define SEG_CHAIN_GWORDS ((SEG_CHAIN_CAP / SEG_CBLK + 2) * 256 / 2)
#define SEG_SM_CHAIN(nseg) ((1024 + 32 + 2 * SEG_CHAIN_POS + SEG_CHAIN_GWORDS) * 4 + SEG_CHAIN_TBYTES(nseg) + SEG_TBL_WORDS * 4 + (SEG_L + 1) * 8 + 64)
/* (exhaustive state sets: no repair path, so no tables and pixel records behind T; npos = the row's chain positions: units) */
#define SEG_SM_CHAIN_X(npos) ((1024 + 32 + 2 * SEG_CHAIN_POS + SEG_CHAIN_GWORDS) * 4 + SEG_CHAIN_TBYTES(npos) + 64)
PLS_HD uint32_t seg_chain_cap(uint32_t sh) { return sh >= 8 ? (uint32_t)SEG_CHAIN_CAP8 : (uint32_t)SEG_CHAIN_CAP; }
template <bool SEEDED, int CT, bool UNITS>
PLS_HD void seg_chain_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int c, unsigned char *smem)
{
    if (cv.finished || cv.active != 1 || (uint32_t)c >= j.bpp) return;
    const uint32_t W = j.W, bpp = j.bpp, nseg = j.nseg;
    const uint32_t sx = cv.start_x;
    if (sx >= W) return;
    const uint32_t first = sx / SEG_L;
    SEG_AS_GLB uint16_t *dnout = j.dnout + (size_t)f * nseg * 4 + c;                           /* + sg * 4 */
    if (sx || nseg == 1) { PLS_THREADS(tid, CT) { if (tid == 0) dnout[(size_t)first * 4] = (uint16_t)SEG_INVALID; } }   /* a walked first segment has no checkpoints */
    if (first + 1 >= nseg) return;
    constexpr bool seeded = SEEDED;                             /* (= P.seeded: two kernels, so that the exhaustive sets' gather does not carry the seeded one's registers) */
    /* The chain composes UNITS: E = SegParams::unit segments enumerated as one (seg_enum_unit_body; 1: every segment on its own, and always for seeded
     * sets).  Unit u = segments u * E .. : its entry map and distinct-state count sit at its FIRST segment (SEGF), its exit index and exit state at
     * its LAST (SEGL); every segment of it carries the unit's dense id, the entry state of an inner segment is the exit state of the one in front. */
    /* (UNITS: a kernel of its own -- with E = 1 a constant, the one-image chain pays nothing for the units: no load of SegParams::unit at its head, no division) */
    const uint32_t E = (seeded || !UNITS) ? 1u : seg_unit_of(P, f), nunit = (nseg + E - 1u) / E, ufirst = first / E;
#define SEGF(u) (E == 1u ? (u) : (u) * E)
#define SEGL(u) (E == 1u ? (u) : seg_umin((u) * E + E, nseg) - 1u)
    if (sx && ufirst + 1 >= nunit) return;                      /* no unit behind the walked one (seg_first_body left the entry states of its segments) */
    /* enumerated units s0 .. nunit-1 (ne of them; a fresh row: from 0) = POSITIONS 0 .. ns of the chain; ns = ne - 1 transitions */
    const uint32_t s0 = sx ? ufirst + 1 : 0u, ne = nunit - s0, ns = ne - 1;
    seg_lds_u32 Hf = (seg_lds_u32)smem, rank = Hf + 256, lut = Hf + 512;   /* (repair only) */
    seg_lds_u32 idxb = Hf + 1024;                               /* [32]: [24] repairs, [25] first position without an id, [26] entry state of the pass's first position, [27] most distinct states of a segment,
                                                                   [28] dense id the pass starts with, [30] some segment has more distinct states than the stride, [31] repair tables loaded */
    seg_lds_u32 dn = idxb + 32;                                /* [SEG_CHAIN_POS] dense id at every position of the pass */
    seg_lds_u32 entL = dn + SEG_CHAIN_POS;                     /* [SEG_CHAIN_POS] entry state at every position of the pass (from 1; 0: idxb[26]) */
    seg_lds_u16 G = (seg_lds_u16)(entL + SEG_CHAIN_POS);       /* [nblk + 1][stride] composed tables of the blocks */
    seg_lds_u16 T = G + 2 * SEG_CHAIN_GWORDS;                  /* [rows][stride]: T[k] takes a dense id of position k to one of position k+1 */
    seg_lds_u32 twr = (seg_lds_u32)((SEG_AS_LDS unsigned char *)T + SEG_CHAIN_TBYTES(nseg));   /* (repair only: seeded sets, whose launch asks for SEG_SM_CHAIN) the candidate's decision tables, then the walked segment's pixel records */
    const uint32_t y = cv.y;
    const SEG_AS_GLB uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SEG_AS_GLB uint16_t *maps = j.maps + ((size_t)f * nseg * 4 + c) * (size_t)P.nsp;     /* + sg * 4 * nsp */
    const SEG_AS_GLB uint32_t *ehash = j.ehash + ((size_t)f * nseg * 4 + c) * SEG_EH_WORDS;    /* + sg * 4 * SEG_EH_WORDS */
    const SEG_AS_GLB uint16_t *rout = j.rout + ((size_t)f * nseg * 4 + c) * SEG_NSP;           /* + sg * 4 * SEG_NSP */
    const SEG_AS_GLB uint32_t *rst = j.rst + ((size_t)f * nseg * 4 + c) * SEG_NSP;
    const SEG_AS_GLB uint32_t *dcnt = j.dcnt + (size_t)f * nseg * 4 + c;                       /* + sg * 4 */
    SEG_AS_GLB uint32_t *entry = j.entry + (size_t)f * nseg * 4 + c;                           /* + sg * 4 */
    const uint32_t mstep32 = 4u * (uint32_t)P.nsp, rstep32 = 4u * SEG_NSP, estep32 = 4u * SEG_EH_WORDS;
    const int nstates = P.ns;
    const uint32_t eflags = (uint32_t)P.engine_flags;
    const SegState start0 = { 0, 0, 0 };
    const bool prof = (eflags & 1) != 0;
    unsigned long long tacc[4] = { 0, 0, 0, 0 }, tc[5] = { 0, 0, 0, 0, 0 };
    PLS_THREADS(tid, CT) { if (tid < 32) idxb[tid] = 0u; }
    PLS_SYNC();
    if (ns == 0) {
        /* ONE enumerated unit (the row's last, behind a walked one -- or the whole of a short row): nothing to compose, its id is the start state's */
        PLS_THREADS(tid, CT) {
            if (tid == 0) {
                const uint32_t start_ps = sx ? j.firstidx[(f * 4 + c) * 2 + 1] : seg_state_pack(start0);
                uint32_t d = SEG_INVALID;
                if (seeded) d = seg_eh_lookup(ehash + (size_t)SEGF(s0) * estep32, seg_eh_key_of_packed(start_ps));
                else {
                    const uint32_t idx = sx ? j.firstidx[(f * 4 + c) * 2] : (seg_is_small(P, f) ? P.idx0_small : P.idx0_big);
                    if (idx != SEG_INVALID && (int)idx < nstates) d = (uint32_t)maps[(size_t)SEGF(s0) * mstep32 + idx];
                }
                const uint32_t dc = dcnt[(size_t)SEGF(s0) * 4];
                if (d >= SEG_NSP || d >= dc) d = SEG_INVALID;
                entry[(size_t)SEGF(s0) * 4] = start_ps;
                dnout[(size_t)SEGF(s0) * 4] = (uint16_t)d;
                if (SEGL(s0) > SEGF(s0)) {
                    /* (a unit: behind its first segment its states have the ids of the second list, rout of the first segment translates) */
                    const uint32_t dB = d != SEG_INVALID ? (uint32_t)rout[(size_t)SEGF(s0) * rstep32 + d] : (uint32_t)SEG_INVALID;
                    const bool okB = dB < SEG_NSP;
                    if (d != SEG_INVALID) entry[(size_t)(SEGF(s0) + 1u) * 4] = rst[(size_t)SEGF(s0) * rstep32 + d];
                    for (uint32_t sg = SEGF(s0) + 1u; sg <= SEGL(s0); sg++) {
                        dnout[(size_t)sg * 4] = (uint16_t)(okB ? dB : (uint32_t)SEG_INVALID);
                        if (sg > SEGF(s0) + 1u && okB) entry[(size_t)sg * 4] = rst[(size_t)(sg - 1u) * rstep32 + dB];
                    }
                }
                /* without an id the unit's first segment is walked by the replay from its entry state; what lies behind it in the unit has no state
                 * to start from: reported as this candidate's first failed decision (an epoch starts there, as behind any failed validation) */
                const uint32_t xfail = (SEGF(s0) + 1u) * SEG_L;
                if (d == SEG_INVALID && SEGL(s0) > SEGF(s0) && xfail < W) {
                    PLS_ATOMIC_MIN(&j.acc[par].fail[f], xfail * 4u + (uint32_t)c);
                    PLS_ATOMIC_OR(&j.acc[par].failmask, 1u << f);
                    PLS_ATOMIC_OR(&j.self->vfail[par], 1u << f);
                }
            }
        }
        return;
    }
    uint32_t sh = seeded ? SEG_CR_SH + 1 : SEG_CR_SH;
    if (eflags & 4) sh++;                                       /* (test hook: a wider stride than needed) */
    uint32_t a = 0;                                             /* first position of the pass */
    bool first_iter = true;
    uint32_t nwide = 0;
    for (;;) {
        const uint32_t ntr = seg_umin(seg_chain_cap(sh), ns - a);
        if (ntr == 0) {
            /* only the row's last segment is left (behind a repair): it has the id the repair looked up */
            PLS_THREADS(tid, CT) { if (tid == 0) { const uint32_t d = idxb[28]; dnout[(size_t)SEGF(s0 + a) * 4] = (uint16_t)(d < SEG_NSP ? d : SEG_INVALID); } }   /* (seeded sets: E = 1) */
            break;
        }
        if (prof) tc[0] = PLS_CLOCK();
        const uint32_t stride = 1u << sh, npos = ntr + 1u, nblk = (npos + SEG_CBLK - 1) / SEG_CBLK;
        const bool useR = sh == SEG_CR_SH;
        seg_lds_u32 R = (seg_lds_u32)(T + (((size_t)nblk * SEG_CBLK + 2) << SEG_CR_SH));   /* [ntr][64] (useR): exit state of position k under that id; behind T's rows at the usual stride */
        const uint32_t dummy = (ntr + 1u) << sh;
        /* -- gather: T[k][d] = index of the successor's cell and R[k][d] = the exit state itself (= entry state of position k+1); per item the
         *    loads that need nothing first, then the dependent ones; SEG_CQ items per thread at a time -- */
        PLS_THREADS(tid, CT) {
            /* (one lane) the state and the id the chain starts from: its loads ride along with the gather's, one level each */
            const bool starter = first_iter && tid == CT - 1;
            uint32_t fi0 = 0, fi1 = 0, dfirst = SEG_INVALID;
            if (starter && sx) { fi0 = j.firstidx[(f * 4 + c) * 2]; fi1 = j.firstidx[(f * 4 + c) * 2 + 1]; }
            const uint32_t start_ps = sx ? fi1 : seg_state_pack(start0);
            const uint32_t idx_first = sx ? fi0 : (seg_is_small(P, f) ? P.idx0_small : P.idx0_big);
            const uint32_t start_key = seg_eh_key_of_packed(start_ps), start_base = seg_eh_base(start_key);
            /* item (k, d): k = position of the pass, d = dense id; this thread's items share d and step through k by kstep.  32-bit offsets
             * from uniform bases (the whole gather is bound by the instructions 1024 threads issue on one CU, not by memory) */
            const uint32_t d = (uint32_t)tid & (stride - 1), k0 = (uint32_t)tid >> sh, kstep = (uint32_t)CT >> sh;
            uint32_t widest = 0;
            if (seeded) {
                /* seeded sets: the gather kernel (seg_gather_seeded_body) left the successor's id of every dense id under rout, SEG_INVALID where there is
                 * none: eight ids a load, turned into the index of the successor's cell and stored eight at a time */
                const uint32_t per = stride >> 3;                   /* 16-byte pieces of a table row */
                for (uint32_t t = (uint32_t)tid; t < ntr * per; t += (uint32_t)CT) {
                    const uint32_t k = t / per, part = t - k * per, sg = s0 + a + k;
                    const SegVec16 in = *(const SEG_AS_GLB SegVec16 *)(rout + sg * rstep32 + part * 8u);
                    const uint32_t nb = (k + 1u) << sh;
                    uint32_t w[4] = { in.a, in.b, in.c, in.d };
                    PLS_UNROLL
                    for (int q = 0; q < 4; q++) {
                        const uint32_t lo = w[q] & 0xFFFFu, hi = w[q] >> 16;
                        w[q] = (lo < SEG_NSP ? nb + lo : dummy) | ((hi < SEG_NSP ? nb + hi : dummy) << 16);
                    }
                    *(SEG_AS_LDS SegVec16 *)(T + (k << sh) + part * 8u) = SegVec16{ w[0], w[1], w[2], w[3] };
                }
                for (uint32_t k = (uint32_t)tid; k < ntr; k += (uint32_t)CT) { const uint32_t dc = dcnt[(s0 + a + k) * 4u]; widest = dc > widest ? dc : widest; }
                if (starter) {
                    const SEG_AS_GLB SegVec16 *w = (const SEG_AS_GLB SegVec16 *)(ehash + (s0 + a) * estep32 + (start_key == SEG_NOKEY ? 0u : start_base));
                    dfirst = seg_eh_match(start_key, w[0], w[1]);
                }
            } else
            for (uint32_t kb = 0; kb < ntr; kb += SEG_CQ * kstep) {
                /* no branch per item: an item beyond the last transition is clamped onto it and does that one's work once more (same
                 * values to the same places) */
                uint32_t dcv[SEG_CQ], r[SEG_CQ], ps[SEG_CQ], v[SEG_CQ];
                PLS_UNROLL
                for (int q = 0; q < SEG_CQ; q++) {
                    const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u), sg = s0 + a + k;
                    const uint32_t o = SEGL(sg) * rstep32 + d;
                    dcv[q] = dcnt[SEGF(sg) * 4u];
                    if (UNITS && !seeded) { r[q] = (uint32_t)rout[SEGF(sg) * rstep32 + d]; ps[q] = SEG_NOSTATE; }     /* (units: first the id in the unit's second list) */
                    else {
                        r[q] = (uint32_t)rout[o];
                        ps[q] = useR ? rst[o] : SEG_NOSTATE;
                    }
                }
                if (UNITS && !seeded) {
                    /* the unit's exit index and exit state sit under the id of the SECOND list (a unit of one segment has no second list: rout is the exit index) */
                    uint32_t r2[SEG_CQ];
                    PLS_UNROLL
                    for (int q = 0; q < SEG_CQ; q++) {
                        const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u), sg = s0 + a + k;
                        const bool one = SEGL(sg) == SEGF(sg), okB = d < dcv[q] && r[q] < SEG_NSP;
                        const uint32_t o2 = SEGL(sg) * rstep32 + (one ? d : (okB ? r[q] : 0u));
                        r2[q] = (uint32_t)rout[o2];
                        ps[q] = useR ? rst[o2] : SEG_NOSTATE;
                    }
                    PLS_UNROLL
                    for (int q = 0; q < SEG_CQ; q++) {
                        const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u), sg = s0 + a + k;
                        const bool one = SEGL(sg) == SEGF(sg), okB = d < dcv[q] && r[q] < SEG_NSP;
                        if (!(one || okB)) { r2[q] = SEG_INVALID; ps[q] = SEG_NOSTATE; }
                        r[q] = r2[q];
                    }
                }
                if (starter && kb == 0) {
                    if (idx_first != SEG_INVALID && (int)idx_first < nstates) dfirst = (uint32_t)maps[SEGF(s0 + a) * mstep32 + idx_first];
                }
                /* (the dependent loads in a loop of their own, all of them requested before the first is used: next to their uses, each one
                 * waits for itself) */
                {
                    PLS_UNROLL
                    for (int q = 0; q < SEG_CQ; q++) {
                        const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u), sg = s0 + a + k;
                        const bool valid = d < dcv[q] && r[q] != SEG_INVALID && (int)r[q] < nstates;
                        v[q] = maps[SEGF(sg + 1u) * mstep32 + (valid ? r[q] : 0u)];       /* (unit sg + 1 <= nunit - 1 is enumerated: its row exists) */
                    }
                    PLS_UNROLL
                    for (int q = 0; q < SEG_CQ; q++) {
                        const bool valid = d < dcv[q] && r[q] != SEG_INVALID && (int)r[q] < nstates;
                        v[q] = valid ? v[q] : (uint32_t)SEG_INVALID;
                    }
                }
                PLS_UNROLL
                for (int q = 0; q < SEG_CQ; q++) {
                    const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u);
                    const bool valid = v[q] < SEG_NSP;                  /* (no id: the state left what the tables cover, or the entry set of the next segment does not hold it) */
                    v[q] = valid ? ((k + 1u) << sh) + v[q] : dummy;   /* (an id beyond the stride: the pass is gathered again, wider) */
                    ps[q] = (d < dcv[q]) ? ps[q] : SEG_NOSTATE;
                    widest = dcv[q] > widest ? dcv[q] : widest;
                }
                PLS_UNROLL
                for (int q = 0; q < SEG_CQ; q++) {
                    const uint32_t k = seg_umin(kb + k0 + (uint32_t)q * kstep, ntr - 1u);
                    const uint32_t t = (k << sh) + d;
                    T[t] = (uint16_t)v[q];
                    if (useR) R[t] = ps[q];
                }
            }
            if (widest > stride) { idxb[30] = 1u; PLS_ATOMIC_MAX_U(&idxb[27], widest); }
            /* rows behind the last table (the landing row, then spare rows up to whole blocks): they lead to the absorbing cell */
            for (uint32_t t = (ntr << sh) + (uint32_t)tid; t < (nblk * SEG_CBLK) << sh; t += CT) T[t] = (uint16_t)dummy;
            if (starter) {
                idxb[28] = dfirst;
                idxb[26] = start_ps;
                entry[(size_t)SEGF(s0) * 4] = start_ps;
            }
            if (tid == 0) { T[dummy] = (uint16_t)dummy; idxb[25] = 0xFFFFFFFFu; }
        }
        PLS_SYNC();
        if (idxb[30]) {
            /* some segment of the pass has more distinct states than the stride holds: once more, wider (its last dense ids would alias) */
            const uint32_t need = idxb[27];
            PLS_SYNC();
            PLS_THREADS(tid, CT) { if (tid == 0) { idxb[30] = 0u; idxb[27] = 0u; } }
            PLS_SYNC();
            while ((1u << sh) < need && sh < 8) sh++;
            nwide++;
            continue;
        }
        if (prof) tc[1] = PLS_CLOCK();
        const uint32_t gdummy = nblk << sh;                        /* G's absorbing cell: the row behind its last block */
        const uint32_t smask = stride - 1u;
        PLS_THREADS(tid, CT) {
            if (tid == 0) G[gdummy] = (uint16_t)gdummy;
            /* block b composes T[b*CBLK .. (b+1)*CBLK - 1]: the id at its first position -> the id at the next block's first position */
            for (uint32_t t = (uint32_t)tid; t + stride < (nblk << sh); t += CT) {
                const uint32_t b = t >> sh;
                uint32_t i = ((b * SEG_CBLK) << sh) + (t & smask);
                PLS_UNROLL
                for (int k = 0; k < SEG_CBLK; k++) i = (uint32_t)T[i];
                /* what it leads to, as the index of THAT id's entry in the next block's composed table (or G's own absorbing cell): the walk
                 * over the block heads is a chain of bare loads too */
                G[t] = (uint16_t)(i == dummy ? gdummy : (((b + 1u) << sh) | (i & smask)));
            }
        }
        PLS_SYNC();
        if (prof) tc[2] = PLS_CLOCK();
        /* -- walk: from position pos0 of the pass (0; behind a repair: the position behind the walked segment -- the tables gathered for the
         *    pass are still good, only the walk is done again from there), whose id is idxb[28] and whose entry state idxb[26] -- */
        uint32_t pos0 = 0;
        bool done = false, next_pass = false;
        for (;;) {
            const uint32_t bq = pos0 / SEG_CBLK, off = pos0 % SEG_CBLK;
            PLS_THREADS(tid, CT) {
                if ((uint32_t)tid >= bq && (uint32_t)tid < nblk) {
                    /* thread b: the true id at the head of block b (across the composed tables), then through the block.  Every lane walks the
                     * heads of ALL blocks (the same chain of loads in every lane: no lane-dependent loop) and keeps its own */
                    const uint32_t sid = idxb[28];
                    uint32_t g, b0 = bq;
                    if (off == 0u) g = sid < stride ? ((bq << sh) | sid) : gdummy;
                    else {
                        /* (behind a repair) first to the end of the block the walk starts inside: the same chain in every lane, lane bq keeps it */
                        uint32_t i = sid < stride ? ((pos0 << sh) | sid) : dummy;
                        for (uint32_t q = off; q < SEG_CBLK; q++) {
                            const uint32_t k = bq * SEG_CBLK + q;
                            if ((uint32_t)tid == bq && k < npos) dn[k] = i == dummy ? (uint32_t)SEG_INVALID : (i & smask);
                            i = (uint32_t)T[i];
                        }
                        b0 = bq + 1u;
                        g = i == dummy ? gdummy : ((b0 << sh) | (i & smask));
                    }
                    uint32_t gm = g;
                    for (uint32_t b = b0; b + 1 < nblk; b++) { g = (uint32_t)G[g]; gm = b + 1 == (uint32_t)tid ? g : gm; }
                    if ((uint32_t)tid >= b0) {
                        uint32_t i = gm == gdummy ? dummy : ((((uint32_t)tid * SEG_CBLK) << sh) | (gm & smask));
                        uint32_t at[SEG_CBLK];
                        PLS_UNROLL
                        for (int q = 0; q < SEG_CBLK; q++) { at[q] = i; i = (uint32_t)T[i]; }      /* (rows behind table ntr - 1 lead to the absorbing cell) */
                        PLS_UNROLL
                        for (int q = 0; q < SEG_CBLK; q++) {
                            const uint32_t k = (uint32_t)tid * SEG_CBLK + (uint32_t)q;
                            if (k < npos) dn[k] = at[q] == dummy ? (uint32_t)SEG_INVALID : (at[q] & smask);
                        }
                    }
                }
            }
            PLS_SYNC();
            PLS_THREADS(tid, CT) {
                /* ids out; entry state of position k+1 = exit state of position k under its id */
                for (uint32_t k = pos0 + (uint32_t)tid; k < npos; k += CT) {
                    uint32_t d = dn[k];
                    if ((eflags & 2) && k >= pos0 + 1u) d = SEG_INVALID;                  /* (test hook: every second segment through the repair) */
                    const uint32_t sg = s0 + a + k;
                    if (d == SEG_INVALID) { PLS_ATOMIC_MIN(&idxb[25], k); continue; }
                    dnout[(size_t)SEGF(sg) * 4] = (uint16_t)d;
                    uint32_t dL = d;                                  /* the id the unit's LAST segment keeps its records under */
                    if (E > 1u && SEGL(sg) > SEGF(sg)) {
                        /* the unit's inner segments: the id of the unit's second list (rout of its first segment translates), entry state = exit state of
                         * the segment in front */
                        const bool have = d < dcnt[(size_t)SEGF(sg) * 4];
                        const uint32_t dB = have ? (uint32_t)rout[(size_t)SEGF(sg) * rstep32 + d] : (uint32_t)SEG_INVALID;
                        const bool okB = dB < SEG_NSP;
                        entry[(size_t)(SEGF(sg) + 1u) * 4] = have ? rst[(size_t)SEGF(sg) * rstep32 + d] : SEG_NOSTATE;
                        for (uint32_t si = SEGF(sg) + 1u; si <= SEGL(sg); si++) {
                            dnout[(size_t)si * 4] = (uint16_t)(okB ? dB : (uint32_t)SEG_INVALID);
                            if (si > SEGF(sg) + 1u) entry[(size_t)si * 4] = okB ? rst[(size_t)(si - 1u) * rstep32 + dB] : SEG_NOSTATE;
                        }
                        dL = okB ? dB : (uint32_t)SEG_INVALID;
                    }
                    if (k >= ntr) continue;
                    uint32_t ps = SEG_NOSTATE;
                    if (useR) ps = R[(k << SEG_CR_SH) + d];
                    else if (d < dcnt[(size_t)SEGF(sg) * 4] && dL < SEG_NSP) ps = rst[(size_t)SEGL(sg) * rstep32 + dL];
                    entL[k + 1u] = ps;
                    if (ps != SEG_NOSTATE) entry[(size_t)SEGF(sg + 1u) * 4] = ps;
                }
            }
            PLS_SYNC();
            if (prof && pos0 == 0u) { tc[3] = PLS_CLOCK(); for (int q = 0; q < 3; q++) tacc[q] += tc[q + 1] - tc[q]; }
            uint32_t fb = idxb[25];
            if (fb == 0xFFFFFFFFu) {
                /* the rest of the pass has ids: the next one starts at its landing position */
                const uint32_t nd = dn[ntr], ne_ = entL[ntr];
                PLS_SYNC();
                a += ntr;
                if (a >= ns) { done = true; break; }                   /* (the landing position was the row's last segment: its id is out) */
                PLS_THREADS(tid, CT) { if (tid == 0) { idxb[28] = nd; idxb[26] = ne_; } }
                PLS_SYNC();
                next_pass = true;
                break;
            }
            if (!seeded) {
                /* Exhaustive state sets: a position without an id is next to never seen (a lane that left what the tables cover), and the code of
                 * the repair below costs this kernel 0.9 us on every row by merely being in it (registers, 100 more spilled scalars).  So this
                 * instantiation has none: the row BREAKS here.  The segment in question is left to the replay (which walks a segment without
                 * checkpoints from its entry state), and the pixel behind it is reported as this candidate's first failed decision: an epoch starts
                 * there, as behind any failed validation (seg_ctl_body) -- exact, and two attempts instead of 16 us on the rare row. */
                if (fb > pos0 && entL[fb] == SEG_NOSTATE) fb--;
                const uint32_t estb = fb > pos0 ? entL[fb] : idxb[26];
                PLS_SYNC();
                const uint32_t kqb = a + fb, sgb = SEGF(s0 + kqb);      /* (the unit's FIRST segment: the replay walks that one from the entry state; with units, what
                                                                            lies behind it in the unit has no state to start from either) */
                PLS_THREADS(tid, CT) {
                    if (tid == 0) {
                        SEG_DEBUG_COUNT(2, fb);
                        SEG_DEBUG_BREAK(f, c, sgb, estb, cv.y);
                        idxb[24]++;
                        PLS_ATOMIC_ADD(&j.self->nbreak, 1u);
                        if (j.break_word) PLS_HOST_VISIBLE_ADD(j.break_word, 1u);
                        dnout[(size_t)sgb * 4] = (uint16_t)SEG_INVALID; entry[(size_t)sgb * 4] = estb;
                        const uint32_t xfail = (sgb + 1u) * SEG_L;
                        if ((kqb < ns || E > 1u) && xfail < W) {
                            PLS_ATOMIC_MIN(&j.acc[par].fail[f], xfail * 4u + (uint32_t)c);
                            PLS_ATOMIC_OR(&j.acc[par].failmask, 1u << f);
                            PLS_ATOMIC_OR(&j.self->vfail[par], 1u << f);
                        }
                    }
                }
                done = true;
                break;
            }
            /* -- repair: position fb has no id (its entry state is not in the segment's entry set) -- or position fb - 1 has an id but no exit
             *    state (its lane left what the tables cover).  That segment is walked step by step from its entry state. -- */
            if (fb > pos0 && entL[fb] == SEG_NOSTATE) fb--;
            const uint32_t est = fb > pos0 ? entL[fb] : idxb[26];
            const bool have_tables = idxb[31] != 0u;
            PLS_SYNC();
            const uint32_t kq = a + fb, sgq = s0 + kq;                  /* the position and the segment walked */
            const uint32_t xq = sgq * SEG_L;
            SegPix *pxr = (SegPix *)(uint32_t *)(twr + SEG_TBL_WORDS);  /* [SEG_L + 1] */
            const SegGeo G_ = seg_geo((int)cv.s);
            PLS_THREADS(tid, CT) {
                if (!have_tables) {
                    for (int i = tid; i < SEG_TBL_WORDS; i += CT) twr[i] = j.tables[(size_t)f * SEG_TBL_WORDS + i];
                    if (tid < 256) seg_load_frozen(j, par, f, (uint32_t *)Hf, (uint32_t *)rank, tid, 256);
                    if (tid >= 256 && tid < 768) lut[tid - 256] = P.lut_a[tid - 256];
                }
                if (tid >= 768 && tid < 768 + SEG_L) pxr[tid - 768] = seg_pix_load(row, nab, e0g, bpp, seg_umin(xq + (uint32_t)(tid - 768), W - 1u), c);
                if (tid == 0) { SEG_DEBUG_COUNT(2, fb); SEG_DEBUG_REPAIR(f, c, sgq, est, idxb[28]); idxb[31] = 1u; idxb[24]++; idxb[25] = 0xFFFFFFFFu; dnout[(size_t)sgq * 4] = (uint16_t)SEG_INVALID; entry[(size_t)sgq * 4] = est; }
            }
            PLS_SYNC();
            if (kq >= ns) { done = true; break; }                      /* the row's last segment: the replay walks it from its entry state, nothing follows */
            PLS_THREADS(tid, CT) {
                if (tid == 0) {
                    SegState st = seg_state_unpack(est);
                    seg_walk(f, pxr, 1, xq, xq + SEG_L, st, SEG_LDS_CU32(twr), SEG_LDS_CU32(lut), (const uint32_t *)Hf, (const uint32_t *)rank, G_, (const uint32_t *)lut, P.bleed, nullptr, nullptr);
                    const uint32_t nps = seg_state_pack(st);
                    uint32_t nid = SEG_INVALID;
                    if (seeded) nid = seg_eh_lookup(ehash + (size_t)(sgq + 1u) * estep32, seg_eh_key(st.left, st.cn, st.th));
                    else {
                        const uint32_t idx = seg_any_encode(P, f, pxr[SEG_L - 1], st);
                        if (idx != SEG_INVALID && (int)idx < nstates) nid = (uint32_t)maps[(size_t)(sgq + 1u) * mstep32 + idx];
                    }
                    if (nid >= SEG_NSP || nid >= dcnt[(size_t)(sgq + 1u) * 4]) nid = SEG_INVALID;
                    entry[(size_t)(sgq + 1u) * 4] = nps;
                    idxb[26] = nps; idxb[28] = nid;
                }
            }
            PLS_SYNC();
            if (fb + 1u > ntr) { a = kq + 1u; next_pass = true; break; }   /* (the walked segment was the pass's landing position: the next pass starts behind it) */
            pos0 = fb + 1u;                                                /* the pass's tables hold what lies behind: walk on from there */
        }
        first_iter = false;
        if (done) break;
        (void)next_pass;
    }
    PLS_THREADS(tid, CT) {
        if (tid == 0) {
            if (idxb[24]) PLS_ATOMIC_ADD((uint32_t *)&j.result[17], idxb[24]);       /* segments walked step by step (reported with the image's result) */
            if (prof) {
                for (int q = 0; q < 3; q++) { PLS_ATOMIC_MAX(&j.result[8 + q], (int32_t)tacc[q]); PLS_ATOMIC_ADD((uint32_t *)&j.result[12 + q], (uint32_t)tacc[q]); }
                PLS_ATOMIC_ADD((uint32_t *)&j.result[16], 1u);
                PLS_ATOMIC_ADD((uint32_t *)&j.result[18], nwide);
            }
        }
    }
#undef SEGF
#undef SEGL
}

/* ---- REPLAY: task (f, grp): lane = (segment of the group, part of the segment, channel) -----------------------------------
 * A lane starts from a state it knows: part 0 from the segment's entry state, part p from the checkpoint the enumeration left for
 * the segment's dense id.  A part without a checkpoint is walked by the lane in front of it.  The walkers write into shared memory;
 * behind them every thread takes ONE pixel of the group: it stores the pixel's four candidate words (one 16-byte store) and adds the pixel
 * to the row's SUMS -- what the row decision (seg_decide_cand) wants from the pixels of a candidate row besides its bump counts: the
 * derivative error (optimize_state.c:265-287) and the sums of libpng's heuristic (:492-562).  Neither depends on a histogram.  (The
 * entropy cost, :326-342, needs no pixel at all: seg_entropy_costs.)  In a later epoch of the row the validated words in front of the epoch
 * are read back for the sums.  The new left neighbour of the group's FIRST pixel belongs to the workgroup in front: it is taken from the
 * entry state of the group's first segment (the chain kernel's), recorded in grpleft, and the validation checks the record against the
 * byte that was written.  Candidate none (f = 0) also gets the LOWER BOUND of its row cost here (see seg_none_reach), run or not. */
PLS_HD int seg_none_reach(const SegJob &j, const SegParams &P, int s, int M, int m);

/* ---- EXTREMES: one workgroup per image, next to the chain's (it has nothing to do with them: the launch has room): the largest and the
 * smallest orig + incoming error over the row's channels (a forced transparent alpha aside), which bound how far candidate none's bytes
 * can overshoot 0..255 (seg_none_reach); read by the replay's workgroups of candidate none in the next launch. */
template <int CT>
PLS_HD void seg_extremes_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, unsigned char *smem)
{
    if (cv.finished || !j.rowmm) return;
    const uint32_t W = j.W, bpp = j.bpp, y = cv.y;
    const uint32_t *row = seg_row_orig(j, y), *e0g = seg_e0(j, y);
    uint32_t *mm = (uint32_t *)smem;                            /* [0] max, [1] min, biased (unsigned compare) */
    PLS_THREADS(tid, CT) { if (tid == 0) { mm[0] = 0x80000000u ^ (uint32_t)(-(1 << 30)); mm[1] = 0x80000000u ^ (uint32_t)(1 << 30); } }
    PLS_SYNC();
    PLS_THREADS(tid, CT) {
        int vmax = -(1 << 30), vmin = 1 << 30;
        /* four pixels per thread and turn, every request in front of the first use (pixels beyond the row are clamped onto its last: they change nothing) */
        for (uint32_t x0 = (uint32_t)tid; x0 < W; x0 += 4u * CT) {
            uint32_t o[4], ea[4], eb[4];
            PLS_UNROLL
            for (int q = 0; q < 4; q++) { const uint32_t x = seg_umin(x0 + (uint32_t)q * CT, W - 1u); o[q] = row[x]; ea[q] = e0g[2 * (size_t)x]; eb[q] = e0g[2 * (size_t)x + 1]; }
            PLS_UNROLL
            for (int q = 0; q < 4; q++) {
                const uint32_t e[2] = { ea[q], eb[q] };
                const bool alpha0 = (bpp & 1u) == 0u && ((o[q] >> (8u * (bpp - 1u))) & 255u) == 0u;
                for (uint32_t c = 0; c < bpp; c++) {
                    if (alpha0 && c == bpp - 1u) continue;
                    const int v = (int)((o[q] >> (8 * c)) & 255u) + seg_err_plane(e, seg_plane_of_channel(bpp, (int)c));
                    vmax = seg_max(vmax, v); vmin = seg_min(vmin, v);
                }
            }
        }
        vmax = pls_wave_max_i(vmax); vmin = pls_wave_min_i(vmin);
        if (PLS_WAVE_LEADER(tid)) { PLS_ATOMIC_MAX_U(&mm[0], 0x80000000u ^ (uint32_t)vmax); PLS_ATOMIC_MIN(&mm[1], 0x80000000u ^ (uint32_t)vmin); }
    }
    PLS_SYNC();
    PLS_THREADS(tid, CT) { if (tid == 0) { SEG_AS_GLB int32_t *rmm = seg_rowmm(j, y); rmm[0] = (int)(mm[0] ^ 0x80000000u); rmm[1] = (int)(mm[1] ^ 0x80000000u); } }
}

template <int RNT>
PLS_HD void seg_replay_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int grp, unsigned char *smem)
{
    const SegCtl &ctl = j.ctl[par];
    if (cv.finished) return;
    const bool lazy = f == 0 && cv.active == 2;               /* candidate none, not run yet: only its cost bound is wanted */
    if (!lazy && cv.active != 1) return;
    const uint32_t W = j.W, bpp = j.bpp, nseg = j.nseg;
    const uint32_t sx = cv.start_x;
    const uint32_t first = sx / SEG_L;
    const uint32_t seg0 = (uint32_t)grp * SEG_GRP, x0g = seg0 * SEG_L;
    const bool walk = !lazy && sx < W && seg0 + SEG_GRP > first;   /* (else: the whole group is validated already -- its share of the row's sums is still wanted) */
    uint32_t *Hf = (uint32_t *)smem, *rank = Hf + 256, *lut = Hf + 512, *tw = Hf + 1024;
    SegPix *px = (SegPix *)(tw + SEG_TBL_WORDS);              /* [SEG_GRP][SEG_L][4] */
    uint32_t *cnt = (uint32_t *)(px + SEG_GRP * SEG_L * 4);   /* [SEG_GRP][64]: four bins a word (seg_cnt_add) */
    uint32_t *lane = cnt + SEG_GRP * 64;                      /* [SEG_GRP * SEG_PARTS * 4][2]: start state, first | end pixel << 16 (or ~0: idle) */
    uint32_t *cwl = lane + SEG_GRP * SEG_PARTS * 4 * 2;       /* [SEG_REPLAY_THREADS][4] the group's candidate words */
    uint32_t *oaL = cwl + SEG_REPLAY_THREADS * 4;             /* [SEG_REPLAY_THREADS + 1] the ORIGINAL row above, from the pixel in front of the group; [+1] the original pixel in front of the group, [+2] its new bytes */
    /* (behind the walk, in the tables' place) */
    uint32_t *rm = tw;                                         /* [768] none's bound: largest H0 within reach of a centre value (centre + 256) */
    uint32_t *red = rm + 768;                                  /* [16] reductions: derr lo/hi, hs[5], -, -, -, lb lo/hi, reach, -, max, min */
    uint32_t *h0s = red + 16;                                  /* [256] the committed histogram */
    const uint32_t y = cv.y;
    const uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const uint32_t *oab = y ? seg_row_orig(j, y - 1u) : nullptr; /* the ORIGINAL row above */
    const SegGeo G = seg_geo((int)cv.s);
    const bool adaptive = !j.row_filters || y == 0;           /* pngloss_image.c:210 */
    const bool rprof = SEG_EXPERIMENT_REPLAY_CLOCKS && (P.engine_flags & 1) != 0 && walk && grp == 3;
    unsigned long long tr_[6] = { 0, 0, 0, 0, 0, 0 };
    if (rprof) tr_[0] = PLS_CLOCK();
    PLS_THREADS(tid, RNT) {
        /* Order of the requests: the walkers' dense ids, then everything the block stages (tables, frozen histogram, pixels), then the
         * checkpoints and entry states (which wait for the dense ids only); the stores to shared memory behind all of them. */
        const bool walker = walk && tid < SEG_GRP * SEG_PARTS * 4;
        const int sl = tid / (SEG_PARTS * 4), part = (tid >> 2) % SEG_PARTS, c = tid & 3;
        const uint32_t sg = seg0 + (uint32_t)sl;
        const bool live = walker && sg < nseg && sg >= first && (uint32_t)c < bpp;
        const size_t sc = ((size_t)f * nseg + (live ? sg : 0u)) * 4 + c;
        const uint32_t d = live ? (uint32_t)j.dnout[sc] : SEG_INVALID;
        constexpr int NTW = (SEG_TBL_WORDS + RNT - 1) / RNT;
        uint32_t vt[NTW], vl = 0, vh = 0, vb = 0, vr = 0;
        PLS_UNROLL
        for (int q = 0; q < NTW; q++) { const int i = tid + q * RNT; vt[q] = (walk && i < SEG_TBL_WORDS) ? j.tables[(size_t)f * SEG_TBL_WORDS + i] : 0u; }
        if (walk && tid < 512) vl = P.lut_a[tid];
        if (walk && tid < 256) { vh = j.H0[par * 256 + tid]; vb = j.base[((size_t)par * SEG_NFILT + f) * 256 + tid]; vr = j.orig_rank[f * 256 + tid]; }
        const bool pixlane = tid < SEG_REPLAY_THREADS;            /* (the first half of the workgroup has a pixel of the group each; the second half helps with the staging and takes the counts) */
        const uint32_t x = pixlane ? x0g + (uint32_t)tid : W;
        const SegPixRaw vp = seg_pix_fetch(row, nab, e0g, x, W);
        /* for the sums: the original row above; what lies in front of the group (lane 0); the validated words of an epoch that starts further right */
        const uint32_t voa = (!lazy && oab && x < W) ? oab[x] : 0u;
        uint32_t voa0 = 0, vol0 = 0;
        SegVec16 vw, vw0; vw.a = vw.b = vw.c = vw.d = 0u; vw0 = vw;
        if (!lazy && tid == 0 && x0g) { voa0 = oab ? oab[x0g - 1] : 0u; vol0 = row[x0g - 1]; if (x0g <= sx) vw0 = *(const SEG_AS_GLB SegVec16 *)(j.cand + ((size_t)f * W + x0g - 1) * 4); }
        if (!lazy && x < W && (x < sx || !walk)) vw = *(const SEG_AS_GLB SegVec16 *)(j.cand + ((size_t)f * W + x) * 4);
        uint32_t ck[SEG_PARTS - 1], ent = 0;
        PLS_UNROLL
        for (int q = 0; q < SEG_PARTS - 1; q++) ck[q] = 0xFFFFFFFFu;
        if (live) {
            if (d != SEG_INVALID && d < SEG_NSP) {
                PLS_UNROLL
                for (int q = 0; q < SEG_PARTS - 1; q++) ck[q] = j.rck[(sc * SEG_NSP + d) * (SEG_PARTS - 1) + q];
            }
            if (part == 0) ent = sg == first ? ctl.state[f][c] : j.entry[sc];
        }
        uint32_t st0 = 0, range = 0xFFFFFFFFu;
        if (live) {
            const uint32_t x0 = sg * SEG_L, xend = (uint32_t)seg_min((int)(x0 + SEG_L), (int)W);
            /* this lane's part has a start state: part 0 always (entry state, or the epoch's start inside the row's first segment) */
            bool mine = part == 0;
            uint32_t xa = sg == first ? sx : x0;
            if (part == 0) st0 = ent;
            else {
                uint32_t ckp = 0xFFFFFFFFu;
                PLS_UNROLL
                for (int q = 0; q < SEG_PARTS - 1; q++) if (q == part - 1) ckp = ck[q];
                if (ckp != 0xFFFFFFFFu && x0 + (uint32_t)part * SEG_PL < xend && x0 + (uint32_t)part * SEG_PL > xa) { mine = true; st0 = ckp; xa = x0 + (uint32_t)part * SEG_PL; }
            }
            if (mine) {
                uint32_t xe = xend;                        /* up to the next part that starts on its own */
                PLS_UNROLL
                for (int q = SEG_PARTS - 1; q >= 1; q--)
                    if (q > part && ck[q - 1] != 0xFFFFFFFFu && x0 + (uint32_t)q * SEG_PL < xend && x0 + (uint32_t)q * SEG_PL > xa) xe = x0 + (uint32_t)q * SEG_PL;
                range = (xa - x0g) | ((xe - x0g) << 16);
                SEG_DEBUG_COUNT(part ? 1 : 0, xe - xa);
            }
        }
        if (tid < SEG_GRP * SEG_PARTS * 4) { lane[2 * tid] = st0; lane[2 * tid + 1] = range; }
        if (walk) {
            PLS_UNROLL
            for (int q = 0; q < NTW; q++) { const int i = tid + q * RNT; if (i < SEG_TBL_WORDS) tw[i] = vt[q]; }
            if (tid < 512) lut[tid] = vl;
            if (tid < 256) { Hf[tid] = vh + vb; rank[tid] = vr; }
            for (int i = tid; i < SEG_GRP * 64; i += RNT) cnt[i] = 0u;
        }
        if (pixlane) {
            seg_pix_split4(px + tid * 4, vp, bpp, x, W);
            oaL[tid + 1] = voa;
            cwl[tid * 4 + 0] = vw.a; cwl[tid * 4 + 1] = vw.b; cwl[tid * 4 + 2] = vw.c; cwl[tid * 4 + 3] = vw.d;
        }
        if (tid == 0) {
            oaL[0] = voa0; oaL[SEG_REPLAY_THREADS + 1] = vol0;
            oaL[SEG_REPLAY_THREADS + 2] = (vw0.a & 255u) | ((vw0.b & 255u) << 8) | ((vw0.c & 255u) << 16) | ((vw0.d & 255u) << 24);
        }
    }
    PLS_SYNC();
    if (rprof) tr_[1] = PLS_CLOCK();
    if (walk) {
        PLS_THREADS(tid, RNT) {
            if (tid < SEG_GRP * SEG_PARTS * 4 && lane[2 * tid + 1] != 0xFFFFFFFFu) {
                const int sl = tid / (SEG_PARTS * 4), c = tid & 3;
                SegState st = seg_state_unpack(lane[2 * tid]);
                const uint32_t ra = lane[2 * tid + 1] & 0xffffu, re = lane[2 * tid + 1] >> 16;       /* pixels of the group */
                seg_walk(f, px + ra * 4 + c, 4, x0g + ra, x0g + re, st, SEG_LDS_CU32(tw), SEG_LDS_CU32(lut), Hf, rank, G, lut, P.bleed, cwl + ra * 4 + c, cnt + sl * 64);
            }
        }
        PLS_SYNC();
        if (rprof) tr_[2] = PLS_CLOCK();
    }
    SegAcc &A = j.acc[par];
    if (rprof) tr_[3] = PLS_CLOCK();
    PLS_THREADS(tid, RNT) { if (tid < 16) red[tid] = tid == 14 ? (0x80000000u ^ (uint32_t)(-(1 << 30))) : (tid == 15 ? (0x80000000u ^ (uint32_t)(1 << 30)) : 0u); }   /* ([14], [15] biased: unsigned max / min) */
    PLS_SYNC();
    {
        /* -- the first half of the workgroup: every thread its pixel: the words out (what the walkers have just written), the pixel's share of the
         *    sums; a quarter meanwhile: the bump counts per segment and group, a bin a thread -- */
        PLS_THREADS(tid, RNT) {
            /* (the counts: 256 threads of the workgroup's second half -- or, in the 512-thread version of batches, the first 256, ahead of their pixel) */
            if (walk && tid >= RNT - SEG_REPLAY_THREADS && tid < RNT - SEG_REPLAY_THREADS + 256) {
                const int b = tid - (RNT - SEG_REPLAY_THREADS);
                uint32_t tot = 0, cvs[SEG_GRP];
                PLS_UNROLL
                for (int sl = 0; sl < SEG_GRP; sl++) cvs[sl] = seg_cnt_get(cnt + sl * 64, b);          /* (all reads, then the stores: one wait) */
                PLS_UNROLL
                for (int sl = 0; sl < SEG_GRP; sl++) {
                    const uint32_t sg = seg0 + (uint32_t)sl;
                    if (sg < nseg && sg >= first) { j.segcnt[((size_t)f * nseg + sg) * 256 + b] = (uint16_t)cvs[sl]; tot += cvs[sl]; }
                }
                j.grpcnt[((size_t)f * j.ngrp + grp) * 256 + b] = tot;
            }
            const uint32_t x = (!lazy && tid < SEG_REPLAY_THREADS) ? x0g + (uint32_t)tid : W;
            uint64_t derr = 0; uint32_t hs[SEG_NFILT] = { 0, 0, 0, 0, 0 };
            if (x < W) {
                SegVec16 w; w.a = cwl[tid * 4 + 0]; w.b = cwl[tid * 4 + 1]; w.c = cwl[tid * 4 + 2]; w.d = cwl[tid * 4 + 3];
                if (walk && x >= sx) *(SEG_AS_GLB SegVec16 *)(j.cand + ((size_t)f * W + x) * 4) = w;
                /* the new bytes of the pixel in front: the thread's neighbour's; for the group's first pixel the validated word in front of it, or
                 * (another workgroup is writing it now) the left bytes of the entry states this group's walkers started from */
                uint32_t lw[4];
                if (tid) { PLS_UNROLL for (int c = 0; c < 4; c++) lw[c] = cwl[(tid - 1) * 4 + c] & 255u; }
                else {
                    uint32_t l0 = 0;
                    if (x0g && x0g <= sx) l0 = oaL[SEG_REPLAY_THREADS + 2];
                    else if (x0g) {
                        for (uint32_t c = 0; c < bpp; c++) l0 |= (walk ? (lane[2 * c] & 255u) : 0u) << (8 * c);
                        j.grpleft[(size_t)f * j.ngrp + grp] = l0;
                    }
                    PLS_UNROLL for (int c = 0; c < 4; c++) lw[c] = (l0 >> (8 * c)) & 255u;
                }
                const uint32_t wv[4] = { w.a, w.b, w.c, w.d };
                const uint32_t oav4 = oaL[tid + 1], odv4 = oaL[tid];                       /* (zero on the first row; the slot in front of pixel 0 holds zero) */
                for (uint32_t c = 0; c < bpp; c++) {
                    const int sh = 8 * (int)c;
                    const SegPix pc = px[tid * 4 + c];
                    const int back = (int)(wv[c] & 255u), nl = x ? (int)lw[c] : 0;
                    const int ov = (int)(pc.w & 255u), nav = (int)((pc.w >> 8) & 255u), ndv = (int)((pc.w >> 16) & 255u);
                    const int olv = tid ? (int)(px[(tid - 1) * 4 + c].w & 255u) : (int)((oaL[SEG_REPLAY_THREADS + 1] >> sh) & 255u);
                    const int oav = (int)((oav4 >> sh) & 255u), odv = (int)((odv4 >> sh) & 255u);
                    const int da = (oav - ov) - (nav - back), dd = (odv - ov) - (ndv - back), dl = (olv - ov) - (nl - back);
                    const uint32_t wgt = (bpp <= 2 && c == 0) ? 3u : 1u;      /* gray is replicated into r,g,b (color_delta.c:11-26) */
                    derr += (uint64_t)(wgt * (uint32_t)(da * da + dd * dd + dl * dl));
                    if (adaptive) {
                        const int preds[SEG_NFILT] = { 0, nl, nav, (nav + nl) >> 1, seg_paeth(nav, ndv, nl) };
                        for (int g = 0; g < SEG_NFILT; g++) { const int bb = (back - preds[g]) & 255; hs[g] += (uint32_t)(bb < 128 ? bb : 256 - bb); }
                    }
                }
            }
            /* (a pixel's share is at most 4 channels x 3 weights x 3 squares of 510: a wave's sum fits 32 bits -- half the shuffles of a 64-bit one) */
            derr = (uint64_t)pls_wave_sum_u32((uint32_t)derr);
            if (adaptive) for (int g = 0; g < SEG_NFILT; g++) hs[g] = pls_wave_sum_u32(hs[g]);
            if (PLS_WAVE_LEADER(tid)) {
                PLS_ATOMIC_ADD64((uint64_t *)&red[0], derr);
                if (adaptive) for (int g = 0; g < SEG_NFILT; g++) PLS_ATOMIC_ADD(&red[2 + g], hs[g]);
            }
        }
    }
    /* -- candidate none only: a LOWER BOUND of its row cost that needs no chain (see seg_none_reach).  Every symbol of none is the
     *    reconstructed byte itself, which lies within R of orig + incoming error (clamped to 0..255); its cost is at least the cost
     *    of the most frequent bin within that reach after the row: 33 + clz(max H0 + all bumps of the row). -- */
    int R = -1;
    if (f == 0 && j.rowmm) {
        const int32_t *rmm = seg_rowmm(j, y);
        PLS_THREADS(tid, RNT) {
            /* the row's extremes of orig + incoming error (seg_extremes_body, the launch before) */
            if (tid == 0) red[12] = (uint32_t)seg_none_reach(j, P, (int)cv.s, rmm[0], rmm[1]);
            if (tid >= 64 && tid < 64 + 256) h0s[tid - 64] = j.H0[par * 256 + (tid - 64)];
        }
        PLS_SYNC();
        R = (int)red[12];
        if (R >= 0) {
            PLS_THREADS(tid, RNT) {
                for (int i = tid; i < 768; i += RNT) {
                    const int centre = i - 256;
                    const int lo = seg_min(seg_max(centre - R, 0), 255), hi = seg_min(seg_max(centre + R, 0), 255);
                    uint32_t m = 0;
                    /* eight reads in flight per turn (indices past the range are clamped onto its end: they change nothing); one read a turn waits
                     * for each of up to 2R + 1 = 50 and more on its own -- these workgroups were as long as the walkers' */
                    for (int b = lo; b <= hi; b += 8) {
                        uint32_t h[8];
                        PLS_UNROLL
                        for (int q = 0; q < 8; q++) h[q] = h0s[seg_min(b + q, hi)];
                        PLS_UNROLL
                        for (int q = 0; q < 8; q++) m = h[q] > m ? h[q] : m;
                    }
                    rm[i] = m;
                }
            }
            PLS_SYNC();
            PLS_THREADS(tid, RNT) {
                uint64_t lb = 0;
                const uint32_t rowbumps = W * bpp;
                const uint32_t x = tid < SEG_REPLAY_THREADS ? x0g + (uint32_t)tid : W;
                if (x < W) {
                    for (uint32_t c = 0; c < bpp; c++) {
                        const SegPix pc = px[tid * 4 + c];
                        uint32_t hmax;
                        if (pc.w >> 24) hmax = h0s[0];                                      /* forced symbol 0 (the alpha of a fully transparent pixel) */
                        else hmax = rm[seg_min(seg_max((int)(pc.w & 255u) + pc.e0, -256), 511) + 256];
                        const uint32_t fr = hmax + rowbumps;
                        lb += 33u + (uint32_t)__builtin_clz(fr ? fr : 1u);
                    }
                }
                lb = pls_wave_sum_u64(lb);
                if (PLS_WAVE_LEADER(tid) && lb) PLS_ATOMIC_ADD64((uint64_t *)&red[10], lb);
            }
        }
    }
    PLS_SYNC();
    PLS_THREADS(tid, RNT) {
        if (tid == 0) {
            if (!lazy) {
                PLS_ATOMIC_ADD64(&A.derr[f], *(uint64_t *)&red[0]);
                if (adaptive) for (int g = 0; g < SEG_NFILT; g++) PLS_ATOMIC_ADD(&A.hs[f][g], red[2 + g]);
            }
            if (f == 0 && R >= 0) { PLS_ATOMIC_ADD64(&A.none_lb, *(uint64_t *)&red[10]); PLS_ATOMIC_ADD(&A.lb_valid, 1u); }
            if (rprof) {
                tr_[4] = PLS_CLOCK();
                for (int q = 0; q < 4; q++) { PLS_ATOMIC_MAX(&j.result[24 + q], (int32_t)(tr_[q + 1] - tr_[q])); PLS_ATOMIC_ADD((uint32_t *)&j.result[28 + q], (uint32_t)(tr_[q + 1] - tr_[q])); }
                PLS_ATOMIC_ADD((uint32_t *)&j.result[32], 1u);
            }
        }
    }
}

/* Validation of one decision d = (pixel k of the group) * 4 + channel.  mode 0: with the block bounds only -- returns 1 good, 0 bad,
 * 2 cannot tell (its bins are marked in wbits); mode 1: exactly, with the per-segment prefix counts of the watched bins (pc), or by
 * counting the earlier decisions of the segment for a bin that got no slot. */
struct SegVal {
    const uint32_t *cw, *ro, *na, *e0, *lut, *H0, *rank, *cum;
    uint32_t bpp, sx, xg0, W;
    int f, bleed;
    SegGeo G;
    uint32_t *wbits;
    const uint8_t *slot_of;
    const uint32_t *pcw;          /* prefix counts of the watched bins, 4 decisions per word: [slot][pc_stride] */
    uint32_t pc_stride;
    uint8_t *binb;                /* bin of every decision, [segment of the group][SEG_BINB_STRIDE bytes] */
    const uint32_t *btop;         /* [2][SEG_NBAND][4]: per band of either sign: largest upper count bound, its bin, second largest */
    const uint32_t *hiG, *loG;    /* [256] frequency of a bin: upper bound (through the group's end), lower bound (at the group's start) */
};
#define SEG_BINB_STRIDE (SEG_L * 4 + 4)
#define SEG_PC_SEG (SEG_L + 1)                   /* words per segment and slot (one pad word: bank spread) */
#define SEG_PC_STRIDE_V(V) ((V) * SEG_PC_SEG + 8)
#define SEG_NBAND 20
PLS_HD uint32_t seg_pc_get(const uint32_t *pcw, uint32_t pc_stride, uint32_t slot, int d)
{
    const uint32_t w = pcw[(size_t)slot * pc_stride + (d / (SEG_L * 4)) * SEG_PC_SEG + ((d % (SEG_L * 4)) >> 2)];
    return (w >> (8 * (d & 3))) & 255u;
}
PLS_HD int seg_validate_one(const SegVal &V, int d, int mode)
{
    const int c = d & 3, k = d >> 2;
    const uint32_t x = V.xg0 + (uint32_t)k, bpp = V.bpp;
    const int sl = k / SEG_L;
    const uint32_t w0 = V.cw[(k + 2) * 4 + c], w1 = V.cw[(k + 1) * 4 + c], w2 = V.cw[k * 4 + c];
    if (mode == 0) V.binb[sl * SEG_BINB_STRIDE + (d - sl * SEG_L * 4)] = (uint8_t)seg_cand_bin(w0);
    if (x >= V.W || x < V.sx || (uint32_t)c >= bpp) return 1;
    const uint32_t o = V.ro[k + 1];
    const int pe0 = seg_err_plane(V.e0 + 2 * k, seg_plane_of_channel(bpp, c));
    const bool trp = (bpp & 1u) == 0u && (uint32_t)c == bpp - 1u && ((o >> (8u * (bpp - 1u))) & 255u) == 0u;
    /* state in front of x from the outputs of x-1, x-2 */
    int rem1, thr1, rem2, thr2;
    seg_rem_thr(V.lut, V.bleed, x >= 1 ? seg_cand_diff(w1) : 0, rem1, thr1);
    seg_rem_thr(V.lut, V.bleed, x >= 2 ? seg_cand_diff(w2) : 0, rem2, thr2);
    const int left = x >= 1 ? seg_cand_byte(w1) : 0, cn = rem1 + thr2;
    const int orig = (int)((o >> (8 * c)) & 255u), above = (int)((V.na[k + 1] >> (8 * c)) & 255u), diag = x ? (int)((V.na[k] >> (8 * c)) & 255u) : 0;
    const int pred = seg_predict(V.f, above, diag, left);
    const int back = seg_cand_byte(w0), diff = seg_cand_diff(w0), bin = seg_cand_bin(w0);
    if (trp) return (back == 0 && diff == 0 && bin == ((0 - pred) & 255)) ? 1 : 0;
    const int osym = seg_sext8(orig - pred), lo = osym - orig;
    const int filt = osym + seg_sext16(pe0 + cn);
    const SegBand bd = seg_band(filt, lo, V.G);
    const int v = back + lo;
    if (!(v >= bd.v0 && v <= bd.v1 && diff == seg_sext16(filt - v) && bin == (v & 255))) return 0;
    if (bd.v0 == bd.v1) return 1;
    if (mode == 0 && bd.t < SEG_NBAND) {
        /* the clamped range is part of band t: if every OTHER bin of the whole band stays strictly below v's lower bound for the
         * whole group, nothing in the range can beat v -- one comparison for most decisions */
        const uint32_t *bt = V.btop + ((size_t)bd.neg * SEG_NBAND + bd.t) * 4;
        const uint32_t other = bt[1] == (uint32_t)bin ? bt[2] : bt[0];
        if (other < V.loG[bin]) return 1;
    }
    const uint32_t *cs = V.cum + sl * 256, *ce = V.cum + (sl + 1) * 256;
    const int kseg = seg_max(sl * SEG_L, (int)V.sx - (int)V.xg0);   /* first pixel of this decision's segment that belongs to the epoch */
    const uint32_t hv_lo = V.H0[bin] + cs[bin], rv = V.rank[bin];
    const int fv = v == osym;
    bool have_hv = false; uint32_t hv_exact = 0;
    int result = 1;
    for (int u = bd.v0; u <= bd.v1; u++) {
        if (u == v) continue;
        const int ub = u & 255;
        const uint32_t ru = V.rank[ub];
        const int fu = u == osym;
        const bool u_wins_ties = ru != rv ? ru > rv : (fu != fv ? fu > fv : u < v);      /* u beats v at equal frequency?  (O, flag, lower v) */
        const uint32_t hu_hi = V.H0[ub] + ce[ub];
        if (u_wins_ties ? hu_hi < hv_lo : hu_hi <= hv_lo) continue;                      /* proven by the bounds */
        if (mode == 0) {
            PLS_ATOMIC_OR(&V.wbits[bin >> 5], 1u << (bin & 31));
            PLS_ATOMIC_OR(&V.wbits[ub >> 5], 1u << (ub & 31));
            result = 2;
            continue;
        }
        if (!have_hv) {
            uint32_t n = 0;
            const uint32_t sv = V.slot_of[bin];
            if (sv != 255u) n = seg_pc_get(V.pcw, V.pc_stride, sv, d);
            else for (int e = kseg * 4; e < d; e++) if ((uint32_t)(e & 3) < bpp && seg_cand_bin(V.cw[e + 8]) == bin) n++;
            hv_exact = hv_lo + n; have_hv = true;
        }
        uint32_t n = 0;
        const uint32_t su = V.slot_of[ub];
        if (su != 255u) n = seg_pc_get(V.pcw, V.pc_stride, su, d);
        else for (int e = kseg * 4; e < d; e++) if ((uint32_t)(e & 3) < bpp && seg_cand_bin(V.cw[e + 8]) == ub) n++;
        const uint32_t hu = V.H0[ub] + cs[ub] + n;
        if (u_wins_ties ? hu >= hv_exact : hu > hv_exact) return 0;
    }
    if (result == 2) PLS_ATOMIC_ADD(&V.wbits[8], 1u);
    return result;
}

/* Candidate none (prediction 0): how far from  orig + incoming error  its reconstructed byte can lie.  With C a bound of the carried
 * terms |cn| and D of the quantisation differences |diff|:  |byte - (orig + e0)| <= s + C  (the chosen v lies in the band of filt, or the
 * clamp moved it towards 0..255);  D <= s + overshoot(C), overshoot = how far orig + e0 +- C can leave 0..255 (rowmm holds the row's
 * extremes of orig + e0);  C <= max |rem| + max |thr| over |d| <= D.  Iterated from C = the table bound to a fixed point; -1 = none found
 * (then no bound is claimed). */
PLS_HD int seg_none_reach(const SegJob &j, const SegParams &P, int s, int M, int m)
{
    if (!j.rowmm) return -1;
    int C = P.cmax;
    for (int it = 0; it < 4; it++) {
        const int ov = seg_max(0, seg_max(M + C - 255, C - m));
        const int D = s + ov;
        if (D > 255) return -1;
        const int C2 = seg_max(C, (int)P.rt_max[D]);
        if (C2 == C) return s + C;
        C = C2;
    }
    return -1;
}

/* ---- VALIDATE + POST: task (f, grp) ----------------------------------------------------------------------------------------
 * A decision (x, c) of the candidate row is CORRECT iff, with everything re-derived from the outputs of x-1 and x-2 (left byte,
 * carried error terms) and from the data, the stored byte/diff/bin are consistent and the chosen v is the reference's arg-max
 * over its clamped band under the histogram H0 + (bumps of all earlier decisions of the row, in chain order).  The bumps in
 * front of a decision: validated prefix (base) + whole groups + whole segments (counts written by the replay) + the earlier
 * decisions of its own segment (counted here).  Cheap bound first (counts at the segment's start and end), exact count only when
 * the bound cannot tell. */
template <int VGRP>
PLS_HD void seg_post_body(const SegJob &j, const SegParams &P, const SegCtlView &cv, int par, int f, int vg, unsigned char *smem)
{
    if (cv.finished || cv.active != 1) return;                 /* (candidate none while it is lazy has no row to validate) */
    const uint32_t W = j.W, bpp = j.bpp, nseg = j.nseg, ngrp = j.ngrp;
    const uint32_t sx = cv.start_x;
    if (sx >= W) return;                                       /* (a row finished serially by the control kernel: exact by construction) */
    const uint32_t first = sx / SEG_L, fgrp = first / SEG_GRP;
    const uint32_t seg0 = (uint32_t)vg * VGRP;             /* vg: validation group = VGRP segments (half a replay group, or all of it) */
    const uint32_t grp = seg0 / SEG_GRP, segp = grp * SEG_GRP;  /* the replay group it lies in, and that group's first segment */
    constexpr int NPX = VGRP * SEG_L;                       /* pixels of a group */
    constexpr int NW = SEG_WATCH_OF(VGRP);                  /* slots of the watched bins */
    constexpr int PCS = SEG_PC_STRIDE_V(VGRP);              /* words per slot of the watched bins' prefix counts */
    uint32_t *H0 = (uint32_t *)smem, *rank = H0 + 256;
    uint32_t *cum = H0 + 768;                                  /* [VGRP + 1][256]: bumps in front of each segment of the group (staging: row sl + 1 = the bumps OF segment sl) */
    uint32_t *cw = cum + (VGRP + 1) * 256;                 /* [(NPX + 2)][4] candidate words, from pixel xg0 - 2 */
    uint32_t *red = cw + (NPX + 2) * 4;                        /* reductions: derr lo/hi, cost, hs[5], fail, lb lo/hi */
    uint32_t *lut = red + 64;                                  /* [512] split table */
    uint32_t *ro = lut + 512;                                  /* [NPX + 1] original row, from pixel xg0 - 1 */
    uint32_t *na = ro + NPX + 2;                               /* [NPX + 1] optimised row above */
    uint32_t *e0 = na + NPX + 2;                               /* [NPX][2] incoming error */
    uint32_t *wbits = e0 + 2 * NPX;                            /* [8] bitmap of watched bins, [8] pending decisions / slots in use, [9..] bin of each slot */
    uint8_t *slot_of = (uint8_t *)(wbits + 32);                /* [256] slot of a watched bin or 255 */
    uint8_t *pend = slot_of + 256;                             /* [NPX * 4] decision waits for pass 3 */
    uint8_t *binb = pend + NPX * 4;                            /* [VGRP][SEG_BINB_STRIDE] bin of every decision */
    uint32_t *pcw = (uint32_t *)(binb + VGRP * SEG_BINB_STRIDE);   /* [NW][PCS] prefix counts of the watched bins */
    uint32_t *cumx = pcw;                                      /* [VGRP][256] (staging only, before pcw is written: barriers lie between) the bumps of the replay group's segments in front of this half */
    constexpr int NFR = SEG_GRP - VGRP;                     /* segments of the replay group that can lie in front of this validation group */
    static_assert(NW * PCS >= NFR * 256, "the staged counts of the segments in front fit where the prefix counts go later");
    static_assert(VGRP + NFR == SEG_THREADS / 64 && (NFR == 0 || NFR == VGRP), "the staging burst's sixteen rows of lanes: this group's segments, then the ones in front");
    uint32_t *btop = pcw + NW * PCS;          /* [2][SEG_NBAND][4] */
    uint32_t *hiG = btop + 2 * SEG_NBAND * 4, *loG = hiG + 256;
    const uint32_t y = cv.y;
    const uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
    const SegGeo G = seg_geo((int)cv.s);
    const uint32_t xg0 = seg0 * SEG_L;
    const bool prof = (P.engine_flags & 1) != 0;                /* debugging: phase clocks (100 MHz ticks) into result[40..], max over the workgroups */
    unsigned long long tk[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (prof) tk[0] = PLS_CLOCK();
    PLS_THREADS(tid, SEG_THREADS) {
        /* every request first, the stores to shared memory behind them: one round trip instead of one per statement (the compiler
         * cannot move a load in front of an earlier store through a generic pointer) */
        constexpr int NCW = ((NPX + 2) * 4 + SEG_THREADS - 1) / SEG_THREADS;
        uint32_t vh0 = 0, vrank = 0, vlut = 0, vcw[NCW], vro = 0, vna = 0, vgl = 0, ve0a = 0, ve0b = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (tid < 256) { vh0 = j.H0[par * 256 + tid]; vrank = j.orig_rank[f * 256 + tid]; }
        /* (round 5) the bumps in front of the group -- validated prefix + the whole replay groups in front -- requested in THIS burst (they need nothing that is staged
         * here): the prefix pass below had a round trip to device memory of its own for them */
        uint32_t vbefore = 0;
        if (tid < 256) {
            const int b = tid;
            vbefore = j.base[((size_t)par * SEG_NFILT + f) * 256 + b];
            uint32_t gv[SEG_NG_BURST];
            PLS_UNROLL
            for (int g = 0; g < SEG_NG_BURST; g++) gv[g] = ((uint32_t)g < grp) ? j.grpcnt[((size_t)f * ngrp + g) * 256 + b] : 0u;   /* (grp < ngrp) */
            PLS_UNROLL
            for (int g = 0; g < SEG_NG_BURST; g++) {
                const bool in = (uint32_t)g >= fgrp && (uint32_t)g < ngrp;
                vbefore += (in && (uint32_t)g < grp) ? gv[g] : 0u;
            }
            for (uint32_t g = SEG_NG_BURST; g < grp && g < ngrp; g++) {           /* (rows beyond 8192 pixels) */
                const uint32_t v = j.grpcnt[((size_t)f * ngrp + g) * 256 + b];
                vbefore += g >= fgrp ? v : 0u;
            }
        }
        if (tid >= 256 && tid < 768) vlut = P.lut_a[tid - 256];
        PLS_UNROLL
        for (int q = 0; q < NCW; q++) {
            const int i = tid + q * SEG_THREADS;
            const long x = (long)xg0 - 2 + (i >> 2);
            vcw[q] = (i < (NPX + 2) * 4 && x >= 0 && x < (long)W) ? j.cand[((size_t)f * W + (size_t)x) * 4 + (i & 3)] : 0u;
        }
        if (tid <= NPX) {
            const long x = (long)xg0 - 1 + tid;
            const bool in = x >= 0 && x < (long)W;
            vro = in ? row[x] : 0u;
            vna = (in && nab) ? nab[x] : 0u;
        }
        if (tid == 9) vgl = j.grpleft[(size_t)f * ngrp + grp];
        if (tid >= 512 && tid - 512 < NPX) {
            const uint32_t x = xg0 + (uint32_t)(tid - 512);
            ve0a = x < W ? e0g[2 * (size_t)x] : 0u;
            ve0b = x < W ? e0g[2 * (size_t)x + 1] : 0u;
        }
        /* bump counts per segment of the group, staged (one 8-byte load per thread), prefix below */
        const int sl = tid >> 6, q4 = tid & 63;
        {
            /* lanes of rows 0 .. VGRP-1: this group's segments; rows VGRP ..: the segments of the same REPLAY group in front of it
             * (the replay's group counts are per SEG_GRP segments: what lies between that group's start and ours is added from these) */
            const uint32_t sg = sl < VGRP ? seg0 + (uint32_t)sl : segp + (uint32_t)(sl - VGRP);
            if (sg < nseg && sg >= first && sx < W && (sl < VGRP || sg < seg0)) {
                const uint16_t *sc = j.segcnt + ((size_t)f * nseg + sg) * 256 + 4 * q4;
                c0 = sc[0]; c1 = sc[1]; c2 = sc[2]; c3 = sc[3];
            }
        }
        if (tid < 256) { H0[tid] = vh0; rank[tid] = vrank; hiG[tid] = vbefore; }      /* (hiG: parked until the prefix pass has read it; the bounds are written behind that) */
        if (tid < 16) red[tid] = tid == 8 ? SEG_NOFAIL : (tid == 9 ? vgl : 0u);
        if (tid >= 256 && tid < 768) lut[tid - 256] = vlut;
        PLS_UNROLL
        for (int q = 0; q < NCW; q++) { const int i = tid + q * SEG_THREADS; if (i < (NPX + 2) * 4) cw[i] = vcw[q]; }
        if (tid <= NPX) { ro[tid] = vro; na[tid] = vna; }
        if (tid >= 512 && tid - 512 < NPX) { e0[2 * (tid - 512)] = ve0a; e0[2 * (tid - 512) + 1] = ve0b; }
        {
            uint32_t *dst = (sl < VGRP ? cum + (sl + 1) * 256 : cumx + (sl - VGRP) * 256) + 4 * q4;
            dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3;
        }
    }
    PLS_SYNC();
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < 256) {
            const int b = tid;
            uint32_t before = hiG[b];                                  /* (base + the replay groups in front: summed by the staging burst) */
            uint32_t add[VGRP + NFR];                              /* (all sixteen rows read first: a read behind a store to the same array waits for it) */
            PLS_UNROLL
            for (int r = 1; r <= VGRP + NFR; r++) add[r - 1] = r <= VGRP ? cum[r * 256 + b] : cumx[(r - VGRP - 1) * 256 + b];
            PLS_UNROLL
            for (int r = VGRP + 1; r <= VGRP + NFR; r++) before += add[r - 1];
            uint32_t run = before;
            PLS_UNROLL
            for (int sl = 0; sl <= VGRP; sl++) {
                cum[sl * 256 + b] = run;
                if (sl < VGRP) run += add[sl];
            }
        }
    }
    PLS_SYNC();
    if (prof) tk[1] = PLS_CLOCK();
    {
    /* -- validation: lane = decision.  Pass 1 settles what the block bounds can settle and marks the bins of the others ("watched");
     *    pass 2 counts, per segment, the bumps of each watched bin in front of every decision; pass 3 settles the rest exactly. -- */
    SegVal V;
    V.cw = cw; V.ro = ro; V.na = na; V.e0 = e0; V.lut = lut; V.H0 = H0; V.rank = rank; V.cum = cum; V.bpp = bpp; V.f = f; V.G = G; V.bleed = P.bleed;
    V.sx = sx; V.xg0 = xg0; V.W = W; V.wbits = wbits; V.slot_of = slot_of; V.pcw = pcw; V.pc_stride = PCS; V.binb = binb; V.btop = btop; V.hiG = hiG; V.loG = loG;
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < 8) wbits[tid] = 0u;
        if (tid == 8) wbits[8] = 0u;                                   /* number of pending decisions */
        if (tid >= 64 && tid < 128) ((uint32_t *)slot_of)[tid - 64] = 0xffffffffu;
        if (tid >= 256 && tid < 512) { const int b = tid - 256; hiG[b] = H0[b] + cum[VGRP * 256 + b]; loG[b] = H0[b] + cum[b]; }
    }
    PLS_SYNC();
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < 2 * SEG_NBAND) {
            const int neg = tid / SEG_NBAND, t = tid % SEG_NBAND;
            const int blo = neg ? -(t * G.q) - G.s : t * G.q;
            uint32_t m1 = 0, b1 = 256, m2 = 0;
            for (int v = blo; v <= blo + G.s; v++) {
                const uint32_t h = hiG[v & 255];
                if (b1 == 256u || h > m1) { m2 = b1 == 256u ? 0u : m1; m1 = h; b1 = (uint32_t)(v & 255); }
                else if (h > m2) m2 = h;
            }
            /* ">=" semantics: a second bin as large as the first must count as "other" for the first as well */
            uint32_t *bt = btop + ((size_t)neg * SEG_NBAND + t) * 4;
            bt[0] = m1; bt[1] = b1; bt[2] = m2; bt[3] = 0u;
        }
    }
    PLS_SYNC();
    PLS_THREADS(tid, SEG_THREADS) {
        for (int d = tid; d < NPX * 4; d += SEG_THREADS) {
            const int r = seg_validate_one(V, d, 0);
            pend[d] = (uint8_t)(r == 2);
            if (r == 0) PLS_ATOMIC_MIN(&red[8], (xg0 + (uint32_t)(d >> 2)) * 4u + (uint32_t)(d & 3));
        }
    }
    PLS_SYNC();
    if (prof) tk[2] = PLS_CLOCK();
    if (wbits[8]) {
        PLS_THREADS(tid, SEG_THREADS) {
            if (tid == 0) {
                int ns = 0;
                red[13] = wbits[8];
                for (int b = 0; b < 256 && ns < NW; b++) if ((wbits[b >> 5] >> (b & 31)) & 1u) { slot_of[b] = (uint8_t)ns; wbits[9 + ns] = (uint32_t)b; ns++; }
                wbits[8] = (uint32_t)ns;
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, SEG_THREADS) {
            const int sl = tid / NW, slot = tid % NW;
            if (sl < VGRP && slot < (int)wbits[8]) {
                /* bumps of the slot's bin in front of every decision of segment sl, four decisions per word in and out */
                const uint32_t b = wbits[9 + slot];
                const uint32_t *src = (const uint32_t *)(binb + sl * SEG_BINB_STRIDE);
                uint32_t *dst = pcw + (size_t)slot * PCS + sl * SEG_PC_SEG;
                const int e0seg = seg_max(0, ((int)sx - (int)xg0 - sl * SEG_L) * 4);          /* decisions of the segment in front of the epoch do not count */
                uint32_t run = 0;
                for (int iw = 0; iw < SEG_L; iw++) {
                    const uint32_t w = src[iw];
                    uint32_t out = 0;
                    for (int t = 0; t < 4; t++) {
                        out |= (run & 255u) << (8 * t);
                        if (iw * 4 + t >= e0seg && (uint32_t)t < bpp && ((w >> (8 * t)) & 255u) == b) run++;
                    }
                    dst[iw] = out;
                }
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, SEG_THREADS) {
            for (int d = tid; d < NPX * 4; d += SEG_THREADS) {
                if (!pend[d]) continue;
                if (seg_validate_one(V, d, 1) == 0) PLS_ATOMIC_MIN(&red[8], (xg0 + (uint32_t)(d >> 2)) * 4u + (uint32_t)(d & 3));
            }
        }
    }
    }
    if (prof) tk[3] = PLS_CLOCK();
    /* the left bytes the row sums took for the pixel in front of this replay group (seg_row_sums) against what was written there */
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < 4 && (uint32_t)tid < bpp && seg0 == segp && xg0 > sx && xg0 > 0u) {
            if (seg_cand_byte(cw[1 * 4 + tid]) != (int)((red[9] >> (8 * tid)) & 255u)) PLS_ATOMIC_MIN(&red[8], xg0 * 4u + (uint32_t)tid);
        }
    }
    PLS_SYNC();
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid == 0) {
            SegAcc &A = j.acc[par];
            if (red[8] != SEG_NOFAIL) { PLS_ATOMIC_MIN(&A.fail[f], red[8]); PLS_ATOMIC_OR(&A.failmask, 1u << f); PLS_ATOMIC_OR(&j.self->vfail[par], 1u << f); }
            if (prof) {
                tk[4] = PLS_CLOCK(); tk[5] = tk[4];
                for (int q = 0; q < 5; q++) { PLS_ATOMIC_MAX(&j.result[40 + q], (int32_t)(tk[q + 1] - tk[q])); PLS_ATOMIC_ADD((uint32_t *)&j.result[48 + q], (uint32_t)(tk[q + 1] - tk[q])); }
                PLS_ATOMIC_ADD((uint32_t *)&j.result[53], 1u);
                PLS_ATOMIC_ADD((uint32_t *)&j.result[46], red[13]);                     /* pending decisions (pass 3) */
            }
        }
    }
}

/* ---- CONTROL -------------------------------------------------------------------------------------------------------------- */
struct SegDecision {
    int kind, winner, dropped_none, start_none, keep_lazy;
    uint32_t failed;                /* bit f: candidate f failed validation in the attempt just finished */
    uint64_t cost[SEG_NFILT];
    uint32_t ignore, pad_;          /* bit f: whether candidate f's row of that attempt is valid does not matter to this decision */
};
static_assert(sizeof(SegDecision) / 4 <= 20, "seg_decide_wg shares it through 20 words");

/* what the attempt that just finished (control block `cur`, sums `A`) means -- in two steps, so that five lanes can look at a
 * candidate each before one lane draws the conclusion.  Step 1, candidate f: its row cost, or why it has none yet. */
static_assert(offsetof(SegDecision, cost) == 24, "seg_ctl_body reads cost[f] out of the shared copy by word index");
struct SegCandDec { uint64_t cost; uint32_t state; uint32_t pad_; };      /* state 0: cost is final (or ~0: no acceptable row), 1: lazy (none, not run yet), 2: failed validation */
/* optimistic: the validation of the attempt is still out (it runs next to this decision): every row is taken for valid */
template <class CT, class AT>
PLS_HD SegCandDec seg_decide_cand(const SegJob &j, const SegParams &P, CT &cur, AT &A, int f, uint32_t ecost, bool optimistic)
{
    SegCandDec r; r.cost = ~0ull; r.state = 0; r.pad_ = 0;
    if (!cur.active[f]) { r.cost = cur.cost[f]; return r; }
    if (cur.active[f] == 2) { r.state = 1; return r; }                             /* candidate none, not run yet */
    if (!optimistic && A.fail[f] != SEG_NOFAIL) { r.state = 2; return r; }
    const bool adaptive = !j.row_filters || cur.y == 0;
    uint64_t cst = A.derr[f] / 128u + ecost;                                      /* optimize_state.c:360 (ecost: seg_entropy_costs) */
    if (adaptive) {
        int bestg = 0;
        for (int g = 1; g < SEG_NFILT; g++) if (A.hs[f][g] < A.hs[f][bestg]) bestg = g;
        if (bestg != f) cst = ~0ull;                                              /* optimize_state.c:319-324 */
    }
    if (P.engine_flags >> 8) cst = f == (P.engine_flags >> 8) - 1 ? 0ull : ~0ull;
    r.cost = cst;
    return r;
}
/* Step 2: the conclusion.  Every workgroup of the control kernel comes to the same one. */
template <class CT, class AT, class DT>
PLS_HD SegDecision seg_decide_combine(const SegJob &j, const SegParams &P, int attempt, CT &cur, AT &A, DT *cd, bool optimistic)
{
    SegDecision D;
    D.kind = SEG_K_INIT; D.winner = -1; D.failed = 0; D.dropped_none = 0; D.start_none = 0; D.keep_lazy = 0; D.ignore = 0; D.pad_ = 0;
    for (int f = 0; f < SEG_NFILT; f++) D.cost[f] = ~0ull;
    if (attempt == 0) return D;
    if (cur.finished) { D.kind = SEG_K_FINISHED; return D; }
    bool any_failed = false;
    bool lazy0 = false;
    for (int f = 0; f < SEG_NFILT; f++) {
        if (cd[f].state == 1) { lazy0 = true; continue; }
        if (cd[f].state == 2) { D.failed |= 1u << f; any_failed = true; continue; }
        D.cost[f] = cd[f].cost;
    }
    const bool none_unsure = optimistic && cur.active[0] == 1u;     /* none was run, and whether its row is valid is not known yet */
    if (((D.failed & 1u) || lazy0 || none_unsure) && !(P.engine_flags >> 8)) {
        /* Candidate none failed validation, or has not been run at all (lazy).  Its row cost is at least none_lb (seg_post_body); it has
         * the lowest index, so it wins ties (pngloss_image.c:257) and loses only to a strictly cheaper row: if one exists already, none
         * cannot be the winner whatever its exact row would be -- running it (again) is not worth it.  (The winner's row, histogram and
         * errors are all that is committed.) */
        uint64_t best_other = ~0ull;
        for (int f = 1; f < SEG_NFILT; f++) if (!((D.failed >> f) & 1u) && D.cost[f] < best_other) best_other = D.cost[f];
        if (A.lb_valid == j.ngrp && best_other < A.none_lb) {
            D.failed &= ~1u; D.cost[0] = ~0ull; D.dropped_none = 1; lazy0 = false;
            D.ignore |= 1u;
            any_failed = D.failed != 0;
        }
    }
    if (lazy0) {
        if (any_failed) D.keep_lazy = 1;             /* others still have epochs to run: none can wait for their costs */
        else { D.start_none = 1; any_failed = true; }/* the bound does not rule none out: run it now */
    }
    if (any_failed) { D.kind = SEG_K_RESTART; return D; }
    uint64_t best = ~0ull;
    for (int f = 0; f < SEG_NFILT; f++) if (D.cost[f] < best) { best = D.cost[f]; D.winner = f; }   /* strict <: pngloss_image.c:257 */
    if (D.winner >= 0) D.kind = SEG_K_COMMIT;
    else D.kind = cur.s == 0 ? SEG_K_ABORT : SEG_K_RETRY;                           /* pngloss_image.c:266-274 */
    return D;
}
/* ---- what a candidate's row has bumped, and what that costs ------------------------------------------------------------------------
 * spec[w][b] (w < SEG_NFILT) = every bump of bin b by candidate w's row: the validated prefix (base) + the groups from its epoch's first on
 * (the replay's counts); spec[SEG_NFILT][b] = the committed histogram.  Requested before the decision is known (it says which candidate's
 * bumps become the next histogram), by `nt` loader lanes t = 0 .. nt-1, NSP = ceil(1536 / nt) items each: requests, then stores. */
template <int NSP> struct SegSpecRegs { uint32_t bs[NSP], g0[NSP][SEG_NG_BURST], wsx[NSP]; };
template <int NSP>
PLS_HD void seg_spec_request(const SegJob &j, int prev, const SegCtl &curg, int t, int nt, SegSpecRegs<NSP> &r)
{
    const uint32_t ngrp = j.ngrp;
    PLS_UNROLL
    for (int q = 0; q < NSP; q++) {
        const int i = t + q * nt, w = i >> 8, b = i & 255;
        r.bs[q] = 0; r.wsx[q] = 0;
        PLS_UNROLL
        for (int g = 0; g < SEG_NG_BURST; g++) r.g0[q][g] = 0u;
        if (i < (SEG_NFILT + 1) * 256) {
            if (w == SEG_NFILT) r.bs[q] = j.H0[prev * 256 + b];
            else {
                r.bs[q] = j.base[((size_t)prev * SEG_NFILT + w) * 256 + b];
                r.wsx[q] = curg.start_x[w];
                PLS_UNROLL
                for (int g = 0; g < SEG_NG_BURST; g++) if ((uint32_t)g < ngrp) r.g0[q][g] = j.grpcnt[((size_t)w * ngrp + g) * 256 + b];
            }
        }
    }
}
template <int NSP>
PLS_HD void seg_spec_store(const SegJob &j, int t, int nt, const SegSpecRegs<NSP> &r, seg_lds_u32 spec)
{
    const uint32_t ngrp = j.ngrp, W = j.W;
    PLS_UNROLL
    for (int q = 0; q < NSP; q++) {
        const int i = t + q * nt, w = i >> 8;
        if (i < (SEG_NFILT + 1) * 256) {
            uint32_t v = r.bs[q];
            if (w != SEG_NFILT) {
                const uint32_t wfg = (r.wsx[q] / SEG_L) / SEG_GRP;
                PLS_UNROLL
                for (int g = 0; g < SEG_NG_BURST; g++) v += ((uint32_t)g >= wfg && (uint32_t)g < ngrp && r.wsx[q] < W) ? r.g0[q][g] : 0u;
                for (uint32_t g = SEG_NG_BURST; g < ngrp; g++) v += (g >= wfg && r.wsx[q] < W) ? j.grpcnt[((size_t)w * ngrp + g) * 256 + (i & 255)] : 0u;   /* (rows beyond 8192 pixels) */
            }
            spec[i] = v;
        }
    }
}
/* (rare) the same for the attempt TWO before, when the one before turned out void: control block, sums and spec once more, plainly */
PLS_HD void seg_ctl_reload(const SegJob &j, int prev, seg_lds_u32 ctlc, seg_lds_u32 accc, seg_lds_u32 spec, int nt)
{
    const uint32_t ngrp = j.ngrp, W = j.W;
    PLS_SYNC();
    PLS_THREADS(tid, nt) {
        if (tid < (int)(sizeof(SegCtl) / 4)) ctlc[tid] = ((const SEG_AS_GLB uint32_t *)&j.ctl[prev])[tid];
        if (tid >= 128 && tid < 128 + (int)(sizeof(SegAcc) / 4)) accc[tid - 128] = ((const SEG_AS_GLB uint32_t *)&j.acc[prev])[tid - 128];
        for (int i = tid; i < (SEG_NFILT + 1) * 256; i += nt) {
            const int w = i >> 8, b = i & 255;
            uint32_t v;
            if (w == SEG_NFILT) v = j.H0[prev * 256 + b];
            else {
                v = j.base[((size_t)prev * SEG_NFILT + w) * 256 + b];
                const uint32_t wsx = j.ctl[prev].start_x[w], wfg = (wsx / SEG_L) / SEG_GRP;
                for (uint32_t g = wfg; g < ngrp && wsx < W; g++) v += j.grpcnt[((size_t)w * ngrp + g) * 256 + b];
            }
            spec[i] = v;
        }
    }
    PLS_SYNC();
}

/* The entropy cost of every candidate row (optimize_state.c:326-342): a pixel is charged 64 - floor(log2 H[symbol]) = 33 + clz(H[symbol]) under
 * the histogram AFTER the row, and its symbol is the bin it bumped (:251-254 -- the stored byte minus the same prediction), so the row costs
 * sum over bins of n * (33 + clz(H0 + n)), n = the row's bumps of the bin: no pass over the pixels.  ecost: SEG_NFILT (8) words of shared
 * memory, zeroed by the caller in front of its last barrier. */
PLS_HD void seg_entropy_costs(seg_lds_u32 spec, seg_lds_u32 ecost, int nt)
{
    /* (ecost was zeroed in front of the barrier behind the burst) one wave per candidate, four bins a lane */
    PLS_THREADS(tid, nt) {
        if (tid < SEG_NFILT * 64) {
            const int f = tid >> 6, lane = tid & 63;
            uint32_t v = 0;
            PLS_UNROLL
            for (int q = 0; q < 4; q++) {
                const int b = lane + 64 * q;
                const uint32_t n = spec[f * 256 + b], h = spec[SEG_NFILT * 256 + b] + n;
                v += n ? n * (33u + (uint32_t)__builtin_clz(h)) : 0u;
            }
            v = pls_wave_sum_u32(v);
            if (PLS_WAVE_LEADER(tid) && v) PLS_ATOMIC_ADD(&ecost[f], v);
        }
    }
    PLS_SYNC();
}

/* the whole workgroup: lanes 0..4 step 1, lane 0 step 2; the result in shared memory (dshare: 16 words, cdw: 20 words in front of it) */
/* the control block, the sums and the decision in shared memory, with the address space in the type (ds_* instead of FLAT accesses) */
typedef SEG_AS_LDS const SegCtl seg_lds_ctl_t;
typedef SEG_AS_LDS const SegAcc seg_lds_acc_t;
/* the whole workgroup: lanes 0..4 step 1, lane 0 step 2; the result in shared memory (dshare: 16 words, cdw: 20 words behind it) */
PLS_HD SegDecision seg_decide_wg(const SegJob &j, const SegParams &P, int attempt, seg_lds_ctl_t &cur, seg_lds_acc_t &A, seg_lds_u32 ecost, seg_lds_u32 cdw, seg_lds_u32 dshare, int nt, bool optimistic)
{
    PLS_THREADS(tid, nt) {
        if (tid < SEG_NFILT && attempt) {
            const SegCandDec r = seg_decide_cand(j, P, cur, A, tid, ecost[tid], optimistic);
            cdw[4 * tid] = (uint32_t)r.cost; cdw[4 * tid + 1] = (uint32_t)(r.cost >> 32); cdw[4 * tid + 2] = r.state; cdw[4 * tid + 3] = 0u;
        }
    }
    PLS_SYNC();
    PLS_THREADS(tid, nt) {
        if (tid == 0) {
            SegCandDec cd[SEG_NFILT];
            for (int f = 0; f < SEG_NFILT; f++) { cd[f].cost = (uint64_t)cdw[4 * f] | ((uint64_t)cdw[4 * f + 1] << 32); cd[f].state = cdw[4 * f + 2]; cd[f].pad_ = 0; }
            const SegDecision D = seg_decide_combine(j, P, attempt, cur, A, cd, optimistic);
            uint32_t w[sizeof(SegDecision) / 4];
            memcpy(w, &D, sizeof D);
            for (int i = 0; i < (int)(sizeof(SegDecision) / 4); i++) dshare[i] = w[i];
        }
    }
    PLS_SYNC();
    uint32_t w[sizeof(SegDecision) / 4];
    for (int i = 0; i < (int)(sizeof(SegDecision) / 4); i++) w[i] = dshare[i];
    SegDecision D;
    memcpy(&D, w, sizeof D);
    return D;
}

/* decision tables of one candidate from a histogram (all threads of the workgroup; H, rank: 256 words each in shared memory;
 * out: SEG_TBL_WORDS words in global memory; stage: SEG_TBL_WORDS words of shared memory).  One lane per band, sign and direction
 * scans its band once (prefix leaders upwards, suffix leaders downwards).
 * Tie classes are band local: cls[sgn][b] = offset inside its band (of that sign) of the FIRST bin with the same (H, rank) as bin b,
 * so two bins of one band have equal classes iff their keys are equal -- all the tie test of seg_step_fast needs (the original
 * symbol and the leader it may replace always lie in the same band). */
PLS_HD void seg_build_tables(SEG_AS_GLB uint32_t *out, seg_lds_u32 H, seg_lds_u32 rank, seg_lds_u32 scratch, seg_lds_u32 stage, int s, int q, int nt, int part, int nparts, int32_t *profslots = nullptr)
{
    unsigned long long tb[4] = { 0, 0, 0, 0 };
    if (profslots) tb[0] = PLS_CLOCK();
    seg_lds_u8 cls = (seg_lds_u8)(stage + 4 * SEG_TN);        /* [2][256] */
    /* [256] (H << 32 | rank) of a bin: one 64-bit read and one compare per bin (scratch: 512 words) */
    SEG_AS_LDS uint64_t *K = (SEG_AS_LDS uint64_t *)scratch;
    PLS_THREADS(tid, nt) { for (int b = tid; b < 256; b += nt) K[b] = ((uint64_t)H[b] << 32) | rank[b]; }
    PLS_SYNC();
    if (profslots) tb[1] = PLS_CLOCK();
    PLS_THREADS(tid, nt) {
        for (int i = tid; i < 512; i += nt) {
            /* class of bin b in the band system of sign sgn: v = the value of the bin in that system */
            const int sgn = i >> 8, b = i & 255;
            const int v = sgn ? (b ? b - 256 : 0) : b;                    /* negative system: bins 1..255 are v = b - 256, bin 0 is v = 0 */
            const int t = (sgn ? -v : v) / q;
            const int blo = sgn ? -(t * q) - s : t * q;
            const uint64_t kb = K[b];
            int first = v - blo;
            /* four reads in flight per turn (indices past the range are clamped onto its end: they change nothing) */
            for (int u = v - 1; u >= blo; u -= 4) {
                const int u1 = seg_max(u - 1, blo), u2 = seg_max(u - 2, blo), u3 = seg_max(u - 3, blo);
                const uint64_t k0 = K[u & 255], k1 = K[u1 & 255], k2 = K[u2 & 255], k3 = K[u3 & 255];
                if (k0 == kb) first = u - blo;
                if (k1 == kb) first = u1 - blo;
                if (k2 == kb) first = u2 - blo;
                if (k3 == kb) first = u3 - blo;
            }
            cls[i] = (uint8_t)first;
        }
    }
    PLS_SYNC();
    if (profslots) tb[2] = PLS_CLOCK();
    /* one entry per thread and turn: the leader of [bandlo, v] (prefix) or [v, bandhi] (suffix) = largest (H, rank), lowest v among equals */
    PLS_THREADS(tid, nt) {
        /* (the workgroups that share a candidate's tables take every nparts-th block of nt entries) */
        for (int i = part * nt + tid; i < 4 * SEG_TN; i += nt * nparts) {
            const int dir = i / (2 * SEG_TN), sgn = (i / SEG_TN) & 1, v = i % SEG_TN - SEG_TOFF;
            uint32_t e = 0u;
            if (sgn ? v <= 0 : v >= 0) {
                const int t = (sgn ? -v : v) / q;
                const int blo = sgn ? -(t * q) - s : t * q, bhi = blo + s;
                const int ua = dir ? v : blo, ue = dir ? bhi : v;
                int L = ua; uint64_t bk = K[ua & 255];
                for (int u = ua + 1; u <= ue; u += 4) {
                    const int u1 = seg_min(u + 1, ue), u2 = seg_min(u + 2, ue), u3 = seg_min(u + 3, ue);
                    const uint64_t k0 = K[u & 255], k1 = K[u1 & 255], k2 = K[u2 & 255], k3 = K[u3 & 255];
                    if (k0 > bk) { L = u; bk = k0; }
                    if (k1 > bk) { L = u1; bk = k1; }
                    if (k2 > bk) { L = u2; bk = k2; }
                    if (k3 > bk) { L = u3; bk = k3; }
                }
                e = (uint32_t)(L + 1024) | ((uint32_t)cls[sgn * 256 + (L & 255)] << 16);     /* the class of the leader's key: the first equal bin of its band */
            }
            out[i] = e;
        }
        if (part == 0) for (int i = tid; i < SEG_TBL_WORDS - 4 * SEG_TN; i += nt) out[4 * SEG_TN + i] = stage[4 * SEG_TN + i];
    }
    PLS_SYNC();
    if (profslots) { PLS_THREADS(tid, nt) { if (tid == 0) { tb[3] = PLS_CLOCK(); for (int q = 0; q < 3; q++) PLS_ATOMIC_ADD((uint32_t *)&profslots[q], (uint32_t)(tb[q + 1] - tb[q])); } } }
}

/* histogram the coming attempt starts from, for the decisions that begin a row attempt afresh (not SEG_K_RESTART):
 * INIT: zero; RETRY / ABORT: the committed histogram; COMMIT: committed histogram + every bump of the winner's row */
PLS_HD void seg_next_hist(const SegDecision &D, seg_lds_u32 spec, seg_lds_u32 Hn, int nt)
{
    PLS_SYNC();
    PLS_THREADS(tid, nt) {
        for (int b = tid; b < 256; b += nt) {
            uint32_t v = 0u;
            if (D.kind == SEG_K_RETRY || D.kind == SEG_K_ABORT) v = spec[SEG_NFILT * 256 + b];
            else if (D.kind == SEG_K_COMMIT) v = spec[SEG_NFILT * 256 + b] + spec[D.winner * 256 + b];
            Hn[b] = v;
        }
    }
    PLS_SYNC();
}

/* ---- commit of the winner's row (pngloss_image.c:277-308), parallel over x: workgroup cw takes pixels [cw * SEG_COMMIT_W, ...) ----
 * Which candidate won is known only after the control block and the sums of the finished attempt have arrived; the candidate words
 * of ALL five are requested in the same burst (they do not depend on the decision), so the winner's are there when it is known.
 * SEG_COMMIT_W lanes work (one wave per SIMD); the other waves of the launch shape only keep the barriers company. */
PLS_HD void seg_ctl_commit(const SegJob &j, const SegParams &P, int par, int cw, unsigned char *smem)
{
    const int k1 = seg_k_prev(par), k2 = seg_k_prev2(par);
    const uint32_t W = j.W, H = j.H, bpp = j.bpp;
    seg_lds_u32 lutb = (seg_lds_u32)smem + 8;                  /* [512] next-rows terms of the split */
    seg_lds_u32 ctlc = lutb + 512, accc = ctlc + (sizeof(SegCtl) + 7) / 8 * 2, dshare = accc + (sizeof(SegAcc) + 7) / 8 * 2;
    seg_lds_u32 ecost = dshare + 40;                           /* [8] entropy cost of every candidate row; [48]: failmask of the attempt two before */
    seg_lds_u32 cw5 = dshare + 56;                             /* [SEG_NFILT][(SEG_COMMIT_W + 4)][4] every candidate's words of this workgroup's pixels, two more on either side; the winner's become their terms */
    seg_lds_u32 ext = cw5 + SEG_NFILT * (SEG_COMMIT_W + 4) * 4;/* [2][SEG_COMMIT_W][2]: err1 of either row parity (which row is committed is in the control block these loads ride along with) */
    seg_lds_u32 spec = ext + SEG_COMMIT_W * 4;                 /* [SEG_NFILT + 1][256] the bumps of every candidate's row, the committed histogram (for the row costs: seg_entropy_costs) */
    const uint32_t xw0 = (uint32_t)cw * SEG_COMMIT_W;
    const bool prof = (P.engine_flags & 1) != 0;
    unsigned long long tc0 = 0;
    if (prof) tc0 = PLS_CLOCK();
    /* (which attempt this one follows: see seg_ctl_body) */
    {
        const SegCtl &curg = j.ctl[k1];
        const SegAcc &Ag = j.acc[k1];
        PLS_THREADS(tid, SEG_THREADS) {
            if (tid < SEG_COMMIT_W) {
                const uint32_t x = xw0 + (uint32_t)tid;
                const uint32_t cword = tid < (int)(sizeof(SegCtl) / 4) ? ((const uint32_t *)&curg)[tid] : 0u;
                const uint32_t aword = (tid >= 128 && tid < 128 + (int)(sizeof(SegAcc) / 4)) ? ((const uint32_t *)&Ag)[tid - 128] : 0u;
                const uint32_t fmw = tid == 0 ? j.acc[k2].failmask : 0u;
                const uint32_t lb0 = P.lut_b[tid], lb1 = P.lut_b[tid + SEG_COMMIT_W];
                const long xh = tid < 2 ? (long)xw0 - 2 + tid : (long)xw0 + SEG_COMMIT_W + (tid - 2);     /* (threads 0..3) the halo pixel */
                /* every request first, then the stores: a loop that loads and stores turn by turn waits for each load on its own */
                uint32_t w5[SEG_NFILT][4], h5[SEG_NFILT][4];
                const bool inrow = x < W, halo = tid < 4 && xh >= 0 && xh < (long)W;
                PLS_UNROLL
                for (int f = 0; f < SEG_NFILT; f++) {
                    PLS_UNROLL
                    for (int q = 0; q < 4; q++) {
                        w5[f][q] = inrow ? j.cand[((size_t)f * W + x) * 4 + q] : 0u;
                        h5[f][q] = halo ? j.cand[((size_t)f * W + (size_t)xh) * 4 + q] : 0u;
                    }
                }
                uint32_t e1v[4] = { 0, 0, 0, 0 };
                if (inrow) { e1v[0] = j.err1[2 * (size_t)x]; e1v[1] = j.err1[2 * (size_t)x + 1]; e1v[2] = j.err1[2 * ((size_t)W + x)]; e1v[3] = j.err1[2 * ((size_t)W + x) + 1]; }
                PLS_UNROLL
                for (int f = 0; f < SEG_NFILT; f++) {
                    PLS_UNROLL
                    for (int q = 0; q < 4; q++) {
                        cw5[(f * (SEG_COMMIT_W + 4) + tid + 2) * 4 + q] = w5[f][q];
                        if (tid < 4) cw5[(f * (SEG_COMMIT_W + 4) + (tid < 2 ? tid : SEG_COMMIT_W + tid)) * 4 + q] = h5[f][q];
                    }
                }
                ext[tid * 2 + 0] = e1v[0]; ext[tid * 2 + 1] = e1v[1]; ext[(SEG_COMMIT_W + tid) * 2 + 0] = e1v[2]; ext[(SEG_COMMIT_W + tid) * 2 + 1] = e1v[3];
                lutb[tid] = lb0; lutb[tid + SEG_COMMIT_W] = lb1;
                if (tid == 0) dshare[48] = fmw;
                if (tid < 8) ecost[tid] = 0u;
                if (tid < (int)(sizeof(SegCtl) / 4)) ctlc[tid] = cword;
                if (tid >= 128 && tid < 128 + (int)(sizeof(SegAcc) / 4)) accc[tid - 128] = aword;
            } else {
                /* (the lanes that have no pixel) what every candidate's row bumped: the decision wants the row costs */
                constexpr int NLD = SEG_THREADS - SEG_COMMIT_W, NSP = ((SEG_NFILT + 1) * 256 + NLD - 1) / NLD;
                SegSpecRegs<NSP> sr;
                seg_spec_request<NSP>(j, k1, curg, tid - SEG_COMMIT_W, NLD, sr);
                seg_spec_store<NSP>(j, tid - SEG_COMMIT_W, NLD, sr, spec);
            }
        }
    }
    PLS_SYNC();
    const bool redo = ((seg_lds_ctl_t *)ctlc)->magic == SEG_MAGIC && (dshare[48] & ~((seg_lds_ctl_t *)ctlc)->ignore) != 0u;
    if (redo) seg_ctl_reload(j, k2, ctlc, accc, spec, SEG_THREADS);
    unsigned long long tk[4] = { 0, 0, 0, 0 };
    if (prof) tk[0] = PLS_CLOCK();
    seg_lds_ctl_t &cur = *(seg_lds_ctl_t *)ctlc;
    seg_lds_acc_t &A = *(seg_lds_acc_t *)accc;
    const bool fresh = cur.magic != SEG_MAGIC;                 /* the image's first attempt: nothing behind it (what the burst read is junk) */
    const int attempt = fresh ? 0 : 1;
    if (fresh) {
        /* the first row: its originals into rowcopy; nothing has been diffused into it */
        PLS_THREADS(tid, SEG_THREADS) {
            if (tid < SEG_COMMIT_W) {
                const uint32_t x = xw0 + (uint32_t)tid;
                if (x < W && H) {
                    seg_row_orig(j, 0u)[x] = j.img[x];
                    SEG_AS_GLB uint32_t *e00 = seg_e0(j, 0u), *e10 = seg_e1(j, 0u);
                    e00[2 * (size_t)x] = 0u; e00[2 * (size_t)x + 1] = 0u; e10[2 * (size_t)x] = 0u; e10[2 * (size_t)x + 1] = 0u;
                }
            }
        }
        return;
    }
    seg_entropy_costs(spec, ecost, SEG_THREADS);
    const SegDecision D = seg_decide_wg(j, P, attempt, cur, A, ecost, dshare + 20, dshare, SEG_THREADS, !redo);
    if (D.kind != SEG_K_COMMIT) return;
    if (prof) tk[1] = PLS_CLOCK();
    const int winner = D.winner;
    seg_lds_u32 cwt = cw5 + winner * (SEG_COMMIT_W + 4) * 4;
    const uint32_t keep = bpp >= 4 ? 0xffffffffu : ((1u << (8 * bpp)) - 1u);
    const uint32_t y = cur.y, ynext = y + 1;
    uint32_t *rowp = j.img + (size_t)y * W;
    SEG_AS_GLB uint32_t *e0n = seg_e0(j, ynext), *e1n = seg_e1(j, ynext);
    seg_lds_u32 npx = cw5 + ((winner + 1) % SEG_NFILT) * (SEG_COMMIT_W + 4) * 4;     /* (a loser's tile) the row's new pixel */
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < SEG_COMMIT_W) {
            /* the row's new pixel, then candidate words -> the next-rows terms of their differences, once per pixel and channel */
            seg_lds_u32 mine = cwt + (tid + 2) * 4;
            npx[tid] = ((mine[0] & 255u) | ((mine[1] & 255u) << 8) | ((mine[2] & 255u) << 16) | ((mine[3] & 255u) << 24)) & keep;
            for (int q = 0; q < 4; q++) mine[q] = seg_terms_lds(lutb, P.bleed, seg_cand_diff(mine[q]));
            if (tid < 4) { seg_lds_u32 halo = cwt + (tid < 2 ? tid : SEG_COMMIT_W + tid) * 4; for (int q = 0; q < 4; q++) halo[q] = seg_terms_lds(lutb, P.bleed, seg_cand_diff(halo[q])); }
        }
    }
    PLS_SYNC();
    if (prof) tk[2] = PLS_CLOCK();
    PLS_THREADS(tid, SEG_THREADS) {
        if (tid < SEG_COMMIT_W) {
            const uint32_t x = xw0 + (uint32_t)tid;
            if (x < W) {
                const uint32_t onext = ynext < H ? j.img[(size_t)ynext * W + x] : 0u;
                const uint32_t e1[2] = { ext[((y & 1u) * SEG_COMMIT_W + tid) * 2 + 0], ext[((y & 1u) * SEG_COMMIT_W + tid) * 2 + 1] };
                /* error rows: err0'[x] = err1[x] + t(x+2)+f(x+1)+v(x)+f(x-1)+t(x-2), err1'[x] = t(x+1)+h(x)+t(x-1) (optimize_state.c:446-465) */
                uint32_t n0[4], n1[4];
                for (int p = 0; p < 4; p++) {
                    const int ch = seg_channel_of_plane(bpp, p);
                    int c1 = 0, c2 = 0;
                    if (ch >= 0)
                        for (int dx = -2; dx <= 2; dx++) {
                            const long sxp = (long)x + dx;
                            if (sxp < 0 || sxp >= (long)W) continue;
                            const uint32_t e = cwt[(tid + 2 + dx) * 4 + ch];
                            const int T_ = (int)(int8_t)(e & 255u), F_ = (int)(int8_t)((e >> 8) & 255u), V_ = (int)(int8_t)((e >> 16) & 255u), H_ = (int)e >> 24;
                            const int ad = dx < 0 ? -dx : dx;
                            c1 += ad == 2 ? T_ : (ad == 1 ? F_ : V_);
                            if (ad <= 1) c2 += ad == 1 ? T_ : H_;
                        }
                    n0[p] = (uint32_t)(seg_err_plane(e1, p) + c1) & 0xffffu;     /* int16 wrap-on-store */
                    n1[p] = (uint32_t)c2 & 0xffffu;
                }
                /* the image row in place (its originals stay in rowcopy for the validation that runs next to this, and for a repetition);
                 * the coming row's originals and incoming errors into the copies of ITS parity */
                rowp[x] = npx[tid];
                e0n[2 * (size_t)x] = n0[0] | (n0[1] << 16); e0n[2 * (size_t)x + 1] = n0[2] | (n0[3] << 16);
                e1n[2 * (size_t)x] = n1[0] | (n1[1] << 16); e1n[2 * (size_t)x + 1] = n1[2] | (n1[3] << 16);
                if (ynext < H) seg_row_orig(j, ynext)[x] = onext;
            }
            if (cw == 0 && tid == 0) {
                if (j.row_filters) j.row_filters[y] = (uint8_t)(0x08u << winner);          /* PNG_FILTER_* flags, pngloss_image.c:288-308 */
                j.row_ids[y] = (uint8_t)winner;
            }
        }
    }
    if (prof) { PLS_THREADS(tid, SEG_THREADS) { if (tid == 0) { const uint32_t dt = (uint32_t)(PLS_CLOCK() - tc0); PLS_ATOMIC_MAX(&j.result[58], (int32_t)dt); PLS_ATOMIC_ADD((uint32_t *)&j.result[62], dt); PLS_ATOMIC_ADD((uint32_t *)&j.result[63], 1u); PLS_ATOMIC_ADD((uint32_t *)&j.result[23], (uint32_t)(tk[0] - tc0)); PLS_ATOMIC_ADD((uint32_t *)&j.result[45], (uint32_t)(tk[1] - tk[0])); PLS_ATOMIC_ADD((uint32_t *)&j.result[54], (uint32_t)(tk[2] - tk[1])); PLS_ATOMIC_ADD((uint32_t *)&j.result[55], (uint32_t)(PLS_CLOCK() - tk[2])); } } }
}

/* Control kernel of an attempt (par = its parity): reads what the attempt before left (control block and sums of parity prev), writes the control block
 * of parity par.  bx < SEG_NFILT: candidate bx (epoch setup, decision tables); bx == SEG_NFILT: the image-wide fields;
 * bx > SEG_NFILT: commit of pixels [(bx - SEG_NFILT - 1) * SEG_THREADS, ...) */
template <int TPARTS>
PLS_HD void seg_ctl_body(const SegJob &j, const SegParams &P, int par, int bx, unsigned char *smem)
{
    constexpr int SEG_CTL_IMG = SEG_NFILT * TPARTS;          /* (= SEG_CTL_IMG_OF(P): the launcher picks the instantiation by SegParams::tparts) */
    if (bx > SEG_CTL_IMG) { seg_ctl_commit(j, P, par, bx - SEG_CTL_IMG - 1, smem); return; }
    /* WHICH attempt this one follows.  Normally the one before it (copy k1), whose validation is still running -- in this very launch
     * (seg_k_ctl carries the validation workgroups of the attempt before next to the control workgroups of this one) -- so the decision
     * is taken OPTIMISTICALLY: every candidate row is assumed valid (they are in all but a handful of rows per image), the winner is
     * committed and the next row prepared while the proof is still out.  Everything the validation (and a repetition of the row)
     * reads is left alone: the commit writes the image row in place but the row's originals live in rowcopy, the error rows and the
     * row's extremes are double buffered by row parity, control block / sums / histogram / prefix bumps are kept three deep.
     * If that proof FAILED -- failmask of the attempt two before (k2), known by now, minus what the decision one before said it does not
     * depend on -- the attempt one before was void (its workgroups saw the same words and did nothing): this one follows the attempt
     * two before instead, with the failures known (an epoch for the failed candidates, as ever).  The words that tell ride along with
     * the burst of loads; the rare repetition pays a second burst. */
    const int k1 = seg_k_prev(par), k2 = seg_k_prev2(par);
    SegCtl &nxt = j.ctl[par];
    const uint32_t W = j.W, H = j.H, bpp = j.bpp, nseg = j.nseg, ngrp = j.ngrp;
    seg_lds_u32 Hn = (seg_lds_u32)smem, rank = Hn + 256, scratch = Hn + 512, basen = Hn + 768;   /* 4 x 256 words */
    seg_lds_u32 stage = Hn + 1024;                               /* SEG_TBL_WORDS: a candidate's tables before they go out; the commit workgroups keep the split table here */
    const bool prof = (P.engine_flags & 1) != 0;
    unsigned long long tc0 = 0;
    if (prof) tc0 = PLS_CLOCK();
    /* What the coming attempt's histogram may need is requested before the decision is known (it depends on which candidate won):
     * per candidate w, base[w] + every group's bumps of w's row, and the committed histogram; groups in front of w's epoch are masked
     * out afterwards.  spec[w][b] in the table staging area, which is free until the tables are built. */
    seg_lds_u32 spec = stage;                                  /* [SEG_NFILT + 1][256] */
    /* the finished attempt's control block and sums, copied into shared memory by the same burst of loads: everything below reads the copies */
    seg_lds_u32 ctlc = stage + (SEG_NFILT + 1) * 256, accc = ctlc + (sizeof(SegCtl) + 7) / 8 * 2;
    seg_lds_u32 dshare = accc + (sizeof(SegAcc) + 7) / 8 * 2;  /* [20] the decision, [20] the candidates' verdicts, [8] entropy costs, [4] which attempt is followed */
    int prev = k1;
    {
        const SegCtl &curg = j.ctl[k1];
        const SegAcc &Ag = j.acc[k1];
        PLS_THREADS(tid, SEG_THREADS) {
            /* every request of the burst first, the stores behind them (the image's first attempt reads junk here and ignores it) */
            const uint32_t cword = tid < (int)(sizeof(SegCtl) / 4) ? ((const uint32_t *)&curg)[tid] : 0u;
            const uint32_t aword = (tid >= 128 && tid < 128 + (int)(sizeof(SegAcc) / 4)) ? ((const uint32_t *)&Ag)[tid - 128] : 0u;
            const uint32_t rword = (bx < SEG_CTL_IMG && tid >= 256 && tid < 512) ? j.orig_rank[(bx / TPARTS) * 256 + (tid - 256)] : 0u;
            const uint32_t fmw = tid == 512 ? j.acc[k2].failmask : 0u;
            constexpr int NSP = ((SEG_NFILT + 1) * 256 + SEG_THREADS - 1) / SEG_THREADS;
            SegSpecRegs<NSP> sr;
            seg_spec_request<NSP>(j, k1, curg, tid, SEG_THREADS, sr);
            if (tid < (int)(sizeof(SegCtl) / 4)) ctlc[tid] = cword;
            if (tid >= 128 && tid < 128 + (int)(sizeof(SegAcc) / 4)) accc[tid - 128] = aword;
            if (bx < SEG_CTL_IMG && tid >= 256 && tid < 512) rank[tid - 256] = rword;
            if (tid == 512) dshare[48] = fmw;
            if (tid >= 520 && tid < 528) dshare[40 + tid - 520] = 0u;     /* (the entropy costs) */
            seg_spec_store<NSP>(j, tid, SEG_THREADS, sr, spec);
        }
    }
    PLS_SYNC();
    const bool redo = ((seg_lds_ctl_t *)ctlc)->magic == SEG_MAGIC && (dshare[48] & ~((seg_lds_ctl_t *)ctlc)->ignore) != 0u;
    if (redo) { prev = k2; seg_ctl_reload(j, k2, ctlc, accc, spec, SEG_THREADS); }
    unsigned long long tq1 = 0, tq2 = 0;
    if (prof) tq1 = PLS_CLOCK();
    seg_lds_ctl_t &cur = *(seg_lds_ctl_t *)ctlc;
    seg_lds_acc_t &A = *(seg_lds_acc_t *)accc;
    /* one lane decides, the workgroup reads the result (16 waves working it out side by side only take each other's issue slots) */
    /* the number of this attempt is kept on the device (the launcher passes the copy index only) */
    const int attempt = cur.magic != SEG_MAGIC ? 0 : (int)cur.attempts + (redo ? 2 : 1);
    seg_lds_u32 ecost = dshare + 40;                            /* [8] entropy cost of every candidate row */
    seg_entropy_costs(spec, ecost, SEG_THREADS);
    const SegDecision D = seg_decide_wg(j, P, attempt, cur, A, ecost, dshare + 20, dshare, SEG_THREADS, !redo);
    if (prof) tq2 = PLS_CLOCK();
    const uint32_t y = attempt ? cur.y : 0u;
    int s_next = attempt ? (int)cur.s : P.strength;
    if (D.kind == SEG_K_RETRY) s_next = (int)cur.s - 1;
    if (D.kind == SEG_K_COMMIT) s_next = P.strength;

    if (bx == SEG_CTL_IMG) {
        /* ---- the image-wide fields ---- */
        if (D.kind == SEG_K_FINISHED) {
            /* every row is committed.  finished == 1: the last row's validation was still out when that was written -- it has passed (else this
             * attempt would follow the one two before): the image is finished for good, and says so */
            PLS_THREADS(tid, SEG_THREADS) {
                if (tid < (int)(sizeof(SegAcc) / 4)) ((uint32_t *)&j.acc[par])[tid] = 0u;
                if (tid >= 256 && tid < 512 && cur.finished == 1u) Hn[tid - 256] = j.H0[prev * 256 + (tid - 256)];
            }
            PLS_SYNC();
            PLS_THREADS(tid, SEG_THREADS) {
                if (tid == 0) {
                    nxt.y = cur.y; nxt.s = cur.s; nxt.status = cur.status; nxt.finished = 2; nxt.retried = cur.retried; nxt.restarts_total = cur.restarts_total; nxt.attempts = (uint32_t)attempt; nxt.magic = SEG_MAGIC;
                    nxt.serial_rows = cur.serial_rows; nxt.dropped_none = cur.dropped_none; nxt.none_eager = cur.none_eager; nxt.ignore = 0u;
                    { SEG_AS_GLB SegViewRec &VW = j.self->v[par]; VW.finished = 2u; VW.y = cur.y; VW.s = cur.s; VW.ignore = 0u; VW.magic = SEG_MAGIC; }
                    j.self->vfail[par] = 0u;
                    if (j.attempt_word) PLS_HOST_VISIBLE_STORE(j.attempt_word, (uint32_t)attempt);
                    if (cur.finished == 1u) {
                        /* epilogue: final histogram + result record (pngloss_image.c:311-325) */
                        uint32_t nz = 0;
                        for (int b = 0; b < 256; b++) { j.final_hist[b] = Hn[b]; nz += Hn[b] != 0; }
                        for (int i = 0; i < 24; i++) if (i < 8 || i == 20 || (i != 17 && !(P.engine_flags & 1))) j.result[i] = 0;   /* (8..18: the chain kernel's phase clocks) */
                        j.result[0] = (int32_t)cur.status; j.result[1] = (int32_t)bpp; j.result[2] = (int32_t)nz; j.result[3] = (int32_t)cur.retried;
                        j.result[4] = (int32_t)cur.restarts_total; j.result[5] = (int32_t)cur.attempts; j.result[6] = (int32_t)cur.serial_rows; j.result[7] = (int32_t)cur.dropped_none; j.result[20] = 3;   /* engine id: segment-parallel */
                        if (j.done_counter) PLS_HOST_VISIBLE_ADD(j.done_counter, 1u);
                    }
                }
            }
            return;
        }
        if (D.kind != SEG_K_RESTART) seg_next_hist(D, spec, Hn, SEG_THREADS);
        PLS_THREADS(tid, SEG_THREADS) {
            for (int b = tid; b < 256; b += SEG_THREADS) j.H0[par * 256 + b] = D.kind == SEG_K_RESTART ? j.H0[prev * 256 + b] : Hn[b];
            /* zero the sums the coming attempt accumulates into */
            if (tid < (int)(sizeof(SegAcc) / 4)) ((uint32_t *)&j.acc[par])[tid] = 0u;
        }
        PLS_SYNC();
        PLS_THREADS(tid, SEG_THREADS) {
            if (tid < SEG_NFILT) j.acc[par].fail[tid] = SEG_NOFAIL;
            if (tid == 0) {
                uint32_t ny = y, fin = 0, st = attempt ? cur.status : 0u, retried = attempt ? cur.retried : 0u, rt = attempt ? cur.restarts_total : 0u;
                uint32_t ser = attempt ? cur.serial_rows : 0u, dropped = (attempt ? cur.dropped_none : 0u) + (uint32_t)D.dropped_none;
                if (D.kind == SEG_K_COMMIT) { ny = y + 1; if (ny >= H) fin = 1; }
                if (D.kind == SEG_K_RETRY) retried += cur.s == (uint32_t)P.strength ? 1u : 0u;
                if (D.kind == SEG_K_ABORT) { st = 65u; fin = 1; }                         /* pngloss_image.c:268-271 aborts here */
                SEG_DEBUG_ROW(D.kind, D.failed, D.winner, D.start_none);
                if (D.kind == SEG_K_RESTART)
                    for (int g = 0; g < SEG_NFILT; g++) if ((D.failed >> g) & 1u) { rt++; if (cur.restarts[g] + 1 > SEG_MAX_RESTARTS) ser++; }
                if (W == 0 || H == 0) fin = 1;
                nxt.y = ny; nxt.s = (uint32_t)(s_next < 0 ? 0 : s_next); nxt.status = st; nxt.finished = fin; nxt.retried = retried; nxt.restarts_total = rt;
                nxt.serial_rows = ser; nxt.attempts = (uint32_t)attempt; nxt.magic = SEG_MAGIC; nxt.dropped_none = dropped;
                {
                    const uint32_t ne = attempt ? cur.none_eager : 0u;
                    /* (eager only after none WON a row was tried on a screenshot where none never wins but its bound rarely rules it
                     * out: 1544 attempts instead of 941 -- the extra start per row costs more than the epochs of the eager rows) */
                    nxt.none_eager = D.start_none ? 16u : ((D.kind == SEG_K_COMMIT || D.kind == SEG_K_RETRY) && ne ? ne - 1u : ne);
                }
                if (j.progress && D.kind == SEG_K_COMMIT) PLS_HOST_VISIBLE_STORE(j.progress, ny);
                if (j.attempt_word) PLS_HOST_VISIBLE_STORE(j.attempt_word, (uint32_t)attempt);
                nxt.ignore = redo ? 0u : D.ignore;           /* (the epilogue waits for the last row's validation: the attempt that finds finished == 1) */
                { SEG_AS_GLB SegViewRec &VW = j.self->v[par]; VW.finished = fin; VW.y = ny; VW.s = (uint32_t)(s_next < 0 ? 0 : s_next); VW.ignore = redo ? 0u : D.ignore; VW.magic = SEG_MAGIC; }
                j.self->vfail[par] = 0u;
            }
        }
        return;
    }
    if (D.kind == SEG_K_FINISHED) return;

    /* ---- candidate f: SEG_TPARTS workgroups; all of them follow the decision and the new histogram, workgroup 0 writes the control
     *      fields, each builds its share of the decision tables ---- */
    const int f = bx / TPARTS, tpart = bx % TPARTS;
    const bool failed = (D.failed >> f) & 1u;
    if (D.kind != SEG_K_RESTART) {
        /* a fresh row attempt: start of the row, no validated prefix */
        seg_next_hist(D, spec, Hn, SEG_THREADS);
        /* (global stores wait until nothing is left to synchronise: a barrier behind a store waits for the store to arrive) */
        const int sn = s_next < 0 ? 0 : s_next;
        unsigned long long tc1 = 0;
        if (prof) tc1 = PLS_CLOCK();
        seg_build_tables(j.tables + (size_t)f * SEG_TBL_WORDS, Hn, rank, scratch, stage, sn, sn + 1, SEG_THREADS, tpart, TPARTS, prof ? &j.result[37] : nullptr);
        PLS_THREADS(tid, SEG_THREADS) {
            if (tpart == 0) {
                for (int b = tid; b < 256; b += SEG_THREADS) j.base[((size_t)par * SEG_NFILT + f) * 256 + b] = 0u;
                if (tid == 0) {
                    /* candidate none starts lazy: its cost bound first, the chain only if that cannot rule it out (seg_decide_combine) */
                    nxt.active[f] = (f == 0 && j.rowmm && !(P.engine_flags >> 8) && !(P.engine_flags & 2) && !(attempt && cur.none_eager)) ? 2u : 1u;
                    nxt.start_x[f] = 0; nxt.restarts[f] = 0; nxt.cost[f] = ~0ull;
                    j.self->v[par].active[f] = nxt.active[f]; j.self->v[par].start_x[f] = 0u;
                    for (int c = 0; c < 4; c++) nxt.state[f][c] = seg_state_pack(SegState{ 0, 0, 0 });
                }
            }
        }
        if (prof) { PLS_THREADS(tid, SEG_THREADS) { if (tid == 0) { const unsigned long long t2 = PLS_CLOCK(); PLS_ATOMIC_MAX(&j.result[56], (int32_t)(tc1 - tc0)); PLS_ATOMIC_MAX(&j.result[57], (int32_t)(t2 - tc1)); PLS_ATOMIC_ADD((uint32_t *)&j.result[59], (uint32_t)(tc1 - tc0)); PLS_ATOMIC_ADD((uint32_t *)&j.result[60], (uint32_t)(t2 - tc1)); PLS_ATOMIC_ADD((uint32_t *)&j.result[61], 1u); PLS_ATOMIC_ADD((uint32_t *)&j.result[19], (uint32_t)(tq1 - tc0)); PLS_ATOMIC_ADD((uint32_t *)&j.result[21], (uint32_t)(tq2 - tq1)); PLS_ATOMIC_ADD((uint32_t *)&j.result[22], (uint32_t)(tc1 - tq2)); } } }
        return;
    }
    if (f == 0 && cur.active[0] == 2 && (D.start_none || D.keep_lazy)) {
        PLS_THREADS(tid, SEG_THREADS) {
            for (int b = tid; b < 256; b += SEG_THREADS) { Hn[b] = j.H0[prev * 256 + b]; if (tpart == 0) j.base[((size_t)par * SEG_NFILT + f) * 256 + b] = 0u; }
            if (tid == 0 && tpart == 0) {
                nxt.active[0] = D.start_none ? 1u : 2u; nxt.start_x[0] = 0; nxt.restarts[0] = 0; nxt.cost[0] = ~0ull;
                j.self->v[par].active[0] = D.start_none ? 1u : 2u; j.self->v[par].start_x[0] = 0u;
                for (int c = 0; c < 4; c++) nxt.state[0][c] = seg_state_pack(SegState{ 0, 0, 0 });
            }
        }
        PLS_SYNC();
        if (D.start_none) seg_build_tables(j.tables + (size_t)f * SEG_TBL_WORDS, Hn, rank, scratch, stage, (int)cur.s, (int)cur.s + 1, SEG_THREADS, tpart, TPARTS);
        return;
    }
    if (tpart) return;                                        /* (an epoch inside the row is set up by one workgroup) */
    if (!failed) {
        /* nothing changes for this candidate: finished (now or earlier), its cost is kept */
        PLS_THREADS(tid, SEG_THREADS) {
            for (int b = tid; b < 256; b += SEG_THREADS) j.base[((size_t)par * SEG_NFILT + f) * 256 + b] = j.base[((size_t)prev * SEG_NFILT + f) * 256 + b];
            if (tid == 0) {
                nxt.active[f] = 0; nxt.start_x[f] = cur.start_x[f]; nxt.restarts[f] = cur.restarts[f];
                j.self->v[par].active[f] = 0u; j.self->v[par].start_x[f] = cur.start_x[f];
                for (int c = 0; c < 4; c++) nxt.state[f][c] = cur.state[f][c];
                nxt.cost[f] = (uint64_t)dshare[6 + 2 * f] | ((uint64_t)dshare[7 + 2 * f] << 32);     /* D.cost[f], from shared memory: indexing the copy in registers with f would send it to the stack */
            }
        }
        return;
    }
    /* epoch setup: bumps in front of the failing pixel xp, that pixel evaluated exactly, new start behind it */
    {
        const uint32_t xp = A.fail[f] >> 2, sx = cur.start_x[f], first = sx / SEG_L, fgrp = first / SEG_GRP;
        const uint32_t sgp = xp / SEG_L, gp = sgp / SEG_GRP;
        const bool serial = cur.restarts[f] + 1 > SEG_MAX_RESTARTS;
        const SegGeo G = seg_geo((int)cur.s);
        PLS_THREADS(tid, SEG_THREADS) {
            for (int b = tid; b < 256; b += SEG_THREADS) {
                uint32_t v = j.base[((size_t)prev * SEG_NFILT + f) * 256 + b];
                for (uint32_t g = fgrp; g < gp; g++) v += j.grpcnt[((size_t)f * ngrp + g) * 256 + b];
                for (uint32_t sg = (uint32_t)seg_max((int)(gp * SEG_GRP), (int)first); sg < sgp; sg++) v += j.segcnt[((size_t)f * nseg + sg) * 256 + b];
                basen[b] = v; Hn[b] = j.H0[prev * 256 + b];
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, SEG_THREADS) {
            if (tid == 0) {
                const uint32_t *row = seg_row_orig(j, y), *nab = y ? j.img + (size_t)(y - 1u) * W : nullptr, *e0g = seg_e0(j, y);
                uint32_t *cd = j.cand + (size_t)f * W * 4;
                for (uint32_t x = (uint32_t)seg_max((int)(sgp * SEG_L), (int)sx); x < xp; x++) for (uint32_t c = 0; c < bpp; c++) basen[seg_cand_bin(cd[(size_t)x * 4 + c])]++;
                const uint32_t xend = serial ? W : xp + 1;
                SegState *st = (SegState *)(uint32_t *)scratch;       /* [4], in shared memory: indexed by the channel, in registers it would go to the stack */
                for (uint32_t c = 0; c < bpp; c++) {
                    int rem1, thr1, rem2, thr2;
                    seg_rem_thr(P.lut_a, P.bleed, xp >= 1 ? seg_cand_diff(cd[(size_t)(xp - 1) * 4 + c]) : 0, rem1, thr1);
                    seg_rem_thr(P.lut_a, P.bleed, xp >= 2 ? seg_cand_diff(cd[(size_t)(xp - 2) * 4 + c]) : 0, rem2, thr2);
                    st[c].left = xp >= 1 ? seg_cand_byte(cd[(size_t)(xp - 1) * 4 + c]) : 0; st[c].cn = rem1 + thr2; st[c].th = thr1;
                }
                /* the reference's own order: channel after channel against the running histogram H0 + basen (optimize_state.c:212-254) */
                for (uint32_t x = xp; x < xend; x++)
                    for (uint32_t c = 0; c < bpp; c++) {
                        const SegPix p = seg_pix_load(row, nab, e0g, bpp, x, (int)c);
                        const uint32_t w = seg_step_scan(f, p, st[c], Hn, basen, rank, G, P.lut_a, P.bleed);
                        cd[(size_t)x * 4 + c] = w;
                        basen[seg_cand_bin(w)]++;
                    }
                for (uint32_t c = 0; c < 4; c++) nxt.state[f][c] = c < bpp ? seg_state_pack(st[c]) : 0u;
                nxt.start_x[f] = xend;
                nxt.active[f] = 1; nxt.restarts[f] = cur.restarts[f] + 1; nxt.cost[f] = ~0ull;
                j.self->v[par].active[f] = 1u; j.self->v[par].start_x[f] = xend;
            }
        }
        PLS_SYNC();
        PLS_THREADS(tid, SEG_THREADS) {
            for (int b = tid; b < 256; b += SEG_THREADS) {
                j.base[((size_t)par * SEG_NFILT + f) * 256 + b] = basen[b];
                Hn[b] += basen[b];                                                /* frozen histogram of the new epoch */
            }
        }
        PLS_SYNC();
        seg_build_tables(j.tables + (size_t)f * SEG_TBL_WORDS, Hn, rank, scratch, stage, G.s, G.q, SEG_THREADS, 0, 1);
    }
}

#endif /* PL_SEG_CORE_H */
