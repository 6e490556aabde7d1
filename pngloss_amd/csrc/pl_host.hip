/*
 * pl_host.hip -- the thin C-ABI shim of libpngloss_hip.so (see include/pngloss_hip.h for the contract).
 *
 * Host side only: context + workspace management, job tables, stream/event plumbing, and the host-pointer
 * drop-in entry points that replace /root/reference/src/pngloss_image.c:29-156.  No arithmetic of the hot path is
 * done here and there is no CPU fallback: without a usable HIP device every entry point fails loudly.
 */
#include "../../include/pngloss_hip.h"
#include "pl_device.h"
#include "pl_deflate.h"
#include "pl_pngread.h"
#include "pl_inflate.h"
#define SEG_PLAIN_POINTERS   /* host plumbing only: SegJob is filled here, never dereferenced */
#include "pl_seg.h"

#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#define PL_CHECK(expr)                                                                                           \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            std::fprintf(stderr, "pngloss_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                                              \
            return e_ == hipErrorOutOfMemory ? PNGLOSS_OUT_OF_MEMORY_ERROR : PNGLOSS_HIP_ERROR;                  \
        }                                                                                                        \
    } while (0)

#define SEG_MAX_GROUPS 8
/* Timing / test hooks of the environment, read ONCE, when a context is created (pngloss_hip_create) -- not per call.  None of them changes results: they pin choices the library
 * otherwise makes itself (launch groups, units, workgroup sizes), pick the blocking variant of the asynchronous entry, or print.  The one hook that is read per call is
 * PNGLOSS_HIP_ENGINE (the tests' pin of the row engine, read once at the top of enqueue; pngloss_hip_set_option(ctx, "engine", ..) takes precedence).  A hook that DOES change results
 * -- "candidate f wins every row", a debugging aid of rounds 1-3 -- exists in builds made with -DPL_DEBUG_FORCE_FILTER=f only: no environment variable of the shipped library can
 * make the drop-in seam write anything but the reference's bytes. */
struct PlHooks {
    int seg_groups = 0;          /* PNGLOSS_HIP_SEG_GROUPS: launch groups of a batch on the segment engine (0: the library's choice) */
    bool no_stream_wait = false; /* PNGLOSS_HIP_NO_STREAM_WAIT: the blocking variant of the asynchronous entry (rocprofv3 --pmc needs it) */
    int seg_unit = -1;           /* PNGLOSS_HIP_SEG_UNIT: 0 / 1 pins the enumeration per segment / in units (-1: the library's choice) */
    int tparts = 0;              /* PNGLOSS_HIP_SEG_TPARTS */
    int enum_nt = 0;             /* PNGLOSS_HIP_ENUM_NT: 512 / 1024 */
    int kin = -1;                /* PNGLOSS_HIP_KIN: run-in pixels of the seeded enumeration */
    int seg_seeds = -1;          /* PNGLOSS_HIP_SEG_SEEDS: 0 = units start from every state, as in round 5 (-1 / 1: from seeds where the pair has a seed set) */
    int seg_seeds1 = -1;         /* PNGLOSS_HIP_SEG_SEEDS1: 0 / 1 pins the per-segment enumeration from seeds (seg_k_enum_unit<1>; -1: batches of two or more images) */
    int pin = -1;                /* PNGLOSS_HIP_PIN: 0 = the launch thread is not pinned to a CPU (-1 / 1: pinned when the affinity set has room, run_seg_engine) */
    int calib = -1;              /* PNGLOSS_HIP_CALIB: 1 = the cost model that picks the row engine of a batch is calibrated on this device by a probe (engine_calib; off by default: see enqueue) */
    int seed_kin = -1;           /* PNGLOSS_HIP_SEED_KIN: run-in pixels of the units' seeds (1 .. SEG_SEED_KMAX) */
    bool segprof = false;        /* PNGLOSS_HIP_SEGPROF: phase clocks inside the kernels (slows them down) */
    bool debug = false;          /* PNGLOSS_HIP_DEBUG */
    bool debug_seam = false;     /* PNGLOSS_HIP_DEBUG_SEAM */
    bool force_careful = false;  /* PNGLOSS_HIP_FORCE_CAREFUL: the int16-wrap variant of the round-1 chains for every row (same bytes) */
    bool no_split = false;       /* PNGLOSS_HIP_NO_SPLIT */
    int split = 0;               /* PNGLOSS_HIP_SPLIT: chunks of a host window */
    static PlHooks from_env()
    {
        PlHooks h;
        auto num = [](const char *name, int dflt) { const char *e = std::getenv(name); return e ? std::atoi(e) : dflt; };
        if (std::getenv("PNGLOSS_HIP_SEG_GROUPS")) h.seg_groups = std::max(1, std::min(SEG_MAX_GROUPS, num("PNGLOSS_HIP_SEG_GROUPS", 0)));
        h.no_stream_wait = std::getenv("PNGLOSS_HIP_NO_STREAM_WAIT") != nullptr;
        if (std::getenv("PNGLOSS_HIP_SEG_UNIT")) h.seg_unit = num("PNGLOSS_HIP_SEG_UNIT", 0) != 0 ? 1 : 0;
        h.tparts = num("PNGLOSS_HIP_SEG_TPARTS", 0);
        h.enum_nt = num("PNGLOSS_HIP_ENUM_NT", 0);
        h.kin = num("PNGLOSS_HIP_KIN", -1);
        h.seg_seeds = num("PNGLOSS_HIP_SEG_SEEDS", -1);
        h.seg_seeds1 = num("PNGLOSS_HIP_SEG_SEEDS1", -1);
        h.pin = num("PNGLOSS_HIP_PIN", -1);
        h.calib = num("PNGLOSS_HIP_CALIB", -1);
        h.seed_kin = num("PNGLOSS_HIP_SEED_KIN", -1);
        h.segprof = std::getenv("PNGLOSS_HIP_SEGPROF") != nullptr;
        h.debug = std::getenv("PNGLOSS_HIP_DEBUG") != nullptr;
        h.debug_seam = std::getenv("PNGLOSS_HIP_DEBUG_SEAM") != nullptr;
        h.force_careful = std::getenv("PNGLOSS_HIP_FORCE_CAREFUL") != nullptr;
        { const char *no = std::getenv("PNGLOSS_HIP_NO_SPLIT"); h.no_split = no && *no == '1'; }
        { const int v = num("PNGLOSS_HIP_SPLIT", 0); if (v >= 1 && v <= 8) h.split = v; }
        return h;
    }
};
struct pngloss_hip_ctx {
    int device = 0;
    PlHooks hooks;                   /* (from the environment, at creation) */
    int pin_slot = 0;                /* which pair of CPUs of the affinity set this context's launch thread is pinned into: max(device, contexts created before it in the process) */
    int opt_launch_groups = 0;       /* pngloss_hip_set_option("launch_groups", "2" | "3" | "auto"): see run_seg_engine */
    /* one device arena, regrown on demand, carved per batch */
    char *d_ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<PlJob> h_jobs;
    size_t n_last = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr }; /* total start, engine start, engine stop, total stop */
    double engine_ms = -1.0, total_ms = -1.0, deflate_ms = -1.0;
    bool pending = false;
    /* host-pointer batches (pngloss_hip_optimize_batch_host*): a persistent device arena and a persistent pinned staging
     * buffer of the same layout, both regrown on demand -- no hipMalloc/hipFree and no pageable copies per call */
    char *d_arena = nullptr;
    size_t arena_bytes = 0;
    char *h_pinned = nullptr;
    size_t pinned_bytes = 0;
    std::string opt_engine;          /* pngloss_hip_set_option("engine", ...): empty = the cost model (or, for tests, $PNGLOSS_HIP_ENGINE) */
    char *d_frames = nullptr;        /* device frames of pngloss_hip_png_decode_batch_device: decoded RGBA8 that stays on the device for the optimiser */
    size_t frames_bytes = 0;
    hipStream_t copy_stream = nullptr;
    double upload_ms = -1.0, download_ms = -1.0;
    /* -v progress display of the single-image seam: a host-mapped word the engine writes the finished row count to */
    uint32_t *h_progress = nullptr;
    bool want_progress = false;
    /* segment-parallel engine: two host-mapped words (images finished, attempt being started) the control kernel writes and the
     * launch loop reads, and what the last batch did */
    bool split_last = false;         /* the last host window ran in chunks: images behind n_last are the peers' */
    std::vector<pngloss_hip_ctx *> peers;   /* further contexts on the same device: the other chunks of a host window (batch_host) */
    std::vector<size_t> chunk_first; /* first image of every chunk of the last split window (chunk 0 is this context's) */
    uint32_t *h_seg_words = nullptr;
    /* the segment engine's launch loop runs on a helper thread and on a stream of its own (the number of row attempts is decided by the
     * data): the caller's stream waits for the "images finished" word instead of for the host */
    hipStream_t seg_stream = nullptr;
    hipStream_t seg_gstream[SEG_MAX_GROUPS] = {};   /* [0] = seg_stream: one stream per GROUP of a batch's images (run_seg_engine) */
    hipEvent_t ev_seg_gdone[SEG_MAX_GROUPS] = {};
    int seg_groups = 1;
    bool seg_async_wait = false;     /* the last batch on the segment engine put a wait for the engine on the caller's stream (the asynchronous entry's non-blocking variant) */
    bool three_groups_ok = false;    /* ... and it is the device-resident synchronous entry point itself: a batch may run as three launch groups (the host-pointer entry points, which the
                                        multi-device wrapper calls from a thread per context, stay at two: several contexts of one process share its hardware queues) */
    bool sync_call = false;          /* the batch under way was started by a SYNCHRONOUS entry point (pngloss_hip_optimize_batch): the caller waits on the host anyway, so the
                                        caller's stream gets no device-side wait for the engine's finished word (run_seg_engine) */
    hipEvent_t ev_prep = nullptr;    /* caller's stream: everything the engine reads is in place */
    hipEvent_t ev_seg_done = nullptr;/* engine's stream: behind the last attempt the launch thread enqueued */
    std::thread seg_worker;
    std::atomic<int> seg_rc{ 0 };
    std::vector<SegJob> h_sj;        /* (stay alive until the asynchronous copies that read them are done: the next enqueue) */
    std::vector<uint32_t> h_sel;
    SegParams h_seg_params;
    int stream_wait_ok = -1;         /* hipStreamWaitValue32 usable on this device (-1: not asked yet) */
    int seg_prio = 0;
    bool seg_prio_distinct = false;

    int last_engine = 0;            /* 0 = one workgroup per image (pl_engine), 3 = segment-parallel (pl_seg) */
    long seg_attempts = 0;
};

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
    size_t jobs, flags, orig_hist, orig_rank, cand, err0, err1, old_above, final_hist, result, row_ids, out_flags, rowstat, total;
};

/* per-image workspace: everything the engine keeps outside the image itself */
WsLayout image_ws(uint32_t width, uint32_t height, bool rows_engine = false)
{
    WsLayout l{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    l.flags = take(sizeof(uint32_t));
    l.orig_hist = take(sizeof(uint32_t) * PL_NFILT * PL_NSYM);
    l.orig_rank = take(sizeof(uint32_t) * PL_NFILT * PL_NSYM);
    l.cand = take(sizeof(uint4) * PL_NFILT * (size_t)width);
    l.err0 = take(sizeof(uint2) * (size_t)width);
    l.err1 = take(sizeof(uint2) * (size_t)width);
    l.old_above = take(sizeof(uint32_t) * (size_t)width);
    l.final_hist = take(sizeof(uint32_t) * PL_NSYM);
    l.result = take(sizeof(int32_t) * 64);
    l.row_ids = take(height ? height : 1);
    l.out_flags = take(sizeof(uint32_t));
    l.rowstat = rows_engine ? take(sizeof(uint32_t) * PL_ROWSTAT_WORDS * (size_t)(height ? height : 1)) : 0;     /* (strength 0: pl_rows.hip) */
    l.total = o;
    return l;
}

float recip_up_host(long d)
{
    /* one ulp above the correctly rounded reciprocal: > 1/d, and far inside the exactness margin (pl_device.h) */
    float r = 1.0f / (float)d;
    r = std::nextafterf(r, 2.0f);
    return r;
}

int ensure_ws(pngloss_hip_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return PNGLOSS_SUCCESS;
    if (ctx->d_ws) PL_CHECK(hipFree(ctx->d_ws));
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    PL_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_ws), want));
    ctx->ws_bytes = want;
    return PNGLOSS_SUCCESS;
}

struct EmitTarget { void *d_ids; void *d_rows; uint32_t pitch; };

/* The segment-parallel engine on the batch in ctx->h_jobs.  All control flow of the algorithm is on the device; the host only keeps
 * a stream fed with row attempts (five kernels each, pl_seg.hip) until every image has reported that it is finished -- how many that
 * takes depends on the data.  So that the entry point stays ASYNCHRONOUS, the attempts are launched by a helper thread on a stream of
 * the context's own, at most SEG_LOOKAHEAD attempts ahead of the one the device says it is working on; the caller's stream is made
 * to wait for the host-visible "images finished" word (hipStreamWaitValue32), so everything the caller enqueues behind this call
 * still runs behind the engine.  (The images of a mixed batch that the other engine takes run on the caller's stream meanwhile.)
 * (Measured and dropped: the attempts as an executable hipGraph of 16 x (parity 0, parity 1) -- 160 kernel nodes per launch call: the
 * same engine time, and 206 - 246 ms of host CPU per 4096x4096 frame against 79 - 94 ms for the plain launches.) */
/* The engine's streams are kept for the life of the PROCESS (a free list per device): a context takes its streams from the list and gives them back when it is
 * destroyed.  Measured (round 5, tools/gpu_r5_benchlegs.sh): streams are mapped onto the process's few hardware queues when they are created; a context that
 * created fresh streams after an earlier context's three had been destroyed got two streams on ONE queue -- its two launch groups then ran one behind the other
 * (the reference's suite as one batch: 49 -> 40 Mpx/s in every bench.py run, depending on which leg came before).  Streams that are never destroyed keep their queues. */
/* process-wide: has any context put a wait for the engine on a caller's stream / does a third engine stream exist (run_seg_engine: launch groups) */
std::atomic<bool> g_stream_wait_used{ false }, g_third_engine_stream{ false };
struct SegStreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;       /* (device, stream) */
    hipError_t take(int device, int prio, hipStream_t *out)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == device) { *out = idle[i].second; idle.erase(idle.begin() + (long)i); return hipSuccess; }
        }
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
    }
    void give(int device, hipStream_t s)
    {
        std::lock_guard<std::mutex> lk(mu);
        idle.insert(idle.begin(), std::make_pair(device, s));      /* (first in: the next context takes them in the order this one had them) */
    }
};
SegStreamPool &seg_stream_pool() { static SegStreamPool *p = new SegStreamPool; return *p; }      /* (never destroyed: contexts may outlive static destruction order) */

struct SegGroups { PlSegBatch b[SEG_MAX_GROUPS]; int n = 1; };
void seg_worker_main(pngloss_hip_ctx *ctx, SegGroups gs, long max_attempts)
{
    /* words: [2g] images of group g that are finished, [2g + 1] the attempt group g's first image is working on */
    volatile uint32_t *words = ctx->h_seg_words;
    int rc = PNGLOSS_SUCCESS;
    if (hipSetDevice(ctx->device) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
    /* Round 6: the launch thread has a CPU of its own.  It must issue four launches every 45 - 100 us for as long as the engine runs; on a node every rank (or every context of
     * pngloss_hip_multi) has one, next to the threads that stage images.  CPU (2 * slot + 1) of the process's affinity set, slot = the context's device ordinal or its serial
     * number in the process, whichever is larger: eight ranks with one device each and eight contexts of one process get eight different CPUs; skipped when the set is too
     * small for that, or with PNGLOSS_HIP_PIN=0.  (Measured on one box, eight contexts on one device: profiles/r06_host_side.txt: no measurable difference with 256 CPUs visible and a quota of 16 -- 181.56 against 181.62 ms for the headline frame, 1730 - 1751 against 1742 - 1796 ms for configs[3] on eight contexts of one device; all of it on TWO CPUs: 1938 ms.) */
    if (ctx->hooks.pin != 0) {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
            std::vector<int> cpus;
            for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
            const size_t want = (size_t)(2 * ctx->pin_slot + 1);
            if (cpus.size() >= 4 && want < cpus.size()) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[want], &one);
                (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);       /* (a refusal is not an error: the thread stays where the scheduler puts it) */
            }
        }
    }
    /* The engine's streams never wait for another stream ON THE DEVICE: streams share a few hardware queues, a queue is served in order,
     * and the callers' streams hold waits for the finished words -- with twelve contexts, engine j's attempts sat behind engine k's wait
     * for k's inputs, whose kernels sat behind caller j's wait for engine j.  So this thread waits for the inputs, on the host. */
    if (rc == PNGLOSS_SUCCESS && hipEventSynchronize(ctx->ev_prep) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
    const long lookahead = 32;                                  /* attempts queued ahead of the one the device works on */
    long launched[SEG_MAX_GROUPS] = {};
    auto t_last = std::chrono::steady_clock::now();
    uint32_t seen[SEG_MAX_GROUPS] = {};
    int idle = 0;
    long lead_sum = 0, lead_n = 0, lead_min = 1 << 30, lead_low = 0;
    for (;;) {
        if (rc != PNGLOSS_SUCCESS) break;
        bool all_done = true, any_launched = false;
        for (int g = 0; g < gs.n && rc == PNGLOSS_SUCCESS; g++) {
            if (words[2 * g] >= (uint32_t)gs.b[g].n) continue;       /* this group's images are finished */
            all_done = false;
            const uint32_t at = words[2 * g + 1];
            if (at != seen[g]) { seen[g] = at; t_last = std::chrono::steady_clock::now(); idle = 0; }
            if (launched[g] - (long)at > lookahead) continue;       /* the device is busy with what is queued for this group */
            if (launched[g] > max_attempts) {
                std::fprintf(stderr, "pngloss_hip: the segment engine needed more than %ld attempts\n", max_attempts);
                rc = PNGLOSS_HIP_ERROR;
                break;
            }
            /* a group whose rows keep breaking off -- its images hold fixed points and cycles the seeds do not reach (flat content; real photographs break in ~5 % of their rows,
             * the generator's frames in 0.3 %) -- goes back to the start from every state for the rest of the batch: more than one image-row in 25, sixteen to begin with (the
             * device-side rule, seg_unit_from_seeds, does the same image by image inside the kernel; this one changes the KERNEL -- for small batches seg_k_enum instead of
             * seg_k_enum_unit<1> with its slow exhaustive path).  Results do not depend on it; the count lags the launches by the look-ahead. */
            if (gs.b[g].seeds && (uint64_t)words[2 * SEG_MAX_GROUPS + g] * 25u > (uint64_t)at * gs.b[g].n + 400u) gs.b[g].seeds = false;
            if (launched[g] >= 64) { const long lead = launched[g] - (long)at; lead_sum += lead; lead_n++; if (lead < lead_min) lead_min = lead; if (lead <= 2) lead_low++; }   /* (PNGLOSS_HIP_DEBUG: how far ahead of the device the launches run) */
            const hipError_t e = pl_seg_launch_attempt(gs.b[g], (int)launched[g], ctx->seg_gstream[g]);
            if (e != hipSuccess) { std::fprintf(stderr, "pngloss_hip: launching a row attempt failed: %s\n", hipGetErrorString(e)); rc = PNGLOSS_HIP_ERROR; break; }
            launched[g]++;
            any_launched = true;
        }
        if (all_done || rc != PNGLOSS_SUCCESS) break;
        if (!any_launched) {
            if (std::chrono::steady_clock::now() - t_last > std::chrono::seconds(20)) {
                std::fprintf(stderr, "pngloss_hip: the segment engine stopped making progress (attempts of the first groups: %u %u %u %u)\n", seen[0], seen[1], seen[2], seen[3]);
                rc = PNGLOSS_HIP_ERROR;
                break;
            }
            /* every group has its look-ahead queued: leave the core to others (a row attempt takes 50 - 200 us) */
            if (++idle < 4) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    for (int g = 0; g < gs.n; g++)
        if (hipEventRecord(ctx->ev_seg_gdone[g], ctx->seg_gstream[g]) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
    if (rc != PNGLOSS_SUCCESS) {
        /* whatever went wrong, the caller's stream must not wait for ever: release it (the images are NOT finished: the error is
         * reported by pngloss_hip_finish), then let what is queued drain */
        for (int g = 0; g < gs.n; g++) words[2 * g] = (uint32_t)gs.b[g].n;
        for (int g = 0; g < gs.n; g++) (void)hipStreamSynchronize(ctx->seg_gstream[g]);
    }
    if (ctx->hooks.debug && lead_n)
        std::fprintf(stderr, "pngloss_hip: launch thread: %d group(s), %ld launches of an attempt; attempts queued ahead of the device when launching: average %.1f, least %ld, at most two ahead in %ld launches (look-ahead %ld)\n",
                     gs.n, lead_n, (double)lead_sum / (double)lead_n, lead_min, lead_low, lookahead);
    long mx = 0;
    for (int g = 0; g < gs.n; g++) mx = std::max(mx, launched[g]);
    ctx->seg_attempts = mx;
    ctx->seg_rc.store(rc, std::memory_order_release);
}

int run_seg_engine(pngloss_hip_ctx *ctx, const PlJob *d_jobs, const std::vector<uint32_t> &list, const SegParams &params, const std::vector<size_t> &seg_offs,
                   size_t jobs_off, size_t params_off, hipStream_t stream, const uint32_t *d_sel, size_t n_wg, const PlEngineParams &prm)
{
    const size_t n = list.size();                              /* the images of the batch this engine takes */
    if (!ctx->h_seg_words) PL_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_seg_words), 16 * SEG_MAX_GROUPS, hipHostMallocMapped | hipHostMallocCoherent));
    if (!ctx->seg_stream) {
        /* a stream of the HIGHEST priority: streams of one priority share a few hardware queues, and a queue whose head is a caller's
         * wait for the finished word holds up everything behind it -- the engine's attempts must never sit in such a queue (twelve
         * contexts with twelve waiting streams deadlocked that way before this stream had a priority of its own) */
        int least = 0, greatest = 0;
        PL_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        ctx->seg_prio = greatest;
        PL_CHECK(seg_stream_pool().take(ctx->device, greatest, &ctx->seg_stream));
        ctx->seg_prio_distinct = greatest != least;
        ctx->seg_gstream[0] = ctx->seg_stream;
    }
    if (!ctx->ev_prep) PL_CHECK(hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
    if (!ctx->ev_seg_done) { PL_CHECK(hipEventCreateWithFlags(&ctx->ev_seg_done, hipEventDisableTiming)); ctx->ev_seg_gdone[0] = ctx->ev_seg_done; }
    /* GROUPS: a batch's images in up to SEG_MAX_GROUPS launch sequences on as many streams.  An attempt is four dependent launches with a latency floor
     * each (the enumeration's dependent steps above all: 75 us with units); images of ONE sequence sit through every floor together, images of different
     * sequences fill each other's floors.  One launch thread feeds all of them. */
    int ngroups = 1;
    {
        size_t segs = 0;
        for (size_t i = 0; i < n; i++) segs += (ctx->h_jobs[list[i]].width + SEG_L - 1) / SEG_L;
        /* (measured, profiles/r05_unit_groups.txt: two sequences 1.2x one from 16 frames of 1080p on, three another 2-6 %, FOUR collapse -- 250 ms for 8 frames
         *  against 113: with the caller's stream they outnumber the hardware queues a process gets, and the stream that shares a queue with the caller's sits
         *  behind its wait for the finished word (without that wait four run, six collapse: profiles/r05_validation_in_enum.txt).  THREE looked 6 % faster in a
         *  process that does nothing else -- and halved every later engine run of bench.py's process, single images included (suite batch 45 -> 14 Mpx/s, 8192 x 8192
         *  137 -> 71): the hardware queues a third engine stream brings into the process's pool stay there, and from then on an engine stream shares one with a
         *  waiting stream.  Two it is for the asynchronous entry: a caller with streams of its own must still fit.  The SYNCHRONOUS entry point puts no wait on
         *  any stream (run_seg_engine below): there three groups are safe -- bench.py's process, every leg after a three-group batch at full speed -- and worth
         *  6 % at 32 frames, 3 % at 64.) */
        /* Round 6 (the advisor's finding on round 5): three is OPT-IN -- pngloss_hip_set_option(ctx, "launch_groups", "3"), for a process that uses the synchronous
         * entry point only (bench.py's batch legs do and say so in their output) --, because nothing stopped a process from running a three-group batch and an asynchronous
         * one with a stream of its own later: the default is two.  Two guards on top, process-wide: once ANY context of the process has put a wait on a caller's stream no
         * third engine stream is created any more; and once a third engine stream exists the asynchronous entry takes its blocking variant (no wait on any stream) instead
         * of running at half speed behind one. */
        const bool three = ctx->opt_launch_groups == 3 && ctx->sync_call && ctx->three_groups_ok && n >= 12 && !g_stream_wait_used.load(std::memory_order_relaxed);
        if (segs > SEG_UNIT_MIN_SEGS && n >= 8) ngroups = three ? 3 : 2;
        else if (n >= 2) ngroups = 2;                              /* (a small batch: see the shares below) */
        if (ctx->hooks.seg_groups) ngroups = ctx->hooks.seg_groups;   /* (timing / test hook: results do not depend on it) */
        ngroups = (int)std::min<size_t>((size_t)ngroups, n);
    }
    for (int g = 1; g < ngroups; g++) {
        if (!ctx->seg_gstream[g] && g >= 2) g_third_engine_stream.store(true, std::memory_order_relaxed);
        if (!ctx->seg_gstream[g]) PL_CHECK(seg_stream_pool().take(ctx->device, ctx->seg_prio, &ctx->seg_gstream[g]));
        if (!ctx->ev_seg_gdone[g]) PL_CHECK(hipEventCreateWithFlags(&ctx->ev_seg_gdone[g], hipEventDisableTiming));
    }
    ctx->seg_groups = ngroups;
    if (ctx->stream_wait_ok < 0) {
        int can = 0;
        ctx->stream_wait_ok = (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can) ? 1 : 0;
        if (ctx->hooks.no_stream_wait) ctx->stream_wait_ok = 0;     /* test hook: the blocking variant */
    }
    void *d_words = nullptr;
    PL_CHECK(hipHostGetDevicePointer(&d_words, ctx->h_seg_words, 0));
    volatile uint32_t *words = ctx->h_seg_words;
    for (int q = 0; q < 4 * SEG_MAX_GROUPS; q++) words[q] = 0;       /* ([2g], [2g + 1]: see seg_worker_main; [2 * SEG_MAX_GROUPS + g]: rows of group g the chain kernel broke off) */
    /* group g = images [gfirst[g], gfirst[g + 1]) of `list` (tallest first: enqueue): equal shares -- or, in a batch of two groups whose tallest image stands out, that
     * image alone and the others together.  A group takes as many attempts as its image with the most, and every attempt costs what ALL its images' workgroups
     * cost: the reference's suite as one batch (configs[2]) spent 71 ms on the 1199 attempts of its tallest image, a screenshot whose candidate none fails in 40 % of
     * its rows -- at the price of eight images each; the other seven need 625 (profiles/r05_suite_groups.txt). */
    size_t gfirst[SEG_MAX_GROUPS + 1];
    for (int g = 0; g <= ngroups; g++) gfirst[g] = n * (size_t)g / (size_t)ngroups;
    if (ngroups == 2 && n > 2 && !ctx->hooks.seg_groups) {
        const uint32_t h0 = ctx->h_jobs[list[0]].height, h1 = ctx->h_jobs[list[1]].height;
        if ((uint64_t)h0 * 100u > (uint64_t)h1 * 105u) gfirst[1] = 1;
    }
    auto group_of = [&](size_t i) { int g = 0; while (g + 1 < ngroups && i >= gfirst[g + 1]) g++; return g; };
    ctx->h_sj.assign(n, SegJob{});
    ctx->h_seg_params = params;
    size_t seg_total = 0;
    {
        /* Enumeration in UNITS of SEG_UNIT segments (seg_enum_unit_body): less than half the instructions per row, a dependent path SEG_UNIT times as long.
         * It pays when the batch is what keeps the GPU busy, not the latency of one row: from a handful of images on.  Results do not depend on it (the
         * validation is the ground truth either way); PNGLOSS_HIP_SEG_UNIT=0 / 1 pins it for tests and timing. */
        size_t segs = 0;
        for (size_t i = 0; i < n; i++) segs += (ctx->h_jobs[list[i]].width + SEG_L - 1) / SEG_L;
        const bool can = !params.seeded && params.ns <= SEG_NSP;       /* (state sets of one chunk of lanes: with more, the distinct states of a workgroup's pairs outgrow its lanes) */
        const bool have_seeds = can && params.seed_n > 0 && ctx->hooks.seg_seeds != 0;
        /* (with seeds the per-segment enumeration stays ahead up to sixteen 1080p frames: pl_seg_core.h -- where it applies: batches of NARROW images, which it does not take, go to units from
         *  the round-5 size on: 40 photographs of 512 .. 768 pixels, 784 segments, 54.8 ms in units against 62.4 per segment from every state) */
        const bool seeds1_fit = have_seeds && n >= 2 && segs >= SEG_SEEDS1_MIN_SEGS && segs >= (size_t)SEG_SEEDS1_MIN_SEGS_PER_IMAGE * n;
        bool units = can && segs > (seeds1_fit ? (size_t)SEG_UNIT_MIN_SEGS_SEEDS : (size_t)SEG_UNIT_MIN_SEGS);
        seg_total = segs;
        if (ctx->hooks.seg_unit >= 0) units = can && ctx->hooks.seg_unit != 0;
        ctx->h_seg_params.unit = units ? SEG_UNIT : 1;
        ctx->h_seg_params.tparts = units ? SEG_TPARTS_BATCH : SEG_TPARTS;     /* (batches: one control workgroup per candidate) */
        if (ctx->hooks.tparts == 1 || ctx->hooks.tparts == SEG_TPARTS) ctx->h_seg_params.tparts = ctx->hooks.tparts == 1 ? SEG_TPARTS_BATCH : SEG_TPARTS;   /* (timing / test hook) */
        if (ctx->hooks.seed_kin >= 1 && ctx->hooks.seed_kin <= SEG_SEED_KMAX) ctx->h_seg_params.seed_kin = ctx->hooks.seed_kin;   /* (timing hook) */
    }
    SegGroups gs{};                     /* (value-initialised: the per-group maxima below start from zero) */
    gs.n = ngroups;
    uint32_t max_h = 0;
    for (size_t i = 0; i < n; i++) {
        const PlJob &pj = ctx->h_jobs[list[i]];
        const PlSegLayout l = pl_seg_layout(pj.width ? pj.width : 1, (uint32_t)params.nsp, params.seeded != 0);
        char *base = ctx->d_ws + seg_offs[i];
        SegJob &s = ctx->h_sj[i];
        s.job_index = list[i];
        s.img = pj.img; s.row_filters = pj.row_filters; s.row_ids = pj.row_ids; s.W = pj.width; s.H = pj.height; s.bpp = 0;
        s.orig_rank = pj.orig_rank; s.cand = reinterpret_cast<uint32_t *>(pj.cand);
        s.err0 = reinterpret_cast<uint32_t *>(base + l.err0); s.err1 = reinterpret_cast<uint32_t *>(base + l.err1); s.rowcopy = reinterpret_cast<uint32_t *>(base + l.rowcopy);
        s.final_hist = pj.final_hist; s.result = pj.result; s.progress = pj.progress;
        s.break_word = static_cast<uint32_t *>(d_words) + 2 * SEG_MAX_GROUPS + group_of(i);
        { const int g = group_of(i); s.done_counter = static_cast<uint32_t *>(d_words) + 2 * g; s.attempt_word = i == gfirst[g] ? static_cast<uint32_t *>(d_words) + 2 * g + 1 : nullptr; }
        s.ctl = reinterpret_cast<SegCtl *>(base + l.ctl); s.base = reinterpret_cast<uint32_t *>(base + l.base);
        s.H0 = reinterpret_cast<uint32_t *>(base + l.h0); s.acc = reinterpret_cast<SegAcc *>(base + l.acc);
        s.tables = reinterpret_cast<uint32_t *>(base + l.tables); s.maps = reinterpret_cast<uint16_t *>(base + l.maps); s.ehash = reinterpret_cast<uint32_t *>(base + l.ehash);
        s.rout = reinterpret_cast<uint16_t *>(base + l.rout); s.rst = reinterpret_cast<uint32_t *>(base + l.rst); s.rck = reinterpret_cast<uint32_t *>(base + l.rck); s.dnout = reinterpret_cast<uint16_t *>(base + l.dnout); s.dcnt = reinterpret_cast<uint32_t *>(base + l.dcnt);
        s.entry = reinterpret_cast<uint32_t *>(base + l.entry); s.segcnt = reinterpret_cast<uint16_t *>(base + l.segcnt);
        s.grpcnt = reinterpret_cast<uint32_t *>(base + l.grpcnt); s.grpleft = reinterpret_cast<uint32_t *>(base + l.grpleft);
        s.firstidx = reinterpret_cast<uint32_t *>(base + l.firstidx); s.rowmm = reinterpret_cast<int32_t *>(base + l.rowmm);
        s.nseg = pj.width ? l.nseg : 0; s.ngrp = pj.width ? l.ngrp : 0;
        PlSegBatch &bg = gs.b[group_of(i)];
        bg.max_nseg = std::max(bg.max_nseg, l.nseg); bg.max_ngrp = std::max(bg.max_ngrp, l.ngrp);
        bg.max_ncommit = std::max(bg.max_ncommit, (pj.width + SEG_COMMIT_W - 1) / SEG_COMMIT_W);
        max_h = std::max(max_h, pj.height);
    }
    SegJob *d_sj = reinterpret_cast<SegJob *>(ctx->d_ws + jobs_off);
    for (size_t i = 0; i < n; i++) ctx->h_sj[i].self = d_sj + i;
    SegParams *d_params = reinterpret_cast<SegParams *>(ctx->d_ws + params_off);
    PL_CHECK(hipMemcpyAsync(d_sj, ctx->h_sj.data(), sizeof(SegJob) * n, hipMemcpyHostToDevice, stream));
    PL_CHECK(hipMemcpyAsync(d_params, &ctx->h_seg_params, sizeof(SegParams), hipMemcpyHostToDevice, stream));
    for (int g = 0; g < ngroups; g++) {
        PlSegBatch &b = gs.b[g];
        if (!b.max_ncommit) b.max_ncommit = 1;
        b.d_sj = d_sj + gfirst[g]; b.d_params = d_params; b.n = gfirst[g + 1] - gfirst[g];
        b.small_ok = params.small_ok != 0;
        b.seeded = params.seeded != 0;
        b.unit = (uint32_t)ctx->h_seg_params.unit;
        /* round 6: units start from seeds with a run-in where the (strength, bleed) pair has a seed set (PNGLOSS_HIP_SEG_SEEDS=0 / 1 pins it for tests and timing; same bytes) */
        b.seeds = b.unit > 1 && ctx->h_seg_params.seed_n > 0 && ctx->hooks.seg_seeds != 0;
        /* ... and a batch of two or more images below that size goes segment by segment from seeds, through the same bodies (seg_k_enum_unit<1>; PNGLOSS_HIP_SEG_SEEDS1=0 / 1 pins it) */
        if (b.unit == 1 && !params.seeded && ctx->h_seg_params.seed_n > 0 && ctx->hooks.seg_seeds != 0 && ctx->h_seg_params.ns <= SEG_NSP)
            b.seeds = ctx->hooks.seg_seeds1 >= 0 ? ctx->hooks.seg_seeds1 != 0 : (n >= 2 && seg_total >= SEG_SEEDS1_MIN_SEGS && seg_total >= (size_t)SEG_SEEDS1_MIN_SEGS_PER_IMAGE * n);
        b.tparts = (uint32_t)ctx->h_seg_params.tparts;
        b.enum_nt = (size_t)b.max_nseg * b.n <= SEG_ENUM_NT_SMALL_MAX_NSEG ? 512u : 1024u;     /* (the images of THIS group: gridDim.y of its launches) */
        if (ctx->hooks.enum_nt == 512 || ctx->hooks.enum_nt == 1024) b.enum_nt = (uint32_t)ctx->hooks.enum_nt;   /* test hook */
    }
    PL_CHECK(pl_seg_launch_resolve(d_jobs, d_sj, n, stream));
    PL_CHECK(hipEventRecord(ctx->ev_prep, stream));
    if (n_wg) PL_CHECK(pl_launch_engine(d_jobs, d_sel, n_wg, prm, stream));      /* (a mixed batch: the other engine's images, side by side with this one's) */
    /* every row needs one attempt, every epoch one more; a bound far above anything real stops a runaway loop */
    /* every row needs one attempt per strength it is tried at (pngloss_image.c:266-274: down to 0 in the worst case), every epoch two more (the
     * attempt under way when its validation fails is void): a bound far above anything real, there to stop a runaway loop -- the stall
     * detector of the launch thread is the other net.  (Seen: 1813 attempts for a 63 x 2 image at strength 200, all rows adaptive.) */
    const long max_attempts = (long)std::min<double>(2.0e9, (double)max_h * ((double)params.strength + 1.0) * (2.0 + 2.0 * SEG_MAX_RESTARTS * SEG_NFILT) + 1024.0);
    ctx->seg_rc.store(PNGLOSS_SUCCESS, std::memory_order_relaxed);
    /* (round 5, measured with tools/gpu_r5_benchlegs.sh: the stream memory operation on the caller's stream is not free -- its queue polls the finished word while
     *  the engine runs -- : without it the headline frame is 1.4 % faster, the seeded 8192 x 8192 points up to 7 %.  The synchronous entry point has no use for it.) */
    bool waiting = ctx->stream_wait_ok != 0 && ctx->seg_prio_distinct && !ctx->sync_call && !g_third_engine_stream.load(std::memory_order_relaxed);
    if (waiting && stream) {
        /* (a caller's stream of the engine's own priority could share its queue: no wait on that one) */
        int prio = 0;
        if (hipStreamGetPriority(stream, &prio) != hipSuccess || prio == ctx->seg_prio) waiting = false;
    }
    if (waiting) {
        /* the caller's stream goes on behind the engine: when every image has counted itself finished */
        for (int g = 0; g < ngroups && waiting; g++) {
            const hipError_t e = hipStreamWaitValue32(stream, static_cast<uint32_t *>(d_words) + 2 * g, (uint32_t)gs.b[g].n, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { (void)hipGetLastError(); ctx->stream_wait_ok = 0; waiting = false; }     /* (a wait already enqueued is satisfied when its group finishes: harmless) */
            else g_stream_wait_used.store(true, std::memory_order_relaxed);
        }
    }
    ctx->seg_async_wait = waiting;
    try { ctx->seg_worker = std::thread(seg_worker_main, ctx, gs, max_attempts); }
    catch (...) { std::fprintf(stderr, "pngloss_hip: cannot start the launch thread\n"); for (int g = 0; g < ngroups; g++) words[2 * g] = (uint32_t)gs.b[g].n; return PNGLOSS_HIP_ERROR; }
    if (!waiting) {
        /* no stream memory operations on this device: wait for the launch loop here, and order the caller's stream behind the engine's */
        ctx->seg_worker.join();
        for (int g = 0; g < ngroups; g++) PL_CHECK(hipStreamWaitEvent(stream, ctx->ev_seg_gdone[g], 0));
        if (ctx->seg_rc.load(std::memory_order_acquire)) return ctx->seg_rc.load();
    }
    return PNGLOSS_SUCCESS;
}

/* ---- the cost model that picks a batch's row engine: an OPTIONAL calibration per device (PNGLOSS_HIP_CALIB=1) ------------------------------
 * Its constants were fitted on one kind of box (MI355X, 256 CUs, performance level "auto").  Round 6 (the review's item): with the switch, the first batch of two or more images
 * whose engine is the library's to choose runs a small synthetic image (1024 x 32, photographic) three times on each engine through a context of its own -- ~30 ms, once per
 * device and process -- and compares with what the reference box takes for it; the model's per-attempt floor and per-pixel cost are scaled by the ratios (dead band 15 %).
 * OFF by default (enqueue says why: a probe that short reads the clock governor).  What is always taken from the device is its CU count: the model's per-workgroup slope and its
 * "one image per CU" terms.  pngloss_hip_last_engine_info does not change: it reports what ran. */
struct EngineCalib { double seg = 1.0, wg = 1.0, cus = 256.0; double seg_ms = 0, wg_ms = 0; bool done = false; };
constexpr double CALIB_REF_SEG_MS = 2.40, CALIB_REF_WG_MS = 5.97;      /* (the reference box, profiles/r06_host_side.txt: min of three runs of the 1024 x 32 image, five processes: 2.39 .. 2.42 and 5.94 .. 6.04 ms) */
std::mutex g_calib_mu;
EngineCalib g_calib[32];
thread_local bool t_calibrating = false;
EngineCalib engine_calib(int device, bool debug)
{
    std::lock_guard<std::mutex> lk(g_calib_mu);
    EngineCalib &c = g_calib[device & 31];
    if (c.done || t_calibrating) return c;
    c.done = true;                                       /* (whatever happens below: once) */
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c.cus = (double)cus;
    t_calibrating = true;
    pngloss_hip_ctx *tmp = pngloss_hip_create(device);
    const uint32_t w = 1024, h = 32;
    std::vector<uint32_t> img((size_t)w * h);
    uint32_t lcg = 12345u;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            lcg = lcg * 1664525u + 1013904223u;
            const uint32_t r = std::min(255u, x * 255u / w + ((lcg >> 8) & 7u)), g = std::min(255u, y * 255u / h + ((lcg >> 12) & 7u)), b = std::min(255u, (x + y) * 255u / (w + h) + ((lcg >> 16) & 7u));
            img[(size_t)y * w + x] = r | (g << 8) | (b << 16) | ((255u - ((x ^ y) & 31u)) << 24);
        }
    void *d_img = nullptr, *d_f = nullptr;
    bool ok = tmp && hipMalloc(&d_img, img.size() * 4) == hipSuccess && hipMalloc(&d_f, h) == hipSuccess;
    double ms[2] = { 0, 0 };
    for (int e = 0; e < 2 && ok; e++) {
        ok = pngloss_hip_set_option(tmp, "engine", e == 0 ? "seg" : "wg") == PNGLOSS_SUCCESS;
        double best = 1e30;
        for (int rep = 0; rep < 3 && ok; rep++) {
            ok = hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            pngloss_hip_image_desc desc{ d_img, d_f, w, h };
            pngloss_hip_result res{};
            if (ok) ok = pngloss_hip_optimize_batch(tmp, &desc, 1, 19, 2, nullptr, &res) == PNGLOSS_SUCCESS && res.status == 0;
            if (ok) best = std::min(best, pngloss_hip_last_engine_ms(tmp));
        }
        ms[e] = best;
    }
    if (d_img) (void)hipFree(d_img);
    if (d_f) (void)hipFree(d_f);
    if (tmp) pngloss_hip_destroy(tmp);
    t_calibrating = false;
    if (ok && ms[0] > 0 && ms[1] > 0) {
        c.seg_ms = ms[0]; c.wg_ms = ms[1];
        const double rs = ms[0] / CALIB_REF_SEG_MS, rw = ms[1] / CALIB_REF_WG_MS, ratio = rs / rw;
        if (ratio < 0.85 || ratio > 1.0 / 0.85) { c.seg = std::min(2.0, std::max(0.5, rs)); c.wg = std::min(2.0, std::max(0.5, rw)); }
    }
    if (debug) std::fprintf(stderr, "pngloss_hip: engine calibration on device %d: %g CUs, 1024x32 frame: segment engine %.3f ms (reference %.2f), workgroup engine %.3f ms (reference %.2f) -> cost model scales %.2f / %.2f\n",
                            device, c.cus, ms[0], CALIB_REF_SEG_MS, ms[1], CALIB_REF_WG_MS, c.seg, c.wg);
    return c;
}

int enqueue(pngloss_hip_ctx *ctx, const pngloss_hip_image_desc *images, size_t n, const uint32_t *forced_bpp,
            unsigned strength, long bleed, hipStream_t stream, const EmitTarget *emits = nullptr)
{
    if (!ctx) return PNGLOSS_INVALID_ARGUMENT;
    if (strength > 255 || bleed < 1 || bleed > 32767) {
        std::fprintf(stderr, "pngloss_hip: strength must be 0..255 and bleed 1..32767 (got %u, %ld)\n", strength, bleed);
        return PNGLOSS_INVALID_ARGUMENT;
    }
    if (ctx->pending) {
        std::fprintf(stderr, "pngloss_hip: previous batch not finished; call pngloss_hip_finish first\n");
        return PNGLOSS_INVALID_ARGUMENT;
    }
    PL_CHECK(hipSetDevice(ctx->device));
    if (ctx->seg_worker.joinable()) ctx->seg_worker.join();          /* (a batch that failed half-way) */
    ctx->h_jobs.clear();
    ctx->n_last = 0;
    ctx->split_last = false;                                         /* (only a split host window sets it again: batch_host) */
    /* drop empty images (the reference's loops simply do nothing for them) */
    /* STRENGTH 0 has a row engine of its own (pl_rows.hip: nothing is quantised, the five candidate rows are the original row, what is left is the filter search):
     * every image of the batch, unless a test pins another engine.  "rows" pins it (a no-op at other strengths). */
    bool use_rows = false;
    /* the pin of the row engine: the option of the ABI first; the environment variable is the tests' hook (the one hook read per call, once: here) */
    const std::string em_s = !ctx->opt_engine.empty() ? ctx->opt_engine : std::string(std::getenv("PNGLOSS_HIP_ENGINE") ? std::getenv("PNGLOSS_HIP_ENGINE") : "");
    const char *const em = em_s.empty() ? nullptr : em_s.c_str();
    const PlHooks &hk = ctx->hooks;
#ifdef PL_DEBUG_FORCE_FILTER
    const bool forced_filter = true;      /* (a debugging BUILD: candidate PL_DEBUG_FORCE_FILTER wins every row -- not the reference's bytes; pngloss_hip_version says so) */
#else
    const bool forced_filter = false;
#endif
    {
        const bool free_choice = !em || std::strcmp(em, "auto") == 0 || std::strcmp(em, "rows") == 0;
        use_rows = strength == 0 && free_choice && !forced_filter && !hk.force_careful;
    }
    if (use_rows) {
        /* the row-statistics engine keeps PL_ROWSTAT_WORDS counters per ROW of every image (5.6 MB per 1080p frame: 2.8 GB for 512 frames) -- the other engines need nothing
         * comparable.  A batch whose counters would not fit beside its images runs strength 0 the long way (the segment / workgroup engines) instead of failing: same bytes. */
        size_t rs = 0, free_b = 0, total_b = 0;
        for (size_t i = 0; i < n; i++) rs += align_up(sizeof(uint32_t) * PL_ROWSTAT_WORDS * (size_t)(images[i].height ? images[i].height : 1), 256);
        const size_t have = ctx->ws_bytes;                       /* (what the context's arena already holds counts as available) */
        if (rs > have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && rs - have > free_b / 2) {
            if (hk.debug) std::fprintf(stderr, "pngloss_hip: strength 0: %zu MB of row counters against %zu MB free: using the other row engines for this batch\n", rs >> 20, free_b >> 20);
            use_rows = false;
        }
    }
    std::vector<size_t> offs;
    size_t total = align_up(sizeof(PlJob) * (n ? n : 1), 256);
    for (size_t i = 0; i < n; i++) {
        if (!images[i].d_rgba && images[i].width && images[i].height) return PNGLOSS_INVALID_ARGUMENT;
        offs.push_back(total);
        total += image_ws(images[i].width ? images[i].width : 1, images[i].height, use_rows).total;
    }
    /* Which row engine, IMAGE BY IMAGE: one workgroup for the image (pl_engine: batches, narrow images) or the image spread over the
     * whole GPU (pl_seg: few wide images).  The segment engine takes every strength / bleed pair and rows up to SEG_MAX_WIDTH pixels;
     * it pays off while the batch leaves it the machine (its work per row is ~250x redundant by design).  In a mixed batch the two
     * engines run side by side -- the segment engine's images in one launch sequence (blockIdx.y = image) on the engine's own stream,
     * the others as one workgroup each on the caller's.
     * Cost model (measured on 1 .. 64 frames of 512x512 and 1920x1080, tests/tools/gpu_seg_batch.py): a row
     * attempt of the segment engine takes ~38 us plus ~0.032 us per workgroup of its widest kernel (about 3 per segment and 40 more per
     * image), whatever the width, and there are as many attempts as the tallest of its images has rows; the workgroup engine ~0.18 us
     * per pixel of its largest image, all images side by side (256 CUs).  State sets beyond the lanes (s = 85 at bleed 1 or 2 ...) are
     * enumerated from seeds with a run-in of one segment: about twice the enumeration and a wider chain.  Greedy: the images go to the
     * segment engine in the order of their cost on the other one, as long as that shortens the batch. */
    SegParams seg_params;
    std::vector<uint8_t> on_seg(n, 0);
    size_t n_seg = 0;
    {
        const bool forced = em && std::strcmp(em, "seg") == 0;
        const bool allowed = !em || forced || std::strcmp(em, "auto") == 0 || std::strcmp(em, "rows") == 0;       /* "wg" / "lead" / "legacy": the one-workgroup-per-image engine */
        bool seg_ok = n && allowed && !use_rows && !hk.force_careful && pl_seg_supported(nullptr, 0, strength, bleed, &seg_params);
        if (seg_ok) {
            EngineCalib cal;
            /* OPT-IN (PNGLOSS_HIP_CALIB=1) since it was measured inside bench.py: the probe is 2 - 6 ms of GPU work, and what it finds depends on what the process did before it
             * (the clock governor follows the load: after the headline's steps the probe read the segment engine 1.3x slower against the other one than in a fresh process, the
             * model sent 128 frames of 1080p to the wrong engine and the 256-frame batch lost 16 %: profiles/r06_host_side.txt).  A wrong calibration costs more than the constants
             * of the reference box cost on a box of another kind; the CU count (deterministic) is always taken from the device. */
            if (!em && n >= 2 && hk.calib > 0 && !t_calibrating) cal = engine_calib(ctx->device, hk.debug);
            else { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) cal.cus = (double)cus; }
            const double cu_scale = 256.0 / cal.cus;
            const double a_us = seg_params.seeded ? 82.0 : 38.0, w_us = seg_params.seeded ? 0.05 : 0.032;   /* (round 4: an attempt is four launches: 49.5 us at 4096 pixels = 424 workgroups in these units, 46 at 1920, 69 at 8192) */
            /* round 5: a batch whose images have more than SEG_UNIT_MIN_SEGS segments between them is enumerated in UNITS, in two launch groups, with the
             * small workgroups of batches (run_seg_engine): an attempt then takes ~45 us + 0.015 us per workgroup-unit, but not less than ~95 us (the
             * dependent steps of a unit): 1080p frames 16 / 32 / 64 = 102 / 150 / 261 us measured (profiles/r05_unit_groups.txt) */
            const bool can_units = !seg_params.seeded && seg_params.ns <= SEG_NSP;
            auto attempt_us_ref = [&](double wgs, double segs, size_t k) {
                const bool have_seeds = can_units && seg_params.seed_n > 0 && hk.seg_seeds != 0;
                /* round 6, from seeds (per row of the tallest image, epochs included; 1080p frames, profiles/r06_seeds.txt): units 24 / 32 / 64 / 128 frames 105 / 114 / 168 / 301 us,
                 * segment by segment 6 / 11 / 16 frames 61 / 76 / 90 us -- 128 frames 325 ms against 373 on the other engine, the crossover near 148 */
                const bool seeds1_fit = have_seeds && k >= 2 && segs >= SEG_SEEDS1_MIN_SEGS && segs >= (double)SEG_SEEDS1_MIN_SEGS_PER_IMAGE * (double)k;
                if (have_seeds && segs > (seeds1_fit ? SEG_UNIT_MIN_SEGS_SEEDS : SEG_UNIT_MIN_SEGS)) return std::max(100.0, 35.0 + 0.00945 * wgs);
                if (seeds1_fit) return 43.0 + 0.0134 * wgs;
                if (can_units && segs > SEG_UNIT_MIN_SEGS) return std::max(100.0, 28.0 + 0.0124 * wgs);   /* (three launch groups, validation in whole replay groups: 16 / 64 / 96 / 112 / 128 frames of 1080p 102 / 205 / 289 / 333 / 377 us: the segment engine up to 116 such frames -- measured: 112 frames 361 against 372 ms, 120 frames 385 against 372) */
                /* (two or more images run as two launch sequences side by side: 4 / 8 / 12 frames of 1080p 58 / 80 / 102 us per attempt, profiles/r05_suite_groups.txt) */
                if (k >= 2 && !seg_params.seeded) return 35.0 + 0.026 * wgs;
                return a_us + w_us * wgs;
            };
            auto attempt_us = [&](double wgs, double segs, size_t k) { return cal.seg * attempt_us_ref(wgs * cu_scale, segs, k); };      /* (fewer CUs: every workgroup weighs more) */
            auto wg_cost = [&](size_t i) { return cal.wg * 0.18 * (double)images[i].width * (double)images[i].height; };
            std::vector<size_t> order;
            for (size_t i = 0; i < n; i++)
                if (images[i].width && images[i].height && images[i].width <= SEG_MAX_WIDTH) order.push_back(i);
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return wg_cost(x) > wg_cost(y); });
            if (forced) { for (size_t i : order) on_seg[i] = 1; n_seg = order.size(); }
            else {
                /* wg side: the largest image left sets its time (or the sum over 256 CUs when there are more images than CUs) */
                double wg_sum = 0;
                for (size_t i = 0; i < n; i++) wg_sum += wg_cost(i);
                double seg_rows = 0, seg_wgs = 0, seg_segs = 0;
                auto batch_us = [&](size_t k, double rows, double wgs, double wsum, double segs) {   /* the first k images of `order` on the segment engine */
                    const double wg_us = k < order.size() ? std::max(wg_cost(order[k]), wsum / cal.cus) : wsum / cal.cus;
                    const double seg_us = k ? rows * attempt_us(wgs, segs, k) : 0.0;
                    /* side by side only while the other engine leaves the segment engine CUs to run on: its workgroups are persistent and own a CU each (104 KB of
                     * LDS, every register) -- next to 200 and more of them the segment engine's launches wait until they are through: one after the other
                     * (measured: 512 frames of 1080p in one call, 130 of them sent to the segment engine by the model of before: 1034 ms against 2 x 375) */
                    if ((double)(n - k) > 0.75 * cal.cus) return wg_us + seg_us;
                    return std::max(wg_us, seg_us);
                };
                double best = batch_us(0, 0, 0, wg_sum, 0);
                size_t best_k = 0;
                double wsum = wg_sum;
                for (size_t k = 1; k <= order.size(); k++) {
                    const size_t i = order[k - 1];
                    seg_rows = std::max(seg_rows, (double)images[i].height);
                    seg_wgs += 3.0 * ((images[i].width + SEG_L - 1) / SEG_L) + 40.0;
                    seg_segs += (images[i].width + SEG_L - 1) / SEG_L;
                    wsum -= wg_cost(i);
                    if (seg_segs > 8192) break;
                    const double t = batch_us(k, seg_rows, seg_wgs, wsum, seg_segs);
                    if (t < best) { best = t; best_k = k; }
                }
                for (size_t k = 0; k < best_k; k++) on_seg[order[k]] = 1;
                n_seg = best_k;
            }
        }
        if (forced && !n_seg && n)
            std::fprintf(stderr, "pngloss_hip: PNGLOSS_HIP_ENGINE=seg: nothing in this batch for the segment engine (widths beyond %u?); using the one-workgroup-per-image engine\n", SEG_MAX_WIDTH);
    }
    const bool use_seg = n_seg != 0;
    std::vector<uint32_t> seg_list, wg_list;
    for (size_t i = 0; i < n; i++) (on_seg[i] ? seg_list : wg_list).push_back((uint32_t)i);
    /* the segment engine's images, tallest first: its launch groups are runs of this list (run_seg_engine), and the tallest image gets a sequence of its own when it stands out */
    std::stable_sort(seg_list.begin(), seg_list.end(), [&](uint32_t x, uint32_t y) { return images[x].height > images[y].height; });
    std::vector<size_t> seg_offs;
    size_t seg_jobs_off = 0, seg_params_off = 0, sel_off = 0;
    if (use_seg) {
        seg_jobs_off = total; total += align_up(sizeof(SegJob) * n_seg, 256);
        seg_params_off = total; total += align_up(sizeof(SegParams), 256);
        sel_off = total; total += align_up(sizeof(uint32_t) * (wg_list.size() ? wg_list.size() : 1), 256);
        for (uint32_t i : seg_list) { seg_offs.push_back(total); total += pl_seg_layout(images[i].width ? images[i].width : 1, (uint32_t)seg_params.nsp, seg_params.seeded != 0).total; }
    }
    int rc = ensure_ws(ctx, total);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) {
        const WsLayout l = image_ws(images[i].width ? images[i].width : 1, images[i].height, use_rows);
        char *b = ctx->d_ws + offs[i];
        PlJob j{};
        j.rowstat = use_rows ? reinterpret_cast<uint32_t *>(b + l.rowstat) : nullptr;
        j.img = static_cast<uint32_t *>(images[i].d_rgba);
        j.row_filters = static_cast<uint8_t *>(images[i].d_row_filters);
        j.width = images[i].width;
        j.height = (images[i].width == 0) ? 0 : images[i].height;
        j.forced_bpp = forced_bpp ? forced_bpp[i] : 0;
        j.flags = reinterpret_cast<uint32_t *>(b + l.flags);
        j.orig_hist = reinterpret_cast<uint32_t *>(b + l.orig_hist);
        j.orig_rank = reinterpret_cast<uint32_t *>(b + l.orig_rank);
        j.cand = reinterpret_cast<uint4 *>(b + l.cand);
        j.err0 = reinterpret_cast<uint2 *>(b + l.err0);
        j.err1 = reinterpret_cast<uint2 *>(b + l.err1);
        j.old_above = reinterpret_cast<uint32_t *>(b + l.old_above);
        j.final_hist = reinterpret_cast<uint32_t *>(b + l.final_hist);
        j.result = reinterpret_cast<int32_t *>(b + l.result);
        j.row_ids = reinterpret_cast<uint8_t *>(b + l.row_ids);
        j.out_flags = reinterpret_cast<uint32_t *>(b + l.out_flags);
        if (emits && emits[i].d_rows) {
            j.emit_ids = static_cast<uint8_t *>(emits[i].d_ids);
            j.emit_rows = static_cast<uint8_t *>(emits[i].d_rows);
            j.emit_pitch = emits[i].pitch;
            j.emit_adaptive_all = images[i].d_row_filters ? 0u : 1u;
        }
        j.progress = nullptr;
        if (ctx->want_progress && i == 0 && ctx->h_progress) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, ctx->h_progress, 0) == hipSuccess) j.progress = static_cast<uint32_t *>(dp);
        }
        ctx->h_jobs.push_back(j);
    }
    if (!n) return PNGLOSS_SUCCESS;
    PlJob *d_jobs = reinterpret_cast<PlJob *>(ctx->d_ws);
    PL_CHECK(hipMemcpyAsync(d_jobs, ctx->h_jobs.data(), sizeof(PlJob) * n, hipMemcpyHostToDevice, stream));

    PlEngineParams prm{};
    prm.strength = (int)strength;
    prm.rq = recip_up_host((long)strength + 1);
    prm.rbleed = recip_up_host(bleed);
    prm.r29 = 2.0f * recip_up_host(9);
    prm.force_careful = hk.force_careful;   /* test hook, see pl_device.h */
    {
        /* "legacy" = round-1 chains only */
        prm.engine_mode = (em && std::strcmp(em, "legacy") == 0) ? 1 : ((em && std::strcmp(em, "lead") == 0) ? 2 : ((em && std::strcmp(em, "mix") == 0) ? 3 : 0));   /* "lead": never fall back adaptively; "mix": alternate every four rows */
#ifdef PL_DEBUG_FORCE_FILTER
        prm.engine_mode |= ((PL_DEBUG_FORCE_FILTER) + 1) << 8;   /* debugging build only */
#endif
    }

    PL_CHECK(hipEventRecord(ctx->ev[0], stream));
    PL_CHECK(pl_launch_prepare(d_jobs, ctx->h_jobs.data(), n, stream, !use_rows));
    ctx->last_engine = use_seg ? 3 : (use_rows ? 4 : 0);
    PL_CHECK(hipEventRecord(ctx->ev[1], stream));
    /* the images of the one-workgroup-per-image engine: all of them, or -- a mixed batch -- those the segment engine did not get; they
     * run on the caller's stream while the segment engine works on its own */
    const uint32_t *d_sel = nullptr;
    if (use_seg && !wg_list.empty()) {
        ctx->h_sel = wg_list;
        PL_CHECK(hipMemcpyAsync(ctx->d_ws + sel_off, ctx->h_sel.data(), sizeof(uint32_t) * wg_list.size(), hipMemcpyHostToDevice, stream));
        d_sel = reinterpret_cast<const uint32_t *>(ctx->d_ws + sel_off);
    }
    if (use_seg) {
#ifdef PL_DEBUG_FORCE_FILTER
        seg_params.engine_flags = ((PL_DEBUG_FORCE_FILTER) + 1) << 8;   /* debugging build only */
#endif
        if (hk.segprof) seg_params.engine_flags |= 1;                                                   /* phase clocks of the validation kernel */
        if (seg_params.seeded && hk.kin >= 0 && hk.kin <= SEG_KIN) seg_params.kin = hk.kin;             /* experiment: run-in pixels of the seeded enumeration */
        rc = run_seg_engine(ctx, d_jobs, seg_list, seg_params, seg_offs, seg_jobs_off, seg_params_off, stream, d_sel, wg_list.size(), prm);
        if (rc) return rc;
    } else if (use_rows) PL_CHECK(pl_launch_rows(d_jobs, ctx->h_jobs.data(), n, stream));
    else PL_CHECK(pl_launch_engine(d_jobs, nullptr, n, prm, stream));
    PL_CHECK(hipEventRecord(ctx->ev[2], stream));
    {
        /* (behind this point the segment engine's launch thread may be running: it is joined before an error is returned) */
        hipError_t e = pl_launch_finish(d_jobs, ctx->h_jobs.data(), n, stream);
        if (e == hipSuccess) e = pl_launch_emit(d_jobs, ctx->h_jobs.data(), n, stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[3], stream);
        if (e != hipSuccess) {
            std::fprintf(stderr, "pngloss_hip: enqueueing the kernels behind the row engine failed: %s\n", hipGetErrorString(e));
            if (ctx->seg_worker.joinable()) ctx->seg_worker.join();
            for (int g = 0; g < SEG_MAX_GROUPS; g++) if (ctx->seg_gstream[g]) (void)hipStreamSynchronize(ctx->seg_gstream[g]);
            return PNGLOSS_HIP_ERROR;
        }
    }
    ctx->n_last = n;
    ctx->last_stream = stream;
    ctx->pending = true;
    return PNGLOSS_SUCCESS;
}

int finish(pngloss_hip_ctx *ctx, pngloss_hip_result *results, size_t n)
{
    if (!ctx) return PNGLOSS_INVALID_ARGUMENT;
    if (!ctx->pending) {
        if (results)
            for (size_t i = 0; i < n; i++) results[i] = pngloss_hip_result{ 0, 0, 0, 0, 0 };
        return PNGLOSS_SUCCESS;
    }
    PL_CHECK(hipSetDevice(ctx->device));
    int seg_rc = PNGLOSS_SUCCESS;
    if (ctx->seg_worker.joinable()) {
        ctx->seg_worker.join();
        seg_rc = ctx->seg_rc.load(std::memory_order_acquire);
    }
    PL_CHECK(hipEventSynchronize(ctx->ev[3]));
    if (ctx->last_engine == 3) for (int g = 0; g < SEG_MAX_GROUPS; g++) if (ctx->seg_gstream[g]) PL_CHECK(hipStreamSynchronize(ctx->seg_gstream[g]));   /* (attempts queued behind the last row: they find the images finished) */
    ctx->pending = false;
    if (seg_rc) return seg_rc;
    float ms = 0.f;
    PL_CHECK(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    ctx->engine_ms = ms;
    PL_CHECK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]));
    ctx->total_ms = ms;
    if (ctx->hooks.debug) std::fprintf(stderr, "pngloss_hip: row engine occupancy query: %d workgroups per CU\n", pl_engine_occupancy());
    int worst = PNGLOSS_SUCCESS;
    for (size_t i = 0; i < ctx->n_last; i++) {
        int32_t r[64] = { 0 };
        PL_CHECK(hipMemcpy(r, ctx->h_jobs[i].result, sizeof r, hipMemcpyDeviceToHost));
        if (results && i < n) results[i] = pngloss_hip_result{ r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3], (uint32_t)r[4] };
        if (r[20] == 3) {
            if (ctx->hooks.debug)
                std::fprintf(stderr, "pngloss_hip: image %zu: segment-parallel engine: %d attempts for %u rows, %d epochs (validation restarts), %d rows finished serially, candidate none dropped by its cost bound %d times, %d segments walked step by step by the chain kernel, engine %.3f ms\n",
                             i, r[5], ctx->h_jobs[i].height, r[4], r[6], r[7], r[17], ctx->engine_ms);
            if (ctx->hooks.segprof)
                std::fprintf(stderr, "pngloss_hip:   validation kernel, slowest workgroup per phase (us): load %.1f  pass1 %.1f  watched bins + pass3 %.1f  none bound %.1f  sums %.1f; pending decisions %d, largest reach %d\n",
                             r[40] / 100.0, r[41] / 100.0, r[42] / 100.0, r[43] / 100.0, r[44] / 100.0, r[46], r[47]);
            if (ctx->hooks.segprof)
                std::fprintf(stderr, "pngloss_hip:   control kernel, slowest (us): candidate workgroup up to the table build %.1f, table build %.1f, commit workgroup %.1f\n", r[56] / 100.0, r[57] / 100.0, r[58] / 100.0);
            if (ctx->hooks.segprof && r[61] && r[63])
                std::fprintf(stderr, "pngloss_hip:   ... average (us): candidate workgroup up to the table build %.2f, table build %.2f, commit workgroup %.2f\n",
                             (uint32_t)r[59] / 100.0 / (uint32_t)r[61], (uint32_t)r[60] / 100.0 / (uint32_t)r[61], (uint32_t)r[62] / 100.0 / (uint32_t)r[63]);
            if (ctx->hooks.segprof && r[61])
                std::fprintf(stderr, "pngloss_hip:   ... candidate workgroup, average (us): requests + copy %.2f, decision %.2f, new histogram + fields %.2f\n",
                             (uint32_t)r[19] / 100.0 / (uint32_t)r[61], (uint32_t)r[21] / 100.0 / (uint32_t)r[61], (uint32_t)r[22] / 100.0 / (uint32_t)r[61]);
            if (ctx->hooks.segprof && r[61])
                std::fprintf(stderr, "pngloss_hip:   ... commit workgroup, average (us): requests + copy %.2f, decision %.2f, terms %.2f, rows + extremes %.2f\n",
                             (uint32_t)r[23] / 100.0 / (uint32_t)r[63], (uint32_t)r[45] / 100.0 / (uint32_t)r[63], (uint32_t)r[54] / 100.0 / (uint32_t)r[63], (uint32_t)r[55] / 100.0 / (uint32_t)r[63]);
            if (ctx->hooks.segprof && r[61])
                std::fprintf(stderr, "pngloss_hip:   ... table build, average (us): keys %.2f, classes %.2f, entries + write %.2f\n",
                             (uint32_t)r[37] / 100.0 / (uint32_t)r[61], (uint32_t)r[38] / 100.0 / (uint32_t)r[61], (uint32_t)r[39] / 100.0 / (uint32_t)r[61]);
            if (ctx->hooks.segprof && r[32])
                std::fprintf(stderr, "pngloss_hip:   enumeration workgroups (us), slowest / average: load %.1f / %.2f  first steps (%d, or %d for a state set of one chunk) + dedupe %.1f / %.2f  remaining steps %.1f / %.2f  map %.1f / %.2f; distinct states per channel after the dedupe %.1f; first-segment walker %.1f / %.2f\n",
                             r[24] / 100.0, (uint32_t)r[28] / 100.0 / (uint32_t)r[32], SEG_K1, SEG_K1_ONE_CHUNK, r[25] / 100.0, (uint32_t)r[29] / 100.0 / (uint32_t)r[32], r[26] / 100.0, (uint32_t)r[30] / 100.0 / (uint32_t)r[32],
                             r[27] / 100.0, (uint32_t)r[31] / 100.0 / (uint32_t)r[32], (uint32_t)r[33] / 4.0 / (uint32_t)r[32], r[34] / 100.0, r[36] ? (uint32_t)r[35] / 100.0 / (uint32_t)r[36] : 0.0);
            if (ctx->hooks.segprof && r[16])
                std::fprintf(stderr, "pngloss_hip:   chain workgroups (us), slowest / average: gather %.1f / %.2f  compose %.1f / %.2f  walk %.1f / %.2f  tail %.1f / %.2f; %u runs, %u through the serial walk, %u at the wide stride\n",
                             r[8] / 100.0, (uint32_t)r[12] / 100.0 / (uint32_t)r[16], r[9] / 100.0, (uint32_t)r[13] / 100.0 / (uint32_t)r[16], r[10] / 100.0, (uint32_t)r[14] / 100.0 / (uint32_t)r[16],
                             r[11] / 100.0, (uint32_t)r[15] / 100.0 / (uint32_t)r[16], (uint32_t)r[16], (uint32_t)r[17], (uint32_t)r[18]);
            if (ctx->hooks.segprof && r[53])
                std::fprintf(stderr, "pngloss_hip:   ... average per workgroup (us): load %.2f  pass1 %.2f  watched bins + pass3 %.2f  none bound %.2f  sums %.2f  (%u workgroup runs)\n",
                             (uint32_t)r[48] / 100.0 / (uint32_t)r[53], (uint32_t)r[49] / 100.0 / (uint32_t)r[53], (uint32_t)r[50] / 100.0 / (uint32_t)r[53], (uint32_t)r[51] / 100.0 / (uint32_t)r[53],
                             (uint32_t)r[52] / 100.0 / (uint32_t)r[53], (uint32_t)r[53]);
            if (r[0]) { std::fprintf(stderr, "pngloss_hip: image %zu: no acceptable filter row (device status %d)\n", i, r[0]); worst = PNGLOSS_INTERNAL_ABORT; }
            if (results && i < n) results[i].repaired_pixels = (uint32_t)r[4];
            continue;
        }
        if (ctx->hooks.debug)
            std::fprintf(stderr, "pngloss_hip: image %zu: chain kcycles per wave %d %d %d %d, repaired pixels %d %d %d %d, engine %.3f ms\n", i,
                         r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15], ctx->engine_ms);
        if (ctx->hooks.debug)
            std::fprintf(stderr, "pngloss_hip: image %zu: band-leader row attempts %d, wave 4 kcycles %d, exact redos %d, band rescans (wave 0 / 4) %d %d\n", i,
                         r[5], r[24], r[25], r[6], r[26]);
        if (ctx->hooks.debug && r[5])
            for (int w = 0; w < 5; w++)
                std::fprintf(stderr, "pngloss_hip:   wave %d (%s) kcycles: vector %d  fast groups %d  exact redo %d  rescan %d  table build %d\n", w,
                             w == 0 ? "up" : (w == 1 ? "sub" : (w == 2 ? "average" : (w == 3 ? "paeth" : "none"))),
                             r[32 + 5 * w], r[33 + 5 * w], r[34 + 5 * w], r[35 + 5 * w], r[36 + 5 * w]);
        if (ctx->hooks.debug && r[5])
            std::fprintf(stderr, "pngloss_hip:   cycles per pixel of undisturbed whole-chunk runs (up sub average paeth none): %d %d %d %d %d\n", r[57], r[58], r[59], r[60], r[61]);
        if (ctx->hooks.debug)
            std::fprintf(stderr, "pngloss_hip:   wave 0 kcycles in the post pass %d, in the commit pass %d; flush + relation check per chain wave %d %d %d %d %d\n", r[62], r[63], r[27], r[28], r[29], r[30], r[31]);
        if (ctx->hooks.debug)
            std::fprintf(stderr, "pngloss_hip:   SIMD of waves 0..7: %d %d %d %d %d %d %d %d\n", r[7] & 3, (r[7] >> 2) & 3, (r[7] >> 4) & 3, (r[7] >> 6) & 3,
                         (r[7] >> 8) & 3, (r[7] >> 10) & 3, (r[7] >> 12) & 3, (r[7] >> 14) & 3);
        if (ctx->hooks.debug && r[5])
            std::fprintf(stderr, "pngloss_hip:   light pixels per chain wave %d %d %d %d %d; rows on the round-1 chains by the adaptive choice %d (last cycles per pixel: band-leader %d, round-1 %d)\n", r[16], r[17], r[18], r[19], r[20], r[21], r[22], r[23]);
        if (ctx->hooks.segprof && r[16])   /* (engine built with PL_SEGPROF) */
            for (int w = 0; w < 4; w++)
                std::fprintf(stderr, "pngloss_hip:   wave %d segments kcycles: head+gather %d  reductions %d  check+lut %d  tail %d\n", w,
                             r[16 + 4 * w], r[17 + 4 * w], r[18 + 4 * w], r[19 + 4 * w]);
        if (r[0]) {
            std::fprintf(stderr, "pngloss_hip: image %zu: no acceptable filter row (device status %d)\n", i, r[0]);
            worst = PNGLOSS_INTERNAL_ABORT;
        }
    }
    return worst;
}

/* ---- process-wide context for the host-pointer drop-in seam ------------------------------------------------ */
std::mutex g_mu;
pngloss_hip_ctx *g_ctx = nullptr;

pngloss_hip_ctx *global_ctx()
{
    if (!g_ctx) g_ctx = pngloss_hip_create(-1);
    return g_ctx;
}

/* Upload a packed bpp-byte image as "slots" words, run, download.  rows[] may be non-contiguous. */
int run_host_image(unsigned char **rows, uint32_t width, uint32_t height, uint32_t src_bpp, uint32_t forced_bpp,
                   unsigned char *row_filters, bool verbose, unsigned strength, long bleed)
{
    if (!width || !height) return PNGLOSS_SUCCESS;
    std::lock_guard<std::mutex> lock(g_mu);
    pngloss_hip_ctx *ctx = global_ctx();
    if (!ctx) {
        std::fprintf(stderr, "pngloss_hip: no usable HIP device -- refusing to fall back to a CPU path\n");
        return PNGLOSS_HIP_ERROR;
    }
    const size_t npx = (size_t)width * height;
    std::vector<uint32_t> staging;
    try { staging.resize(npx); } catch (const std::bad_alloc &) { return PNGLOSS_OUT_OF_MEMORY_ERROR; }
    for (uint32_t y = 0; y < height; y++) {
        const unsigned char *s = rows[y];
        uint32_t *d = staging.data() + (size_t)y * width;
        if (src_bpp == 4) std::memcpy(d, s, (size_t)width * 4);
        else
            for (uint32_t x = 0; x < width; x++) {
                uint32_t w = 0;
                for (uint32_t c = 0; c < src_bpp; c++) w |= (uint32_t)s[(size_t)x * src_bpp + c] << (8 * c);
                d[x] = w;
            }
    }
    void *d_img = nullptr, *d_filt = nullptr;
    PL_CHECK(hipSetDevice(ctx->device));
    PL_CHECK(hipMalloc(&d_img, npx * 4));
    if (row_filters) {
        hipError_t e = hipMalloc(&d_filt, height);
        if (e != hipSuccess) { (void)hipFree(d_img); return PNGLOSS_OUT_OF_MEMORY_ERROR; }
    }
    int rc = PNGLOSS_SUCCESS;
    pngloss_hip_result res{};
    do {
        if (hipMemcpy(d_img, staging.data(), npx * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = PNGLOSS_HIP_ERROR; break; }
        pngloss_hip_image_desc desc{ d_img, d_filt, width, height };
        if (verbose && !ctx->h_progress &&
            hipHostMalloc(reinterpret_cast<void **>(&ctx->h_progress), sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
            ctx->h_progress = nullptr;                     /* no display then; not an error */
        if (ctx->h_progress) *ctx->h_progress = 0;
        ctx->want_progress = verbose && ctx->h_progress;
        ctx->sync_call = !ctx->want_progress;            /* (the progress display polls while the engine runs: it needs the call back at once; else the host waits anyway) */
        rc = enqueue(ctx, &desc, 1, forced_bpp ? &forced_bpp : nullptr, strength, bleed, nullptr);
        ctx->sync_call = false;
        ctx->want_progress = false;
        if (rc) break;
        if (verbose && ctx->h_progress) {
            /* the progress display of pngloss_image.c:214-237: spinner at 10 Hz and the share of finished rows, on stderr */
            static const char spinner[] = "|/-\\";
            unsigned spin = 0;
            while (hipEventQuery(ctx->ev[3]) == hipErrorNotReady) {
                const uint32_t rows_done = *reinterpret_cast<volatile uint32_t *>(ctx->h_progress);
                std::fprintf(stderr, "\x1B[\x01G%c %.1f%% complete", spinner[spin++ & 3], 100.0 * rows_done / (double)height);
                std::fflush(stderr);
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        rc = finish(ctx, &res, 1);
        if (rc) break;
        if (hipMemcpy(staging.data(), d_img, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = PNGLOSS_HIP_ERROR; break; }
        if (row_filters && hipMemcpy(row_filters, d_filt, height, hipMemcpyDeviceToHost) != hipSuccess) { rc = PNGLOSS_HIP_ERROR; break; }
    } while (0);
    (void)hipFree(d_img);
    if (d_filt) (void)hipFree(d_filt);
    if (rc == PNGLOSS_HIP_ERROR) std::fprintf(stderr, "pngloss_hip: device transfer or kernel failure: %s\n", hipGetErrorString(hipGetLastError()));
    if (rc) return rc;
    for (uint32_t y = 0; y < height; y++) {
        unsigned char *d = rows[y];
        const uint32_t *s = staging.data() + (size_t)y * width;
        if (src_bpp == 4) std::memcpy(d, s, (size_t)width * 4);
        else
            for (uint32_t x = 0; x < width; x++)
                for (uint32_t c = 0; c < src_bpp; c++) d[(size_t)x * src_bpp + c] = (unsigned char)(s[x] >> (8 * c));
    }
    if (verbose) {
        /* pngloss_image.c:309-325 */
        std::fputs("\x1B[\x01G  compression complete\n", stderr);
        std::fprintf(stderr, "  used %u unique symbols\n", res.unique_symbols);
    }
    return PNGLOSS_SUCCESS;
}

} // namespace

/* ================================================================================================ C ABI */

extern "C" {

int pngloss_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -PNGLOSS_HIP_ERROR;
    return n;
}

pngloss_hip_ctx *pngloss_hip_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        std::fprintf(stderr, "pngloss_hip: no HIP device available\n");
        return nullptr;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= n) {
        std::fprintf(stderr, "pngloss_hip: device %d out of range (%d visible)\n", device, n);
        return nullptr;
    }
    pngloss_hip_ctx *ctx = new (std::nothrow) pngloss_hip_ctx;
    if (!ctx) return nullptr;
    ctx->device = device;
    ctx->hooks = PlHooks::from_env();
    { static std::atomic<int> serial{ 0 }; ctx->pin_slot = std::max(device, serial.fetch_add(1)); }
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return nullptr; }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { pngloss_hip_destroy(ctx); return nullptr; }
    return ctx;
}

void pngloss_hip_destroy(pngloss_hip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->seg_worker.joinable()) ctx->seg_worker.join();
    if (ctx->pending) (void)hipEventSynchronize(ctx->ev[3]);
    for (int g = SEG_MAX_GROUPS - 1; g >= 0; g--) if (ctx->seg_gstream[g]) { (void)hipStreamSynchronize(ctx->seg_gstream[g]); seg_stream_pool().give(ctx->device, ctx->seg_gstream[g]); }   /* (back to the process's list, [0] first) */
    for (int g = 1; g < SEG_MAX_GROUPS; g++) if (ctx->ev_seg_gdone[g]) (void)hipEventDestroy(ctx->ev_seg_gdone[g]);
    if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
    if (ctx->ev_seg_done) (void)hipEventDestroy(ctx->ev_seg_done);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_arena) (void)hipFree(ctx->d_arena);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->h_progress) (void)hipHostFree(ctx->h_progress);
    if (ctx->h_seg_words) (void)hipHostFree(ctx->h_seg_words);
    for (pngloss_hip_ctx *p : ctx->peers) if (p) pngloss_hip_destroy(p);
    ctx->peers.clear();
    delete ctx;
}

int pngloss_hip_optimize_batch_async(pngloss_hip_ctx *ctx, const pngloss_hip_image_desc *images, size_t n,
                                     unsigned quantization_strength, long bleed_divider, void *stream)
{
    return enqueue(ctx, images, n, nullptr, quantization_strength, bleed_divider, static_cast<hipStream_t>(stream));
}

int pngloss_hip_finish(pngloss_hip_ctx *ctx, pngloss_hip_result *results, size_t n) { return finish(ctx, results, n); }

int pngloss_hip_optimize_batch(pngloss_hip_ctx *ctx, const pngloss_hip_image_desc *images, size_t n,
                               unsigned quantization_strength, long bleed_divider, void *stream,
                               pngloss_hip_result *results)
{
    if (!ctx) return PNGLOSS_INVALID_ARGUMENT;
    ctx->sync_call = true; ctx->three_groups_ok = true;
    int rc = enqueue(ctx, images, n, nullptr, quantization_strength, bleed_divider, static_cast<hipStream_t>(stream));
    ctx->sync_call = false; ctx->three_groups_ok = false;
    if (rc) return rc;
    return finish(ctx, results, n);
}

/* chunks of one host window take turns at the two phases that are bound by the host's memory (staging in, fanning out), so that
 * chunk k+1 stages while chunk k computes instead of every chunk being in the same phase at the same time */
struct HostTurns { std::atomic<int> stage_turn{ 0 }; std::mutex out_mu; };

static int batch_host_one(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n, unsigned quantization_strength,
                      long bleed_divider, pngloss_hip_result *results, pngloss_hip_scanlines *lines,
                      pngloss_hip_zstream *zs, HostTurns *turns = nullptr, int my_turn = 0)
{
    /* whatever happens below, the next chunk must get its turn */
    struct TurnGuard { HostTurns *t; int mine; bool passed = false; void pass() { if (t && !passed) { while (t->stage_turn.load(std::memory_order_acquire) != mine) std::this_thread::yield(); t->stage_turn.store(mine + 1, std::memory_order_release); passed = true; } } ~TurnGuard() { pass(); } } turn{ turns, my_turn };
    if (!ctx || (n && !images)) return PNGLOSS_INVALID_ARGUMENT;
    PL_CHECK(hipSetDevice(ctx->device));
    /* one device arena for the whole batch, 256-B aligned: first [image | filter flags] of every image -- the part that has a pinned
     * mirror on the host --, behind them [emitted ids | emitted rows] of every image (those come back straight into the caller's memory) */
    std::vector<size_t> img_off(n), flt_off(n), ids_off(n), rows_off(n);
    std::vector<EmitTarget> emits(n);
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t px = (size_t)images[i].width * images[i].height;
        if (px && !images[i].rgba) return PNGLOSS_INVALID_ARGUMENT;
        img_off[i] = total; total = align_up(total + px * 4, 256);
        flt_off[i] = total; total = align_up(total + (images[i].row_filters ? images[i].height : 0), 256);
    }
    const size_t mirrored = total;
    for (size_t i = 0; i < n; i++) {
        const size_t px = (size_t)images[i].width * images[i].height;
        const bool want = ((lines && lines[i].scanlines && lines[i].filter_types) || (zs && zs[i].data)) && px;
        const uint32_t pitch = want ? (uint32_t)align_up((size_t)images[i].width * 4, 16) : 0;
        if (want && lines && lines[i].pitch < (size_t)images[i].width * 4) return PNGLOSS_INVALID_ARGUMENT;
        ids_off[i] = total; total = align_up(total + (want ? images[i].height : 0), 256);
        rows_off[i] = total; total = align_up(total + (size_t)pitch * (want ? images[i].height : 0), 256);
        emits[i].pitch = pitch;
    }
    /* persistent arena + pinned staging of the same layout; images are staged by a few host threads and go up as asynchronous
     * copies from pinned memory, one per image (pageable per-image copies were 0.33 s of a 1.6 s window of 256 720p files in round 1).
     * Copies and kernels of a context run on its own non-blocking stream, so that two contexts (the two halves of a window,
     * batch_host) overlap: one half's transfers with the other half's kernels. */
    const auto ta0 = std::chrono::steady_clock::now();
    if (total > ctx->arena_bytes) {
        if (ctx->d_arena) PL_CHECK(hipFree(ctx->d_arena));
        ctx->d_arena = nullptr; ctx->arena_bytes = 0;
        const size_t want = align_up(total + total / 8, 1 << 20);
        PL_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_arena), want));
        ctx->arena_bytes = want;
    }
    if (mirrored > ctx->pinned_bytes) {
        if (ctx->h_pinned) PL_CHECK(hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr; ctx->pinned_bytes = 0;
        const size_t want = align_up(mirrored + mirrored / 8, 1 << 20);
        PL_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pinned), want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    if (!ctx->copy_stream) PL_CHECK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (ctx->hooks.debug_seam) std::fprintf(stderr, "pngloss_hip: host window chunk %d: arena %zu MB + pinned mirror %zu MB ready after %.1f ms\n", my_turn, total >> 20, mirrored >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta0).count());
    char *const arena = ctx->d_arena;
    int rc = PNGLOSS_SUCCESS;
    std::vector<pngloss_hip_image_desc> descs(n);
    if (turns) while (turns->stage_turn.load(std::memory_order_acquire) != my_turn) std::this_thread::yield();
    const auto tu0 = std::chrono::steady_clock::now();
    {
        const unsigned nthreads = (unsigned)std::min<size_t>(12, std::max<size_t>(1, n));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() {
                for (size_t i = t; i < n; i += nthreads) {
                    const size_t px = (size_t)images[i].width * images[i].height;
                    if (px) std::memcpy(ctx->h_pinned + img_off[i], images[i].rgba, px * 4);
                }
            });
        for (auto &th : pool) th.join();
    }
    turn.pass();                                              /* the next chunk may stage while this one uploads and computes */
    for (size_t i = 0; i < n; i++) {
        const size_t px = (size_t)images[i].width * images[i].height;
        descs[i] = pngloss_hip_image_desc{ px ? arena + img_off[i] : nullptr,
                                           (px && images[i].row_filters) ? arena + flt_off[i] : nullptr, images[i].width, images[i].height };
        emits[i].d_ids = emits[i].pitch ? arena + ids_off[i] : nullptr;
        emits[i].d_rows = emits[i].pitch ? arena + rows_off[i] : nullptr;
        /* asynchronous DMA from pinned memory, one per image (the areas between images are filled by the kernels) */
        if (px && rc == PNGLOSS_SUCCESS &&
            hipMemcpyAsync(arena + img_off[i], ctx->h_pinned + img_off[i], px * 4, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
    }
    if (rc == PNGLOSS_SUCCESS && hipStreamSynchronize(ctx->copy_stream) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
    ctx->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
    std::vector<pngloss_hip_result> own_results;
    if (!results) { own_results.resize(n ? n : 1); results = own_results.data(); }
    if (rc == PNGLOSS_SUCCESS) {
        ctx->sync_call = true;                  /* (finish follows at once: no device-side wait on the copy stream, run_seg_engine) */
        rc = enqueue(ctx, descs.data(), n, nullptr, quantization_strength, bleed_divider, ctx->copy_stream, emits.data());
        ctx->sync_call = false;
    }
    if (rc == PNGLOSS_SUCCESS) rc = finish(ctx, results, n);
    /* a row without an acceptable filter (device status 65, pngloss_image.c:268-271) fails THAT image only: the others of
     * the batch are downloaded and the call reports PNGLOSS_INTERNAL_ABORT with the per-image status in results[] */
    const bool some_aborted = rc == PNGLOSS_INTERNAL_ABORT;
    if (some_aborted) rc = PNGLOSS_SUCCESS;
    const auto td0 = std::chrono::steady_clock::now();
    if (rc == PNGLOSS_SUCCESS) {
        /* pixels + filter flags of every image come back as one copy into the pinned mirror, then fan out on host threads */
        bool any_pixels = false;
        for (size_t i = 0; i < n; i++) {
            const bool stream_only = zs && zs[i].data && (zs[i].flags & PNGLOSS_HIP_Z_STREAM_ONLY);
            if (!stream_only && (size_t)images[i].width * images[i].height && results[i].status == 0) any_pixels = true;
        }
        if (any_pixels) {
            for (size_t i = 0; i < n && rc == PNGLOSS_SUCCESS; i++) {
                const size_t px = (size_t)images[i].width * images[i].height;
                const bool stream_only = zs && zs[i].data && (zs[i].flags & PNGLOSS_HIP_Z_STREAM_ONLY);
                if (!px || stream_only || results[i].status != 0) continue;
                /* the filter flags sit right behind the image in the arena: one copy takes both */
                const size_t bytes = images[i].row_filters ? flt_off[i] + images[i].height - img_off[i] : px * 4;
                if (hipMemcpyAsync(ctx->h_pinned + img_off[i], arena + img_off[i], bytes, hipMemcpyDeviceToHost, ctx->copy_stream) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
            }
            if (rc == PNGLOSS_SUCCESS && hipStreamSynchronize(ctx->copy_stream) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
            if (rc == PNGLOSS_SUCCESS) {
                std::unique_lock<std::mutex> out_lock;
                if (turns) out_lock = std::unique_lock<std::mutex>(turns->out_mu);
                const unsigned nthreads = (unsigned)std::min<size_t>(12, std::max<size_t>(1, n));
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < nthreads; t++)
                    pool.emplace_back([&, t]() {
                        for (size_t i = t; i < n; i += nthreads) {
                            const size_t px = (size_t)images[i].width * images[i].height;
                            const bool stream_only = zs && zs[i].data && (zs[i].flags & PNGLOSS_HIP_Z_STREAM_ONLY);
                            if (!px || stream_only || results[i].status != 0) continue;
                            std::memcpy(images[i].rgba, ctx->h_pinned + img_off[i], px * 4);
                            if (images[i].row_filters) std::memcpy(images[i].row_filters, ctx->h_pinned + flt_off[i], images[i].height);
                        }
                    });
                for (auto &th : pool) th.join();
            }
        }
    }
    for (size_t i = 0; i < n && rc == PNGLOSS_SUCCESS; i++) {
        const size_t px = (size_t)images[i].width * images[i].height;
        if (!px || results[i].status != 0) continue;
        if (emits[i].pitch && lines) {
            uint32_t fl = 0;
            if (hipMemcpy(&fl, ctx->h_jobs[i].out_flags, sizeof fl, hipMemcpyDeviceToHost) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
            const bool g = fl & PL_FLAG_GRAY, o = fl & PL_FLAG_OPAQUE;
            lines[i].color_type = g ? (o ? 0 : 4) : (o ? 2 : 6);
            const size_t rowbytes = (size_t)images[i].width * (g ? (o ? 1 : 2) : (o ? 3 : 4));
            if (hipMemcpy(lines[i].filter_types, emits[i].d_ids, images[i].height, hipMemcpyDeviceToHost) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
            if (hipMemcpy2D(lines[i].scanlines, lines[i].pitch, emits[i].d_rows, emits[i].pitch, rowbytes, images[i].height,
                            hipMemcpyDeviceToHost) != hipSuccess) rc = PNGLOSS_HIP_ERROR;
        }
    }
    ctx->download_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count();
    if (ctx->hooks.debug_seam)
        std::fprintf(stderr, "pngloss_hip: host window chunk %d: %zu images, wait+stage+upload %.1f ms, engine %.1f ms (enqueue..finish %.1f ms), download+fan-out %.1f ms\n", my_turn, n, ctx->upload_ms, ctx->engine_ms,
                     std::chrono::duration<double, std::milli>(td0 - tu0).count() - ctx->upload_ms, ctx->download_ms);
    if (zs && rc == PNGLOSS_SUCCESS) {
        /* the colour type decides the scanline length, so it is fetched before the deflate stage is laid out */
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<pl_deflate_image> dz;
        std::vector<size_t> who;
        for (size_t i = 0; i < n && rc == PNGLOSS_SUCCESS; i++) {
            zs[i].size = 0; zs[i].color_type = 6; zs[i].blocks[0] = zs[i].blocks[1] = zs[i].blocks[2] = 0;
            if (!emits[i].pitch || results[i].status != 0) continue;
            uint32_t fl = 0;
            if (hipMemcpy(&fl, ctx->h_jobs[i].out_flags, sizeof fl, hipMemcpyDeviceToHost) != hipSuccess) { rc = PNGLOSS_HIP_ERROR; break; }
            const bool g = fl & PL_FLAG_GRAY, o = fl & PL_FLAG_OPAQUE;
            zs[i].color_type = g ? (o ? 0 : 4) : (o ? 2 : 6);
            pl_deflate_image d{};
            d.d_filter_types = static_cast<const uint8_t *>(emits[i].d_ids);
            d.d_scanlines = static_cast<const uint8_t *>(emits[i].d_rows);
            d.pitch = emits[i].pitch;
            d.rowbytes = images[i].width * (g ? (o ? 1u : 2u) : (o ? 3u : 4u));
            d.height = images[i].height;
            d.out = zs[i].data;
            d.out_capacity = zs[i].capacity;
            dz.push_back(d);
            who.push_back(i);
        }
        if (rc == PNGLOSS_SUCCESS && !dz.empty()) {
            const hipError_t e = pl_deflate_images(dz.data(), dz.size(), nullptr);
            if (e == hipErrorInvalidValue) rc = PNGLOSS_INVALID_ARGUMENT;
            else if (e == hipErrorOutOfMemory) rc = PNGLOSS_OUT_OF_MEMORY_ERROR;
            else if (e != hipSuccess) rc = PNGLOSS_HIP_ERROR;
            for (size_t k = 0; k < dz.size() && rc == PNGLOSS_SUCCESS; k++) {
                zs[who[k]].size = dz[k].out_size;
                zs[who[k]].blocks[0] = dz[k].blocks_stored; zs[who[k]].blocks[1] = dz[k].blocks_fixed; zs[who[k]].blocks[2] = dz[k].blocks_dynamic;
            }
        }
        ctx->deflate_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ctx->hooks.debug_seam) std::fprintf(stderr, "pngloss_hip: host window: deflate stage %.1f ms for %zu images\n", ctx->deflate_ms, dz.size());
    }
    if (rc == PNGLOSS_HIP_ERROR) std::fprintf(stderr, "pngloss_hip: batch transfer or kernel failure: %s\n", hipGetErrorString(hipGetLastError()));
    if (rc == PNGLOSS_SUCCESS && some_aborted) rc = PNGLOSS_INTERNAL_ABORT;
    return rc;
}


/* A window of host images: in two halves on two contexts of the same device, a host thread each, when it is large enough -- the
 * second half is staged and uploaded while the first one computes, the first one is downloaded while the second one computes
 * (profiles/r02_host_seam.txt: staging + PCIe were 0.26 s next to a 0.28 s engine for 256 x 720p, one after the other).  The deflate
 * stage (zs) keeps its single pass: it sorts the whole window's scanlines as one stream. */
static int batch_host(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n, unsigned quantization_strength,
                      long bleed_divider, pngloss_hip_result *results, pngloss_hip_scanlines *lines,
                      pngloss_hip_zstream *zs = nullptr)
{
    if (!ctx || (n && !images)) return PNGLOSS_INVALID_ARGUMENT;
    ctx->split_last = false;
    /* Chunks, each on its own context and stream, staggered by the staging turns: chunk k+1 is staged and uploaded while chunk k
     * computes, chunk k is downloaded while chunk k+1 computes.  Measured on 256 x 1280x720 (profiles/r03_host_seam.txt): one chunk
     * 0.237 s, two 0.221 s, four 0.218 s (the engine alone: 0.176 s) -- two it is; PNGLOSS_HIP_SPLIT=k for experiments. */
    size_t K = n >= 16 ? 2 : 1;
    if (ctx->hooks.split) K = (size_t)ctx->hooks.split;
    if (K > n) K = n ? n : 1;
    if (K <= 1 || zs || ctx->hooks.no_split) return batch_host_one(ctx, images, n, quantization_strength, bleed_divider, results, lines, zs);
    while (ctx->peers.size() < K - 1) {
        pngloss_hip_ctx *p = pngloss_hip_create(ctx->device);
        if (!p) break;
        ctx->peers.push_back(p);
    }
    K = std::min(K, ctx->peers.size() + 1);
    if (K <= 1) return batch_host_one(ctx, images, n, quantization_strength, bleed_divider, results, lines, zs);
    /* cut where the pixels are: equal shares */
    size_t total = 0;
    for (size_t i = 0; i < n; i++) total += (size_t)images[i].width * images[i].height;
    std::vector<size_t> first(K + 1, n);
    first[0] = 0;
    {
        size_t run = 0, c = 1;
        for (size_t i = 0; i < n && c < K; i++) {
            run += (size_t)images[i].width * images[i].height;
            if (run * K >= total * c && i + 1 < n) first[c++] = i + 1;
        }
        for (; c < K; c++) first[c] = n;
    }
    std::vector<pngloss_hip_result> own;
    if (!results) { own.resize(n); results = own.data(); }
    HostTurns turns;
    std::vector<int> rcs(K, PNGLOSS_SUCCESS);
    auto run_chunk = [&](size_t c) {
        pngloss_hip_ctx *cc = c == 0 ? ctx : ctx->peers[c - 1];
        rcs[c] = batch_host_one(cc, images + first[c], first[c + 1] - first[c], quantization_strength, bleed_divider, results + first[c],
                                lines ? lines + first[c] : nullptr, nullptr, &turns, (int)c);
    };
    std::vector<std::thread> others;
    for (size_t c = 1; c < K; c++) others.emplace_back(run_chunk, c);
    run_chunk(0);
    for (auto &t : others) t.join();
    ctx->split_last = true;
    ctx->chunk_first.assign(first.begin(), first.begin() + (long)K);
    for (size_t c = 1; c < K; c++) {
        pngloss_hip_ctx *p = ctx->peers[c - 1];
        ctx->engine_ms = std::max(ctx->engine_ms, p->engine_ms);
        ctx->total_ms = std::max(ctx->total_ms, p->total_ms);
        ctx->upload_ms += p->upload_ms; ctx->download_ms += p->download_ms;
    }
    bool aborted = false;
    for (size_t c = 0; c < K; c++) {
        if (rcs[c] == PNGLOSS_INTERNAL_ABORT) aborted = true;
        else if (rcs[c] != PNGLOSS_SUCCESS) return rcs[c];
    }
    return aborted ? PNGLOSS_INTERNAL_ABORT : PNGLOSS_SUCCESS;
}

int pngloss_hip_optimize_batch_host(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                    unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results)
{
    return batch_host(ctx, images, n, quantization_strength, bleed_divider, results, nullptr);
}

/* ---- all the GPUs of the node (replaces the one-file-at-a-time loop of /root/reference/src/pngloss.c:173-208 for a whole
 * node): one context per device, images dealt out by size (longest-processing-time first), one host thread per context,
 * results back in input order.  No collective: images are independent. ---------------------------------------------- */
struct pngloss_hip_multi {
    std::vector<pngloss_hip_ctx *> ctx;
};

void pngloss_hip_multi_split(const pngloss_hip_host_image *images, size_t n, int parts, int *owner)
{
    /* LPT greedy, deterministic: items by descending pixel count (ties: lower index first), each to the least loaded part
     * (ties: lower part) -- the same rule as pngloss_amd/shard.py:lpt_partition */
    if (parts < 1) parts = 1;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return (uint64_t)images[a].width * images[a].height > (uint64_t)images[b].width * images[b].height;
    });
    std::vector<uint64_t> load((size_t)parts, 0);
    for (size_t k = 0; k < n; k++) {
        const size_t i = order[k];
        int best = 0;
        for (int p2 = 1; p2 < parts; p2++) if (load[(size_t)p2] < load[(size_t)best]) best = p2;
        owner[i] = best;
        load[(size_t)best] += (uint64_t)images[i].width * images[i].height;
    }
}

pngloss_hip_multi *pngloss_hip_multi_create(const char *devices)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        std::fprintf(stderr, "pngloss_hip: no HIP device available\n");
        return nullptr;
    }
    if (!devices || !*devices) devices = std::getenv("PNGLOSS_DEVICES");
    std::vector<int> want;
    if (devices && *devices) {
        const char *p2 = devices;
        while (*p2) {
            char *end = nullptr;
            const long d = std::strtol(p2, &end, 10);
            if (end == p2 || d < 0 || d >= visible) {
                std::fprintf(stderr, "pngloss_hip: bad device list \"%s\" (%d device(s) visible)\n", devices, visible);
                return nullptr;
            }
            want.push_back((int)d);
            p2 = end;
            while (*p2 == ',' || *p2 == ' ') p2++;
        }
    } else {
        for (int d = 0; d < visible; d++) want.push_back(d);
    }
    if (want.empty()) return nullptr;
    pngloss_hip_multi *m = new (std::nothrow) pngloss_hip_multi;
    if (!m) return nullptr;
    for (int d : want) {
        pngloss_hip_ctx *c = pngloss_hip_create(d);
        if (!c) { pngloss_hip_multi_destroy(m); return nullptr; }
        m->ctx.push_back(c);
    }
    return m;
}

void pngloss_hip_multi_destroy(pngloss_hip_multi *m)
{
    if (!m) return;
    for (pngloss_hip_ctx *c : m->ctx) pngloss_hip_destroy(c);
    delete m;
}

int pngloss_hip_multi_count(const pngloss_hip_multi *m) { return m ? (int)m->ctx.size() : 0; }

int pngloss_hip_multi_optimize_batch_host(pngloss_hip_multi *m, const pngloss_hip_host_image *images, size_t n,
                                          unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                          pngloss_hip_scanlines *scanlines, pngloss_hip_zstream *streams)
{
    if (!m || m->ctx.empty() || (n && !images)) return PNGLOSS_INVALID_ARGUMENT;
    const int parts = (int)m->ctx.size();
    std::vector<int> owner(n ? n : 1, 0);
    pngloss_hip_multi_split(images, n, parts, owner.data());
    std::vector<int> rcs((size_t)parts, PNGLOSS_SUCCESS);
    std::vector<std::thread> pool;
    for (int p2 = 0; p2 < parts; p2++)
        pool.emplace_back([&, p2]() {
            std::vector<size_t> mine;
            for (size_t i = 0; i < n; i++) if (owner[i] == p2) mine.push_back(i);
            if (mine.empty()) return;
            std::vector<pngloss_hip_host_image> im(mine.size());
            std::vector<pngloss_hip_result> rs(mine.size());
            std::vector<pngloss_hip_scanlines> ln(scanlines ? mine.size() : 0);
            std::vector<pngloss_hip_zstream> zz(streams ? mine.size() : 0);
            for (size_t k = 0; k < mine.size(); k++) {
                im[k] = images[mine[k]];
                if (scanlines) ln[k] = scanlines[mine[k]];
                if (streams) zz[k] = streams[mine[k]];
            }
            rcs[(size_t)p2] = batch_host(m->ctx[(size_t)p2], im.data(), im.size(), quantization_strength, bleed_divider, rs.data(),
                                         scanlines ? ln.data() : nullptr, streams ? zz.data() : nullptr);
            for (size_t k = 0; k < mine.size(); k++) {
                if (results) results[mine[k]] = rs[k];
                if (scanlines) scanlines[mine[k]] = ln[k];
                if (streams) streams[mine[k]] = zz[k];
            }
        });
    for (auto &th : pool) th.join();
    int worst = PNGLOSS_SUCCESS;
    for (int rc : rcs) if (rc != PNGLOSS_SUCCESS && (worst == PNGLOSS_SUCCESS || worst == PNGLOSS_INTERNAL_ABORT)) worst = rc;
    return worst;
}

int pngloss_hip_optimize_batch_host_emit(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                         unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                         pngloss_hip_scanlines *scanlines)
{
    return batch_host(ctx, images, n, quantization_strength, bleed_divider, results, scanlines);
}

int pngloss_hip_optimize_batch_host_zlib(pngloss_hip_ctx *ctx, const pngloss_hip_host_image *images, size_t n,
                                         unsigned quantization_strength, long bleed_divider, pngloss_hip_result *results,
                                         pngloss_hip_zstream *streams)
{
    if (!streams && n) return PNGLOSS_INVALID_ARGUMENT;
    return batch_host(ctx, images, n, quantization_strength, bleed_divider, results, nullptr, streams);
}

size_t pngloss_hip_zlib_bound(uint32_t width, uint32_t height) { return pl_deflate_bound(width, height); }
double pngloss_hip_last_deflate_ms(const pngloss_hip_ctx *ctx) { return ctx ? ctx->deflate_ms : -1.0; }

double pngloss_hip_last_engine_ms(const pngloss_hip_ctx *ctx) { return ctx ? ctx->engine_ms : -1.0; }
double pngloss_hip_last_total_ms(const pngloss_hip_ctx *ctx) { return ctx ? ctx->total_ms : -1.0; }

/* the context that ran image `index` of the last (possibly split) host window, and the image's index there */
static pngloss_hip_ctx *chunk_of(pngloss_hip_ctx *ctx, size_t &index)
{
    if (!ctx || !ctx->split_last || index < ctx->n_last) return ctx;
    for (size_t c = ctx->chunk_first.size(); c-- > 1;)
        if (index >= ctx->chunk_first[c] && c - 1 < ctx->peers.size() && ctx->peers[c - 1]) { index -= ctx->chunk_first[c]; return ctx->peers[c - 1]; }
    return ctx;
}

int pngloss_hip_last_histogram(pngloss_hip_ctx *ctx, size_t index, uint32_t *hist256)
{
    ctx = chunk_of(ctx, index);
    if (!ctx || !hist256 || index >= ctx->n_last || ctx->pending) return PNGLOSS_INVALID_ARGUMENT;
    PL_CHECK(hipSetDevice(ctx->device));
    PL_CHECK(hipMemcpy(hist256, ctx->h_jobs[index].final_hist, sizeof(uint32_t) * PL_NSYM, hipMemcpyDeviceToHost));
    return PNGLOSS_SUCCESS;
}

int pngloss_hip_png_decode_batch_host(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n)
{
    return pngloss_hip_png_decode_batch_host_status(ctx, src, n, nullptr);
}

} /* extern "C" */
namespace {
/* d_out == nullptr: the decoded images are downloaded to src[i].rgba.  Else they STAY on the device, in the context's frame arena (apart
 * from the workspace the optimiser carves up), and d_out[i] receives their device pointers. */
/* zs != nullptr: src[i].scanlines is not used; the scanlines are INFLATED ON THE DEVICE from zs[i] (the concatenated IDAT payloads), one wave per file */
struct ZRef { const unsigned char *z; size_t bytes; };
/* per_image: set once the batch as a whole went through -- from there on the return value is the worst PER-IMAGE status and status[] explains it.
 * A return before that point is a failure of the WHOLE batch (bad argument, allocation, copy, launch, device fault): png_decode_common then
 * writes that code into every status[i], so that a caller who looks at status[] alone never takes an undecoded image for a decoded one. */
int png_decode_body(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, int *status, void **d_out, hipStream_t stream, const ZRef *zs, bool &per_image)
{
    if (!ctx || (!src && n)) return PNGLOSS_INVALID_ARGUMENT;
    if (status) for (size_t i = 0; i < n; i++) status[i] = PNGLOSS_SUCCESS;
    if (d_out) for (size_t i = 0; i < n; i++) d_out[i] = nullptr;
    if (!n) { per_image = true; return PNGLOSS_SUCCESS; }
    if (ctx->pending) {
        /* (the workspace this call carves up belongs to the batch in flight) */
        std::fprintf(stderr, "pngloss_hip: a batch is in flight on this context; call pngloss_hip_finish first\n");
        return PNGLOSS_INVALID_ARGUMENT;
    }
    PL_CHECK(hipSetDevice(ctx->device));
    std::vector<PrJob> jobs(n);
    std::vector<size_t> raw_off(n), out_off(n), last_off(n), prog_off(n), z_off(n);
    static const unsigned char bad_stream[6] = { 0, 0, 0, 0, 0, 0 };       /* (CMF 0: not deflate) */
    std::vector<ZRef> zsub(zs ? n : 0);
    uint32_t max_bands = 0;
    size_t total = align_up(sizeof(PrJob) * n, 256) + 2 * align_up(sizeof(int32_t) * n, 256) + align_up(sizeof(PliStream) * n, 256), ftotal = 0;
    const size_t jobs_bytes = align_up(sizeof(PrJob) * n, 256), st_bytes = align_up(sizeof(int32_t) * n, 256);
    for (size_t i = 0; i < n; i++) {
        if ((!zs && !src[i].scanlines) || (zs && !zs[i].z) || (!d_out && !src[i].rgba)) return PNGLOSS_INVALID_ARGUMENT;
        /* a stream too short to be zlib's (header + one block + Adler-32) or beyond the inflater's 32-bit positions is THAT file's problem: it gets
         * status 25 like any other stream the inflater refuses (it is handed a six-byte stream with an invalid header), the batch goes on */
        if (zs && (zs[i].bytes < 6 || zs[i].bytes > 0xFFFFFFF0u)) zsub[i] = ZRef{ bad_stream, sizeof bad_stream };
        else if (zs) zsub[i] = zs[i];
        if (!pr_format(jobs[i].F, src[i].width, src[i].height, src[i].color_type, src[i].bit_depth, src[i].palette, src[i].palette_entries, src[i].trns, src[i].trns_bytes)) {
            std::fprintf(stderr, "pngloss_hip: image %zu: colour type %d with bit depth %d (or an empty image / a palette image without PLTE) is not a PNG format\n", i, src[i].color_type, src[i].bit_depth);
            return PNGLOSS_INVALID_ARGUMENT;
        }
        raw_off[i] = total; total += align_up(((size_t)jobs[i].F.rowbytes + 1) * src[i].height, 256);
        if (zs) {
            if (((size_t)jobs[i].F.rowbytes + 1) * src[i].height > 0xFFFFFFF0u) return PNGLOSS_INVALID_ARGUMENT;     /* (32-bit positions in the inflater) */
            z_off[i] = total; total += align_up(zsub[i].bytes + 16, 256);
        }
        const size_t out_bytes = align_up((size_t)src[i].width * src[i].height * 4, 256);
        if (d_out) { out_off[i] = ftotal; ftotal += out_bytes; } else { out_off[i] = total; total += out_bytes; }
        /* per band of PR_ROWS rows: its last row (for the band below) and a progress word */
        jobs[i].nbands = (src[i].height + PR_ROWS - 1) / PR_ROWS;
        jobs[i].lastpitch = (uint32_t)align_up(jobs[i].F.rowbytes, 256);
        max_bands = std::max(max_bands, jobs[i].nbands);
        last_off[i] = total; total += (size_t)jobs[i].lastpitch * jobs[i].nbands;
        prog_off[i] = total; total += align_up(sizeof(uint32_t) * jobs[i].nbands, 256);
    }
    const auto tr0 = std::chrono::steady_clock::now();
    auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
    int rc = ensure_ws(ctx, total);
    if (rc) return rc;
    if (d_out && ftotal > ctx->frames_bytes) {
        if (ctx->d_frames) PL_CHECK(hipFree(ctx->d_frames));
        ctx->d_frames = nullptr; ctx->frames_bytes = 0;
        const size_t want = align_up(ftotal + ftotal / 4, 1 << 20);
        PL_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_frames), want));
        ctx->frames_bytes = want;
    }
    const double ms_ws = ms_since();
    char *b = ctx->d_ws, *fb = d_out ? ctx->d_frames : ctx->d_ws;
    int32_t *d_status = reinterpret_cast<int32_t *>(b + jobs_bytes), *d_zstatus = reinterpret_cast<int32_t *>(b + jobs_bytes + st_bytes);
    PliStream *d_zjobs = reinterpret_cast<PliStream *>(b + jobs_bytes + 2 * st_bytes);
    std::vector<PliStream> zjobs(zs ? n : 0);
    PL_CHECK(hipMemsetAsync(d_status, 0, 2 * st_bytes, stream));
    for (size_t i = 0; i < n; i++) {
        jobs[i].raw = reinterpret_cast<const uint8_t *>(b + raw_off[i]);
        jobs[i].rgba = reinterpret_cast<uint32_t *>(fb + out_off[i]);
        jobs[i].lastrow = reinterpret_cast<uint8_t *>(b + last_off[i]);
        jobs[i].progress = reinterpret_cast<uint32_t *>(b + prog_off[i]);
        PL_CHECK(hipMemsetAsync(b + prog_off[i], 0, sizeof(uint32_t) * jobs[i].nbands, stream));
        jobs[i].status = d_status + i;
        /* (from pinned memory -- pngloss_hip_pinned_alloc -- this is one DMA; from pageable memory the runtime stages it: 33 ms against 1.3 for 64 MiB) */
        if (zs) {
            PL_CHECK(hipMemcpyAsync(b + z_off[i], zsub[i].z, zsub[i].bytes, hipMemcpyHostToDevice, stream));
            zjobs[i].z = reinterpret_cast<const uint8_t *>(b + z_off[i]); zjobs[i].zbytes = (uint32_t)zsub[i].bytes;
            zjobs[i].out = reinterpret_cast<uint8_t *>(b + raw_off[i]); zjobs[i].expect = (uint32_t)(((size_t)jobs[i].F.rowbytes + 1) * src[i].height);
            zjobs[i].status = d_zstatus + i;
        } else
        PL_CHECK(hipMemcpyAsync(b + raw_off[i], src[i].scanlines, ((size_t)jobs[i].F.rowbytes + 1) * src[i].height, hipMemcpyHostToDevice, stream));
    }
    PL_CHECK(hipMemcpyAsync(b, jobs.data(), sizeof(PrJob) * n, hipMemcpyHostToDevice, stream));
    if (zs) {
        PL_CHECK(hipMemcpyAsync(d_zjobs, zjobs.data(), sizeof(PliStream) * n, hipMemcpyHostToDevice, stream));
        PL_CHECK(pl_launch_inflate(d_zjobs, n, stream));
    }
    const bool seam_dbg = ctx->hooks.debug_seam;
    double ms_up = 0, ms_k = 0;
    if (seam_dbg) { PL_CHECK(hipStreamSynchronize(stream)); ms_up = ms_since(); }
    PL_CHECK(pl_launch_png_decode(reinterpret_cast<const PrJob *>(b), n, max_bands, stream));
    if (seam_dbg) { PL_CHECK(hipStreamSynchronize(stream)); ms_k = ms_since(); }
    std::vector<int32_t> st(n), zst(n, 0);
    if (zs) PL_CHECK(hipMemcpyAsync(zst.data(), d_zstatus, sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream));
    if (!d_out)
        for (size_t i = 0; i < n; i++)
            PL_CHECK(hipMemcpyAsync(src[i].rgba, b + out_off[i], (size_t)src[i].width * src[i].height * 4, hipMemcpyDeviceToHost, stream));
    PL_CHECK(hipMemcpyAsync(st.data(), d_status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream));
    PL_CHECK(hipStreamSynchronize(stream));
    if (seam_dbg) std::fprintf(stderr, "pngloss_hip: read side: %zu files, workspace %zu MB ready after %.1f ms, upload %.1f ms, unfilter + expand %.1f ms, %s %.1f ms\n", n, (total + ftotal) >> 20, ms_ws, ms_up - ms_ws, ms_k - ms_up, d_out ? "status (the frames stay on the device)" : "download", ms_since() - ms_k);
    /* every image has been decoded (and downloaded); the ones that failed say so -- one damaged file does not take the window with it */
    per_image = true;
    int worst = PNGLOSS_SUCCESS;
    for (size_t i = 0; i < n; i++) {
        if (d_out) d_out[i] = fb + out_off[i];
        if (zst[i]) {
            /* the stream is not one the device inflater takes (damaged, or beyond what it checks): the caller reads the file on the host */
            std::fprintf(stderr, "pngloss_hip: image %zu: the device inflater stopped (code %d): corrupt or unusual zlib stream\n", i, zst[i]);
            if (status) status[i] = 25;
            if (worst == PNGLOSS_SUCCESS) worst = 25;
            continue;
        }
        if (!st[i]) continue;
        const int code = st[i] == 25 ? 25 : PNGLOSS_HIP_ERROR;
        if (st[i] == 25) std::fprintf(stderr, "pngloss_hip: image %zu: a scanline has a filter type beyond 4 (corrupt stream)\n", i);
        else std::fprintf(stderr, "pngloss_hip: image %zu: the row bands of the decoder lost step (internal error %d)\n", i, st[i]);
        if (status) status[i] = code;
        if (worst == PNGLOSS_SUCCESS || code == PNGLOSS_HIP_ERROR) worst = code;
    }
    return worst;
}
int png_decode_common(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, int *status, void **d_out, hipStream_t stream, const ZRef *zs = nullptr)
{
    bool per_image = false;
    const int rc = png_decode_body(ctx, src, n, status, d_out, stream, zs, per_image);
    if (!per_image && rc != PNGLOSS_SUCCESS) {
        /* the batch as a whole failed: nothing in rgba / d_out is a decoded image */
        if (status) for (size_t i = 0; i < n; i++) status[i] = rc;
        if (d_out) for (size_t i = 0; i < n; i++) d_out[i] = nullptr;
    }
    return rc;
}
} // namespace
extern "C" {

int pngloss_hip_png_decode_batch_host_status(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, int *status)
{
    return png_decode_common(ctx, src, n, status, nullptr, nullptr);
}

int pngloss_hip_png_decode_batch_device(pngloss_hip_ctx *ctx, const pngloss_hip_png_source *src, size_t n, void **d_rgba, int *status, void *stream)
{
    if (!d_rgba && n) return PNGLOSS_INVALID_ARGUMENT;
    return png_decode_common(ctx, src, n, status, d_rgba, static_cast<hipStream_t>(stream));
}

int pngloss_hip_png_decode_batch_device_z(pngloss_hip_ctx *ctx, const pngloss_hip_png_zsource *zsrc, size_t n, void **d_rgba, int *status, void *stream)
{
    if ((!d_rgba || !zsrc) && n) return PNGLOSS_INVALID_ARGUMENT;
    std::vector<pngloss_hip_png_source> src(n);
    std::vector<ZRef> zs(n);
    for (size_t i = 0; i < n; i++) {
        src[i] = pngloss_hip_png_source{ nullptr, zsrc[i].width, zsrc[i].height, zsrc[i].color_type, zsrc[i].bit_depth, zsrc[i].palette, zsrc[i].palette_entries, zsrc[i].trns, zsrc[i].trns_bytes, nullptr };
        zs[i] = ZRef{ zsrc[i].zstream, zsrc[i].zbytes };
    }
    return png_decode_common(ctx, src.data(), n, status, d_rgba, static_cast<hipStream_t>(stream), zs.data());
}

int pngloss_hip_set_option(pngloss_hip_ctx *ctx, const char *name, const char *value)
{
    if (!ctx || !name || !value) return PNGLOSS_INVALID_ARGUMENT;
    if (ctx->pending) return PNGLOSS_INVALID_ARGUMENT;
    if (std::strcmp(name, "engine") == 0) {
        static const char *const known[] = { "auto", "seg", "wg", "lead", "legacy", "mix", "rows" };
        for (const char *k : known)
            if (std::strcmp(value, k) == 0) { ctx->opt_engine = std::strcmp(value, "auto") == 0 ? "" : value; return PNGLOSS_SUCCESS; }
        return PNGLOSS_INVALID_ARGUMENT;
    }
    if (std::strcmp(name, "launch_groups") == 0) {
        /* launch groups of a large batch on the segment engine through the SYNCHRONOUS entry point: "auto" / "2" (default) or "3" -- for a process that never hands the
         * asynchronous entry a stream of its own (run_seg_engine: a third engine stream in the process halves every later engine run that waits on a caller's stream) */
        if (std::strcmp(value, "auto") == 0 || std::strcmp(value, "2") == 0) { ctx->opt_launch_groups = 0; return PNGLOSS_SUCCESS; }
        if (std::strcmp(value, "3") == 0) { ctx->opt_launch_groups = 3; return PNGLOSS_SUCCESS; }
        return PNGLOSS_INVALID_ARGUMENT;
    }
    return PNGLOSS_INVALID_ARGUMENT;
}

void *pngloss_hip_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void pngloss_hip_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

int pngloss_hip_last_engine_info(pngloss_hip_ctx *ctx, size_t index, int32_t info[8])
{
    ctx = chunk_of(ctx, index);
    if (!ctx || !info || index >= ctx->n_last || ctx->pending) return PNGLOSS_INVALID_ARGUMENT;
    PL_CHECK(hipSetDevice(ctx->device));
    int32_t r[64] = { 0 };
    PL_CHECK(hipMemcpy(r, ctx->h_jobs[index].result, sizeof r, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) info[i] = 0;
    if (r[20] == 3) { info[0] = 3; info[1] = r[5]; info[2] = r[4]; info[3] = r[6]; info[4] = r[7]; info[5] = r[17]; info[6] = ctx->seg_groups; info[7] = ctx->seg_async_wait ? 1 : 0; }
    else if (r[20] == 4) { info[0] = 4; info[1] = r[5]; }
    else { info[0] = 0; info[1] = r[5]; info[2] = r[4]; info[3] = r[21]; }
    return PNGLOSS_SUCCESS;
}

#define PL_STR2(x) #x
#define PL_STR(x) PL_STR2(x)
const char *pngloss_hip_version(void)
{
#ifdef PL_DEBUG_FORCE_FILTER
    return "pngloss_hip 0.5 DEBUGGING BUILD -DPL_DEBUG_FORCE_FILTER=" PL_STR(PL_DEBUG_FORCE_FILTER) ": one candidate wins every row, results are NOT the reference's (gfx950)";
#else
    return "pngloss_hip 0.5 (gfx950; row engines: segment-parallel v4 (units and segments from seeds, launch groups) + band-leader v2 + row statistics (strength 0); seam: pngloss_image.h:14-29)";
#endif
}

/* ---- the reference's seam ------------------------------------------------------------------------------- */

int optimize_with_rows(unsigned char **rows, uint32_t width, uint32_t height, unsigned char *row_filters,
                       bool verbose, uint_fast8_t quantization_strength, int_fast16_t bleed_divider)
{
    return run_host_image(rows, width, height, 4, 0, row_filters, verbose, quantization_strength, bleed_divider);
}

void optimize_with_stride(unsigned char *pixels, uint32_t width, uint32_t height, uint32_t stride, bool verbose,
                          uint_fast8_t quantization_strength, int_fast16_t bleed_divider)
{
    std::vector<unsigned char *> rows(height);
    for (uint32_t i = 0; i < height; i++) rows[i] = pixels + (size_t)i * stride;
    (void)optimize_with_rows(rows.data(), width, height, nullptr, verbose, quantization_strength, bleed_divider);
}

void optimizeForAverageFilter(unsigned char pixels[], int width, int height, int quantization)
{
    /* pngloss_image.c:29-38: RGBA, stride 4*w, bleed divider fixed at 2 */
    optimize_with_stride(pixels, (uint32_t)width, (uint32_t)height, (uint32_t)width * 4u, false,
                         (uint_fast8_t)quantization, 2);
}

int optimize_image(pngloss_image *image, unsigned char *row_filters, bool verbose, uint_fast8_t quantization_strength,
                   int_fast16_t bleed_divider)
{
    if (!image || image->bytes_per_pixel < 1 || image->bytes_per_pixel > 4) return PNGLOSS_INVALID_ARGUMENT;
    const uint32_t bpp = (uint32_t)image->bytes_per_pixel;
    return run_host_image(image->rows, image->width, image->height, bpp, bpp, row_filters, verbose,
                          quantization_strength, bleed_divider);
}

} /* extern "C" */
