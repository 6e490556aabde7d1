/*
 * pngloss_port.h -- CPU ORACLE for the pngloss filter+quantise hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (pngloss_amd/, include/, the pngloss CLI) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / the timed CPU baseline -- never as the thing that produces results.
 *
 * This is a from-scratch restatement (not a copy) of the algorithm in the reference's
 *   /root/reference/src/pngloss_image.c   (optimize_with_rows :52, optimize_image :159)
 *   /root/reference/src/optimize_state.c  (init :28, run :114, row :292, diffuse :390, adaptive :492, ulog2 :565)
 *   /root/reference/src/color_delta.c     (:4, :43, :60)
 * organised the way the HIP kernels are organised (row pre-pass / serial chain / vectorisable post-pass), so that
 * every algebraic shortcut the GPU code takes is first proven bit-exact on the CPU against the real reference
 * (oracle/_ref/libpngloss_ref.so, built from the reference sources in place by oracle/Makefile).
 *
 * Parity status: PINNED -- tests/test_oracle.py checks this port against (a) the real reference on seeded inputs,
 * (b) the committed golden fixtures under tests/golden/ (generated from the real reference by
 * tests/golden/make_golden.py) and (c) the reference-measured digests of SURVEY.md Appendix B.
 */
#ifndef PNGLOSS_PORT_H
#define PNGLOSS_PORT_H

#include <stdbool.h>
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Optional per-row trace (all arrays caller-allocated, any may be NULL). */
typedef struct {
    uint64_t *cost;          /* [height*5] cost of each candidate filter at the strength finally used (UINT64_MAX = rejected) */
    uint8_t  *strength_used; /* [height]   strength at which the row was accepted */
    uint8_t  *winner;        /* [height]   0..4 */
    uint32_t *final_hist;    /* [256]      symbol histogram after the last row */
} port_trace;

/* Same contract as the reference's optimize_with_rows (pngloss_image.h:21-25): RGBA8 rows modified in place,
 * row_filters (nullable) receives PNG_FILTER_* flag values 0x08,0x10,0x20,0x40,0x80.  Returns 0 or 17 (OOM). */
int port_optimize_with_rows(unsigned char **rows, uint32_t width, uint32_t height, unsigned char *row_filters,
                            bool verbose, uint_fast8_t quantization_strength, int_fast16_t bleed_divider);

/* Lower seam == the reference's optimize_image (pngloss_image.h:26-29) on a packed, contiguous bpp-byte image. */
int port_optimize_packed(unsigned char *pix, uint32_t width, uint32_t height, uint32_t bpp,
                         unsigned char *row_filters, unsigned strength, long bleed, port_trace *trace);

/* Building blocks, exported so the tests can compare them one by one with the HIP kernels / the reference. */
void port_classify(const unsigned char *const *rows, uint32_t width, uint32_t height, int *grayscale, int *opaque);
void port_orig_histograms(const unsigned char *pix, uint32_t width, uint32_t height, uint32_t bpp, uint32_t out[5][256]);
int  port_adaptive_filter(const unsigned char *above /*nullable*/, const unsigned char *row, uint32_t width, uint32_t bpp);
/* parts[0..4] = t ("twos"), h ("threes"), f ("fours"), v ("five"), rem (remainder to x+1) */
void port_sierra_split(int diff16, long bleed, int parts[5]);
unsigned port_symbol_cost(uint32_t freq);

/* 0 (default): straightforward chain.  1: the speculative-channels + rank-key formulation of the round-1 HIP chains.
 * 2: the band-leader formulation of the round-2 chains (decision by tracked band leaders; every deferred bump must leave the
 * bumped bin strictly below the leader of every usable band that holds it -- band_watch_breaks; a clamped band that leaves a
 * single value is settled without the histogram ("light" pixels); exact evaluation + rescan otherwise).  Same results, proven by tests/test_oracle.py; process-global,
 * test use only. */
void port_set_chain_variant(int variant);
/* debugging aid: f >= 0 makes candidate filter f the winner of every row (-1: normal) */
void port_set_force_filter(int f);
/* statistics of variant 2: [0..7] all chains, [8+8f..] chain f: pixels, fast, light (fast pixels with a channel whose clamped band is
 * a single value), slow:unusable band or watched relation, slow:leader clamped away, slow:forced symbol, band scans, rows */
void port_lead_stats(unsigned long long out[48], int reset);

#ifdef __cplusplus
}
#endif
#endif
