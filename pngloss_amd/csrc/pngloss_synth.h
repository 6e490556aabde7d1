/* Deterministic synthetic RGBA8 frames + FNV-1a-64. Host utility, plain C. */
#ifndef PNGLOSS_SYNTH_H
#define PNGLOSS_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Fill width*height*4 bytes (row-major RGBA8, no padding). mode 0..5, see pngloss_synth.c. */
void pngloss_synth_rgba(unsigned char *rgba, uint32_t width, uint32_t height, int mode, uint64_t frame);

/* FNV-1a 64-bit (offset 0xcbf29ce484222325, prime 0x100000001b3). */
uint64_t pngloss_fnv1a64(const unsigned char *data, size_t n);

/* Same hash with a caller-chosen offset basis.  SURVEY.md Appendix B's digest table was produced with the basis
 * 0x14650fb0739d0383 (the decimal FNV basis 14695981039346656037 with its last digit dropped), so tests that check
 * against that table pass PNGLOSS_SURVEY_FNV_BASIS here. */
#define PNGLOSS_SURVEY_FNV_BASIS 0x14650fb0739d0383ull
uint64_t pngloss_fnv1a64_seed(const unsigned char *data, size_t n, uint64_t basis);

#ifdef __cplusplus
}
#endif
#endif
