/*
 * inflate_host.cpp -- TEST INFRASTRUCTURE: the device inflater's body (pngloss_amd/csrc/pl_inflate_core.h, the same source hipcc compiles
 * into pl_inflate.hip's kernel) run on the CPU with lane loops, so that the CPU suite can check it against zlib without a GPU.
 * Never shipped, never loaded by the product.
 *   inflate_host(z, zbytes, out, expect) -> 0 or a PLI_E_* code
 *   inflate_host_stats(out8) : rounds, sets, literals of runs, matches, matched bytes, symbols with a long code, runs -- since the last call
 */
static unsigned long long pli_stat[8];
#define PLI_STAT(i, n) (pli_stat[i] += (n))
#include "../../pngloss_amd/csrc/pl_inflate_core.h"
#include <vector>
#include <cstring>

extern "C" int inflate_host(const unsigned char *z, unsigned zbytes, unsigned char *out, unsigned expect)
{
    static PliShared S;
    std::memset(&S, 0x5A, sizeof S);          /* nothing may rely on zeroed shared memory */
    int32_t status = -1;
    PliStream st{ z, zbytes, out, expect, &status };
    pli_inflate(st, S);
    return status;
}

extern "C" void inflate_host_stats(unsigned long long *out8)
{
    for (int i = 0; i < 8; i++) { out8[i] = pli_stat[i]; pli_stat[i] = 0; }
}
