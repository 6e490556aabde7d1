#!/usr/bin/env python3
"""Generates pngloss_amd/csrc/pl_lead_asm.h: the hand-scheduled gfx950 inner loops of the band-leader chains
(pl_engine.hip, lead_fast_run).  One asm statement = a 4x unrolled loop over pixels; all rotating values (table
entries, records, addresses) live in fixed VGPRs v200.. so that the rotation costs no moves.

Per pixel step k (pixel i+k), with E[j] = table entry pairs, Q[j] = record quads, A[j] = looked-up addresses:
  critical  s_waitcnt lgkmcnt(3)          entry of pixel i+k-1 has arrived (3 younger LDS ops may be in flight)
            ...filter specific...          8*byte of the previous pixel -> prediction -> 8*osym -> table address
            ds_read_b64 E[k], A            THE lookup
  shadow    v_cmp / s_cbranch_vccnz        previous pixel reconstructs outside 0..255 -> leave (almost never)
            v_and_or + ds_add_u32          histogram bump of the previous pixel (table address HB | (8v & 0x7f8))
            v_sub_sdwa + v_lshl_or + ds_write_b64   result record of the previous pixel {8*byte | 8*v << 16, 8*diff + TB}
            ds_read_b128 Q[k+2]            record of pixel i+k+2
            ...                            8*lo of this pixel, pre-added address parts of the next
Usage: python tools/gen_lead_asm.py > pngloss_amd/csrc/pl_lead_asm.h
"""
import os
BURST = int(os.environ.get("PL_LEAD_BURST", "4"))
PREF = os.environ.get("PL_LEAD_PREF", "single")   # where the record prefetch sits: "start" of the step or in its "shadow"
ILV = int(os.environ.get("PL_LEAD_ILV", "0"))   # 1 (tried, 3 % slower: it delays the lookup): independent instructions between the dependent ones of the critical chain
XW = int(os.environ.get("PL_LEAD_XW", "0"))     # timing experiments that keep the results exact: repeat the record write / the record prefetch / add VALU moves
XR = int(os.environ.get("PL_LEAD_XR", "0"))
XV = int(os.environ.get("PL_LEAD_XV", "0"))
ABL = int(os.environ.get("PL_LEAD_ABLATE", "0"))   # timing experiments only (tools/lead_ablate.sh): >0 drops pieces, results become wrong
E = [(200, 201), (202, 203), (204, 205), (206, 207)]
A = [209, 213]            # looked-up addresses; the register below each one takes 8*byte of that pixel: {8*byte | 8*v << 16, address} is the
                          # result record as it is written (8*diff = address - 8*v - TB is left to the readers)
LO = 210
PRE = 211
T0, BADACC, OSYM, T1, T2, T3 = 214, 215, 216, 217, 218, 219
Q = [220, 224, 228, 232]       # quads (paeth: second quads at 240..)
Q2 = [240, 244, 248, 252]
RPTR, OPTR, ONE, VHB = 236, 237, 238, 239
SDWA_W0 = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
SDWA_W1 = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
SDWA_S1W0 = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"


def v(n):
    return f"v{n}"


def vr(lo, n):
    return f"v[{lo}:{lo + n - 1}]"


def step(mode, k):
    """mode: 'nu' (none/up), 'sub', 'avg', 'pae'"""
    ep = E[(k + 3) % 4]          # entry of the previous pixel
    e2 = E[(k + 2) % 4]          # entry of the pixel before it (its .y = thr for this pixel)
    en = E[k]
    an, ap = A[k % 2], A[(k + 1) % 2]
    BACK = ap - 1
    q, qprev, qnext, qpre = Q[k], Q[(k + 3) % 4], Q[(k + 1) % 4], Q[(k + 2) % 4]
    rw = 2 if mode == "pae" else 1
    L = []
    npre = 2 if mode == "pae" else 1
    pref = [f"ds_read_b128 {vr(qpre, 4)}, {v(RPTR)} offset:{(k + 2) * 64 * rw}"]
    if mode == "pae":
        pref.append(f"ds_read_b128 {vr(Q2[(k + 2) % 4], 4)}, {v(RPTR)} offset:{(k + 2) * 64 * rw + 16}")
    pre_add = {"nu": f"v_add_u32_e32 {v(PRE)}, {v(q)}, {v(e2[1])}", "sub": f"v_add_u32_e32 {v(PRE)}, {v(q + 1)}, {v(e2[1])}",
               "avg": f"v_add_u32_e32 {v(PRE)}, {v(q + 1)}, {v(e2[1])}", "pae": f"v_add_u32_e32 {v(PRE)}, {v(Q2[k] + 3)}, {v(e2[1])}"}[mode]
    if PREF == "start":
        # the record of pixel i+k+2 first: issued while the wave waits for the previous lookup anyway, and -- being older
        # than this step's lookup -- guaranteed complete behind the NEXT step's s_waitcnt, which is where it is first used
        L += pref
        L.append("s_waitcnt lgkmcnt(%d)" % (1 + npre))
    else:
        # record prefetch in the shadow (behind the lookup, so that it does not queue in front of it); the record of THIS pixel
        # (fetched two steps ago) is complete once at most lookup, write and prefetch of the previous step are outstanding
        if PREF == "single":
            # one wait: LDS operations complete in order, so the previous lookup's arrival implies this pixel's record (fetched
            # a step earlier than that lookup was issued)
            L.append("s_waitcnt lgkmcnt(%d)" % (1 + XW + npre * (1 + XR)))
            L.append(pre_add)
        else:
            L.append("s_waitcnt lgkmcnt(%d)" % (2 + npre))
            L.append(pre_add)
            L.append("s_waitcnt lgkmcnt(%d)" % (1 + npre))
    if mode == "nu":
        # record: x = 8*osym + 8*e0 + TB, y = 8*lo.  PRE = x + thr(i-2) was added in the previous step's shadow
        L.append(f"v_add_u32_sdwa {v(an)}, sext({v(ep[0])}), {v(PRE)} {SDWA_W1}")
        L.append(f"ds_read_b64 {vr(en[0], 2)}, {v(an)}")
        L.append(f"v_sub_u32_sdwa {v(BACK)}, sext({v(ep[0])}), {v(qprev + 1)} {SDWA_W0}")
    elif ILV and mode in ("sub", "avg", "pae"):
        # A VALU instruction that needs the result of the one in front of it issues ~4 cycles later than an independent one
        # (profiles/r02_lead_ablation.txt section 8): the record word, the validity OR and the rem + PRE part of the address fill
        # those slots instead of waiting in the shadow.
        orr = [f"v_or_b32_e32 {v(BADACC)}, {v(BADACC)}, {v(BACK)}"] if ABL < 3 else []
        pack = f"v_lshl_or_b32 {v(BACK)}, {v(ep[0])}, 16, {v(BACK)}"
        L.append(f"v_sub_u32_sdwa {v(BACK)}, sext({v(ep[0])}), {v(LO)} {SDWA_W0}")
        L.append(f"v_add_u32_sdwa {v(T3)}, sext({v(ep[0])}), {v(PRE)} {SDWA_W1}")
        if mode == "sub":
            L.append(f"v_sub_u32_e32 {v(T0)}, {v(q)}, {v(BACK)}")
            L += orr
            L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 11")
            L.append(pack)
            L.append(f"v_add_u32_e32 {v(an)}, {v(T3)}, {v(OSYM)}")
        elif mode == "avg":
            L.append(f"v_add_u32_e32 {v(T0)}, {v(BACK)}, {v(q + 2)}")
            L += orr
            L.append(f"v_bfe_u32 {v(T0)}, {v(T0)}, 4, 8")
            L.append(pack)
            L.append(f"v_sub_u32_e32 {v(T0)}, {v(q)}, {v(T0)}")
            L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 8")
            L.append(f"v_lshl_add_u32 {v(an)}, {v(OSYM)}, 3, {v(T3)}")
        else:
            q2 = Q2[k]
            L.append(f"v_sad_u32 {v(T0)}, {v(BACK)}, {v(q)}, 0")
            L.append(f"v_add_u32_e32 {v(T1)}, {v(BACK)}, {v(q + 1)}")
            L.append(f"v_sub_u32_e32 {v(T2)}, {v(q + 3)}, {v(BACK)}")
            L.append(f"v_sad_u32 {v(T1)}, {v(T1)}, {v(q + 2)}, 0")
            L.append(f"v_lshl_or_b32 {v(T0)}, {v(T0)}, 14, {v(q2)}")
            L += orr
            L.append(f"v_lshl_or_b32 {v(T1)}, {v(T1)}, 14, {v(q2 + 1)}")
            L.append(pack)
            L.append(f"v_min3_u32 {v(T0)}, {v(T2)}, {v(T0)}, {v(T1)}")
            L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 11")
            L.append(f"v_add_u32_e32 {v(an)}, {v(T3)}, {v(OSYM)}")
        L.append(f"ds_read_b64 {vr(en[0], 2)}, {v(an)}")
    elif mode == "sub":
        # record: x = 8*orig, y = 8*e0 + TB
        L.append(f"v_sub_u32_sdwa {v(BACK)}, sext({v(ep[0])}), {v(LO)} {SDWA_W0}")
        L.append(f"v_sub_u32_e32 {v(T0)}, {v(q)}, {v(BACK)}")
        L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 11")
        L.append(f"v_add_u32_sdwa {v(T0)}, sext({v(ep[0])}), {v(PRE)} {SDWA_W1}")
        L.append(f"v_add_u32_e32 {v(an)}, {v(T0)}, {v(OSYM)}")
        L.append(f"ds_read_b64 {vr(en[0], 2)}, {v(an)}")
    elif mode == "avg":
        # record: x = orig, y = 8*e0 + TB, z = 8*above, w = -8*orig
        L.append(f"v_sub_u32_sdwa {v(BACK)}, sext({v(ep[0])}), {v(LO)} {SDWA_W0}")
        L.append(f"v_add_u32_e32 {v(T0)}, {v(BACK)}, {v(q + 2)}")
        L.append(f"v_bfe_u32 {v(T0)}, {v(T0)}, 4, 8")
        L.append(f"v_sub_u32_e32 {v(T0)}, {v(q)}, {v(T0)}")
        L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 8")
        L.append(f"v_add_u32_sdwa {v(T0)}, sext({v(ep[0])}), {v(PRE)} {SDWA_W1}")
        L.append(f"v_lshl_add_u32 {v(an)}, {v(OSYM)}, 3, {v(T0)}")
        L.append(f"ds_read_b64 {vr(en[0], 2)}, {v(an)}")
    else:
        # records: q  = { (8*diag << 14) + BIAS, (8*(2*diag-above) << 14) + BIAS, -, (8*|above-diag| << 14) + 8*orig + 2048 }
        #          q2 = { (1<<12) + 8*(orig-above) + 2048, (2<<12) + 8*(orig-diag) + 2048, 8*orig, 8*e0 + TB }
        q2 = Q2[k]
        L.append(f"v_sub_u32_sdwa {v(BACK)}, sext({v(ep[0])}), {v(LO)} {SDWA_W0}")
        L.append(f"v_lshl_add_u32 {v(T3)}, {v(BACK)}, 14, %[bias]")      # (8*left << 14) + BIAS
        L.append(f"v_sub_u32_e32 {v(T2)}, {v(q + 3)}, {v(BACK)}")         # key of 'left'
        L.append(f"v_sad_u32 {v(T0)}, {v(T3)}, {v(q)}, {v(q2)}")          # key of 'above':      |left - diag| << 14 | ...
        L.append(f"v_sad_u32 {v(T1)}, {v(T3)}, {v(q + 1)}, {v(q2 + 1)}")  # key of 'upper left': |left + above - 2 diag| << 14 | ...
        L.append(f"v_min3_u32 {v(T0)}, {v(T2)}, {v(T0)}, {v(T1)}")
        L.append(f"v_bfe_i32 {v(OSYM)}, {v(T0)}, 0, 11")
        L.append(f"v_add_u32_sdwa {v(T0)}, sext({v(ep[0])}), {v(PRE)} {SDWA_W1}")
        L.append(f"v_add_u32_e32 {v(an)}, {v(T0)}, {v(OSYM)}")
        L.append(f"ds_read_b64 {vr(en[0], 2)}, {v(an)}")
    # ---- shadow ----
    if not (ILV and mode in ("sub", "avg", "pae")):
        if ABL < 3:
            L.append(f"v_or_b32_e32 {v(BADACC)}, {v(BADACC)}, {v(BACK)}")
        # the record's first word also carries 8*v in its upper half: (8v >> 3) & 255 is the histogram bin the deferred bump goes to
        L.append(f"v_lshl_or_b32 {v(BACK)}, {v(ep[0])}, 16, {v(BACK)}")
    L.append(f"ds_write_b64 {v(OPTR)}, {vr(BACK, 2)} offset:{32 * k}")
    for _ in range(XW):
        L.append(f"ds_write_b64 {v(OPTR)}, {vr(BACK, 2)} offset:{32 * k}")
    for _ in range(XV):
        L.append(f"v_mov_b32_e32 {v(T3)}, {v(BADACC)}")
    if PREF != "start":
        L += pref
        for _ in range(XR):
            L += pref
    # thr of the next pixel = entry of the previous pixel .y ; pre-add it to the next record's address part
    if mode == "nu":
        if PREF == "start":
            L.append(f"v_add_u32_e32 {v(PRE)}, {v(qnext)}, {v(ep[1])}")
    elif mode == "sub":
        L.append(f"v_sub_u32_e32 {v(LO)}, {v(OSYM)}, {v(q)}")
        if PREF == "start":
            L.append(f"v_add_u32_e32 {v(PRE)}, {v(qnext + 1)}, {v(ep[1])}")
    elif mode == "avg":
        L.append(f"v_lshl_add_u32 {v(LO)}, {v(OSYM)}, 3, {v(q + 3)}")
        if PREF == "start":
            L.append(f"v_add_u32_e32 {v(PRE)}, {v(qnext + 1)}, {v(ep[1])}")
    else:
        L.append(f"v_sub_u32_e32 {v(LO)}, {v(OSYM)}, {v(Q2[k] + 2)}")
        if PREF == "start":
            L.append(f"v_add_u32_e32 {v(PRE)}, {v(Q2[(k + 1) % 4] + 3)}, {v(ep[1])}")
    return L


def gen(mode):
    rw = 2 if mode == "pae" else 1
    body = []
    # outer loop: bursts of BURST iterations (4 pixels each); the validity test -- any byte so far outside 0..255 -- sits
    # between the bursts: a scalar branch on a VALU-written condition costs ~190 cycles on gfx950 even when the v_cmp is 40
    # instructions old (profiles/r02_lead_ablation.txt), so it is paid once per 16 pixels, not per pixel or per iteration
    body.append("2:")
    # (the first burst of a run may be shorter -- slow pixels cluster, and everything a burst does behind a bad byte is wasted --
    # and the burst length doubles from there up to BURST)
    body.append("s_min_u32 %[inner], %[cnt], %[burst]")
    body.append("s_sub_u32 %[cnt], %[cnt], %[inner]")
    body.append("s_lshl_b32 %[burst], %[burst], 1")
    body.append("s_min_u32 %[burst], %[burst], " + str(BURST))
    body.append("1:")
    for k in range(4):
        body += step(mode, k)
    body.append(f"v_add_u32_e32 {v(RPTR)}, {hex(256 * rw)}, {v(RPTR)}")
    body.append(f"v_add_u32_e32 {v(OPTR)}, 0x80, {v(OPTR)}")
    body.append("s_sub_u32 %[inner], %[inner], 1")
    body.append("s_cmp_lg_u32 %[inner], 0")
    body.append("s_cbranch_scc1 1b")
    if ABL < 2:
        body.append(f"v_cmp_lt_u32_e32 vcc, %[c2047], {v(BADACC)}")
    if ABL < 1:
        body.append("s_cbranch_vccnz 99f")
    body.append("s_cmp_lg_u32 %[cnt], 0")
    body.append("s_cbranch_scc1 2b")
    body.append("99:")
    body.append("s_waitcnt lgkmcnt(0)")
    return body


def emit(mode, name):
    rw = 2 if mode == "pae" else 1
    out = []
    out.append(f"/* {name}: see tools/gen_lead_asm.py */")
    args = "LeadState &st, const uint32_t rptr, const uint32_t optr, int &iters, int burst, u32x4 &qa, u32x4 &qb"
    if rw == 2:
        args += ", u32x4 &qx, u32x4 &qy"
    out.append(f"__device__ __forceinline__ uint32_t {name}({args})")
    out.append("{")
    out.append("    uint32_t bad = 0; int inner;")
    out.append("    uint32_t e0 = st.e0; int h1 = st.h1, h2 = st.h2, lo8 = st.lo8, addr = st.addr, cnt = iters;")
    pre_in = {"nu": Q[0], "sub": Q[0] + 1, "avg": Q[0] + 1, "pae": Q2[0] + 3}[mode]
    lines = []
    # move inputs into the fixed registers
    mv = [(BADACC, "0"), (E[3][0], "%[e0]"), (E[3][1], "%[h1]"), (E[2][1], "%[h2]"), (A[1], "%[addr]"), (RPTR, "%[rptr]"), (OPTR, "%[optr]")]
    for dst, src in mv:
        lines.append(f"v_mov_b32_e32 {v(dst)}, {src}")
    for j in range(4):
        lines.append(f"v_mov_b32_e32 {v(Q[0] + j)}, %[qa{j}]")
        lines.append(f"v_mov_b32_e32 {v(Q[1] + j)}, %[qb{j}]")
        if rw == 2:
            lines.append(f"v_mov_b32_e32 {v(Q2[0] + j)}, %[qc{j}]")
            lines.append(f"v_mov_b32_e32 {v(Q2[1] + j)}, %[qd{j}]")
    if mode == "nu":
        lines.append(f"v_mov_b32_e32 {v(Q[3] + 1)}, %[lo8]")        # 8*lo of the previous pixel sits in its record
    else:
        lines.append(f"v_mov_b32_e32 {v(LO)}, %[lo8]")
    lines.append(f"v_add_u32_e32 {v(PRE)}, {v(pre_in)}, {v(E[2][1])}")
    lines += gen(mode)
    # outputs
    lines.append(f"v_mov_b32_e32 %[bad], {v(BADACC)}")
    lines.append(f"v_mov_b32_e32 %[e0], {v(E[3][0])}")
    lines.append(f"v_mov_b32_e32 %[h1], {v(E[3][1])}")
    lines.append(f"v_mov_b32_e32 %[h2], {v(E[2][1])}")
    lines.append(f"v_mov_b32_e32 %[addr], {v(A[1])}")
    lines.append(f"v_mov_b32_e32 %[lo8], {v(Q[3] + 1 if mode == 'nu' else LO)}")
    for j in range(4):
        lines.append(f"v_mov_b32_e32 %[qa{j}], {v(Q[0] + j)}")
        lines.append(f"v_mov_b32_e32 %[qb{j}], {v(Q[1] + j)}")
        if rw == 2:
            lines.append(f"v_mov_b32_e32 %[qc{j}], {v(Q2[0] + j)}")
            lines.append(f"v_mov_b32_e32 %[qd{j}], {v(Q2[1] + j)}")
    out.append("    uint32_t " + ", ".join([f"qa{j} = qa[{j}], qb{j} = qb[{j}]" for j in range(4)]) + ";")
    if rw == 2:
        out.append("    uint32_t " + ", ".join([f"qc{j} = qx[{j}], qd{j} = qy[{j}]" for j in range(4)]) + ";")
    out.append("    asm volatile(")
    for ln in lines:
        out.append(f'        "{ln}\\n"')
    outs = ['[bad] "=&v"(bad)', '[cnt] "+s"(cnt)', '[inner] "=&s"(inner)', '[burst] "+s"(burst)', '[e0] "+v"(e0)', '[h1] "+v"(h1)', '[h2] "+v"(h2)', '[addr] "+v"(addr)', '[lo8] "+v"(lo8)']
    for j in range(4):
        outs += [f'[qa{j}] "+v"(qa{j})', f'[qb{j}] "+v"(qb{j})']
        if rw == 2:
            outs += [f'[qc{j}] "+v"(qc{j})', f'[qd{j}] "+v"(qd{j})']
    ins = ['[rptr] "v"(rptr)', '[optr] "v"(optr)', '[c2047] "s"(2047u)', '[bias] "s"(1u << 27)']
    hi = 256 if rw == 2 else 240
    clob = ", ".join([f'"v{n}"' for n in range(200, hi)] + ['"vcc"', '"scc"', '"memory"'])
    out.append("        : " + ", ".join(outs))
    out.append("        : " + ", ".join(ins))
    out.append("        : " + clob + ");")
    out.append("    st.e0 = e0; st.h1 = h1; st.h2 = h2; st.lo8 = lo8; st.addr = addr;")
    out.append("    " + " ".join([f"qa[{j}] = qa{j}; qb[{j}] = qb{j};" for j in range(4)]))
    if rw == 2:
        out.append("    " + " ".join([f"qx[{j}] = qc{j}; qy[{j}] = qd{j};" for j in range(4)]))
    out.append("    iters = cnt + inner;")
    out.append("    return bad;")
    out.append("}")
    return "\n".join(out)


print("/* GENERATED by tools/gen_lead_asm.py -- do not edit.  Hand-scheduled gfx950 inner loops of the band-leader chains. */")
print("#ifndef PL_LEAD_ASM_H\n#define PL_LEAD_ASM_H\n")
for m, n in [("nu", "lead_asm_noneup"), ("sub", "lead_asm_sub"), ("avg", "lead_asm_avg"), ("pae", "lead_asm_paeth")]:
    print(emit(m, n))
    print()
print("#endif")
