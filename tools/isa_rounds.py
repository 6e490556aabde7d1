#!/usr/bin/env python3
"""isa_rounds.py -- where a kernel of the segment engine waits for device memory.

Compiles pngloss_amd/csrc/pl_seg.hip to gfx950 assembly (device only, with line info) and lists, per kernel, every "round":
a group of vector loads closed by an `s_waitcnt vmcnt(0)`.  A round of ONE load in straight-line code is a whole round trip
(0.25-0.45 us on this machine, tools/ubench_boundary.hip) spent on one value -- the pattern DESIGN.md section 5 describes
(`if (a || b)` on two device-memory fields, a load between two stores to shared memory, a load inside a branch, a kernel
parameter re-read behind a store).  Also prints the instruction mix (flat / ds / global / scalar loads) and the stack size.

usage: tools/isa_rounds.py [kernel-name-to-detail]        e.g.  tools/isa_rounds.py seg_k_chain
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pngloss_amd", "csrc", "pl_seg.hip")
KERNELS = ["seg_k_ctl", "seg_k_enum", "seg_k_chain", "seg_k_replay"]   # (the validation body rides in seg_k_ctl since round 4)


def main():
    asm = os.path.join(tempfile.mkdtemp(prefix="isa_rounds_"), "pl_seg.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-g", "-o", asm, SRC],
                   check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(SRC))
    src = open(asm).read().split("\n")
    detail = sys.argv[1] if len(sys.argv) > 1 else None
    todo = []
    for name in KERNELS:
        for i, l in enumerate(src):
            m = re.match(r"^_ZN12_GLOBAL__N_1[0-9]+" + name + r"(ILi(\d+)E)?E\S*:", l)
            if m:
                todo.append((name + ("<%s>" % m.group(2) if m.group(2) else ""), i))
    for name, a in todo:
        b = [i for i, l in enumerate(src) if i > a and ".amdhsa_kernel" in l][0]
        body = src[a:b]
        meta = "\n".join(src[b:b + 80])
        stack = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
        count = lambda pat: sum(1 for l in body if re.search(pat, l))
        rounds, cur, loc = [], [], "?"
        for l in body:
            m = re.search(r"\.loc\s+\d+\s+\d+\s+\d+.*; (\S+):(\d+)", l)
            if m:
                loc = os.path.basename(m.group(1)) + ":" + m.group(2)
            if re.search(r"\b(global_load|flat_load|scratch_load)", l):
                cur.append(loc)
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if m and cur and int(m.group(1)) == 0:
                rounds.append((len(cur), sorted(set(cur)), loc))
                cur = []
        print("%-13s %5d instructions, stack %s B; loads: %3d vector, %3d scalar, %2d flat; ds_read %3d ds_write %3d; %2d load->wait rounds, %2d of them a single load"
              % (name, sum(1 for l in body if re.match(r"^\t[a-z]", l)), stack.group(1) if stack else "?", count(r"\bglobal_load"), count(r"\bs_load"),
                 count(r"\bflat_(load|store|atomic)"), count(r"\bds_read"), count(r"\bds_write"), len(rounds), sum(1 for r in rounds if r[0] == 1)))
        if detail and name.startswith(detail):
            for n, where, at in rounds:
                print("    %2d load(s) from %s -> wait at %s" % (n, ", ".join(where[:6]), at))


if __name__ == "__main__":
    main()
