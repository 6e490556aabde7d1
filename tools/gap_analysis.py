"""gap_analysis.py TRACE.csv -- for every control kernel of a batch on the segment engine: the gap to its queue's previous kernel, and what the OTHER queues were running when that
kernel ended (which kernel, and whether the control kernel's start coincides with that other kernel's end): is the gap a resource the other launch groups hold?"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "seg_k_" not in n or "resolve" in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("seg_k_")[1].split("<")[0].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
lastq = {}
byq = collections.defaultdict(list)
for s, e, n, q in rows: byq[q].append((s, e, n))
hist = collections.Counter(); coinc = collections.Counter(); running_kind = collections.Counter(); tot = 0
for s, e, n, q in rows:
    if n == "ctl" and q in lastq:
        gap = s - lastq[q]
        b = "<1us" if gap < 1000 else ("1-4us" if gap < 4000 else ("4-10us" if gap < 10000 else ("10-20us" if gap < 20000 else ">20us")))
        hist[b] += 1; tot += 1
        if gap >= 4000:
            # kernels of other queues running at the end of the previous kernel of this queue
            t0 = lastq[q]
            others = [(s2, e2, n2) for q2 in byq if q2 != q for (s2, e2, n2) in byq[q2] if s2 <= t0 < e2]
            running_kind[tuple(sorted(o[2] for o in others))] += 1
            ends = [e2 for (s2, e2, n2) in others if abs(e2 - s) < 1500]
            coinc["starts within 1.5 us of another queue's kernel END" if ends else "no"] += 1
    lastq[q] = e
print("control kernels", tot, "gap histogram", dict(hist))
print("for gaps >= 4 us: what the other queues ran when the previous kernel ended:", running_kind.most_common(6))
print("for gaps >= 4 us:", dict(coinc))
