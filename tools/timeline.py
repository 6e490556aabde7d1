"""timeline.py TRACE.csv [first_dispatch count] -- what a rocprofv3 --kernel-trace of the segment engine looks like in TIME: per kernel the average duration and the
average gap to the previous kernel of its queue, how much of the engine's span at least one / two kernels were running, and a window of dispatches
(start offset, duration, queue) from the steady state.  Used for batches in launch groups (two queues side by side)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "seg_k_" not in n or "resolve" in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("seg_k_")[1].split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("dispatches %d, span %.2f ms" % (len(rows), (t1 - t0) / 1e6))
by = {}
lastq = {}
for s, e, n, q in rows:
    d = by.setdefault(n, [0, 0, 0, 0]); d[0] += 1; d[1] += e - s
    if q in lastq: d[2] += s - lastq[q]; d[3] += 1
    lastq[q] = e
for n, d in by.items(): print("%-40s calls %6d avg %8.2f us   gap to the queue's previous kernel %6.2f us" % (n, d[0], d[1] / d[0] / 1e3, d[2] / max(1, d[3]) / 1e3))
ev = []
for s, e, n, q in rows: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl = 0; last = t0; tot = {}
for t, dlt in ev:
    tot[lvl] = tot.get(lvl, 0) + t - last; last = t; lvl += dlt
print("time with k kernels running: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / (t1 - t0)) for k, v in sorted(tot.items())))
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
c = int(sys.argv[3]) if len(sys.argv) > 3 else 24
base = rows[a][0]
for s, e, n, q in rows[a:a + c]: print("  +%8.2f us  %7.2f us  queue %s  %s" % ((s - base) / 1e3, (e - s) / 1e3, q, n))
