"""kstats.py DIR [DIR ...] -- average duration (us) of the segment engine's five kernels from rocprofv3 --kernel-trace --stats output directories."""
import csv, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    row = {}
    for r in csv.reader(open(f)):
        if len(r) > 4 and "seg_k_" in r[0] and "resolve" not in r[0]:
            row[r[0].split("seg_k_")[1].split("(")[0].split("<")[0]] = float(r[3]) / 1e3
    print(d, " ".join("%s %.2f" % (k, row.get(k, 0)) for k in ("ctl", "enum", "chain", "replay", "post")), "sum %.1f" % sum(row.values()))
