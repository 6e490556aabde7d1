"""Dump the judged parts of rocprofv3's rocpd SQLite output (this image's rocprofv3 writes *_results.db) as text:
per-kernel stats (the --stats summary) and, when present, PMC counter values per dispatch.
usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    print(f"## {path}")
    print("# kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total_us, average_us, percent")
    for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-90s %6d %16.1f %16.1f %8.4f" % (r[0][:90], r[1], r[2], r[3], r[4]))
    try:
        rows = list(con.execute("select kernel_name,counter_name,value,grid_size,workgroup_size,(end-start) from counters_collection"))
    except sqlite3.OperationalError:
        rows = []
    if rows:
        print("# PMC per dispatch: kernel, counter, value, grid, workgroup, duration_ns(end-start)")
        for r in rows:
            print("%-90s %-12s %14.3f %9d %5d %14d" % (r[0][:90], r[1], r[2], r[3], r[4], r[5]))
    print()
