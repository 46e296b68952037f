#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as a per-kernel
table: calls, total / average / min / max duration.  Used to turn the scratch
output under gpurun_out/ into the small text summaries committed in profiles/."""
import glob
import sqlite3
import sys


def main(path):
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    for p in dbs:
        db = sqlite3.connect(p)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        cols = [r[1] for r in db.execute("pragma table_info('%s')" % ks)]
        namecol = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else cols[-1])
        rows = db.execute(
            "select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
            "max(d.private_segment_size), max(d.group_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
            "from '%s' d join '%s' s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol)).fetchall()
        total = sum(r[2] for r in rows) or 1
        print("# %s" % p)
        print("%-60s %6s %14s %14s %12s %12s %6s %8s %8s %6s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "scratch", "lds", "grid", "wg"))
        for r in rows[:25]:
            print("%-60s %6d %14d %14.0f %12d %12d %6.2f %8d %8d %6d %6d" % (str(r[0])[:60], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9]))


if __name__ == "__main__":
    main(sys.argv[1])
