#!/usr/bin/env python
"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): our kernels and the large copies in start order."""
import csv
import glob
import sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("lh_"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 2000000:
            ev.append((s, e, "copy " + r.get("Direction", r.get("Name", "?"))[:24]))
ev.sort()
t0 = ev[0][0] if ev else 0
tail = ev[-60:]
for s, e, n in tail:
    print("%10.3f ms  +%9.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
