#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV output (…_counter_collection.csv) per kernel and counter:
sum over dispatches, and value per dispatch.  Usage: pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict


def main(path, want=""):
    files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(float)
    disp = defaultdict(set)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                if want and want not in k:
                    continue
                c = row.get("Counter_Name", "?")
                acc[(k, c)] += float(row.get("Counter_Value", 0) or 0)
                disp[(k, c)].add(row.get("Dispatch_Id", "0"))
    print("%-40s %-28s %8s %20s %20s" % ("kernel", "counter", "launches", "sum", "per_launch"))
    for (k, c) in sorted(acc):
        n = max(1, len(disp[(k, c)]))
        print("%-40s %-28s %8d %20.0f %20.1f" % (k[:40], c, n, acc[(k, c)], acc[(k, c)] / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
