#!/usr/bin/env python3
"""Developer helper: print per-kernel average durations from a rocprofv3 --kernel-trace --stats output directory.
    python tools/kstats.py gpurun_out/<dir> [substring ...]"""
import csv, glob, sys
d = sys.argv[1]
subs = sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if subs and not any(s in n for s in subs):
            continue
        print("%8.1f us avg %6s calls  min %8.1f  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["MinNs"]) / 1e3, n[:110]))
