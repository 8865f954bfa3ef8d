#!/usr/bin/env python3
"""Per-kernel averages per launch of one rocprofv3 --pmc counter.
    python tools/pmc_by_kernel.py gpurun_out/<pmc dir> COUNTER > profiles/rNN_x_pmc_<counter>_by_kernel.csv"""
import collections, csv, glob, sys
d, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Launches", "Avg_" + counter + "_per_launch"])
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, n, "%.1f" % (s / n)])
