#!/usr/bin/env python3
"""Developer helper: per-kernel sums of every counter in a rocprofv3 --pmc output directory, as ratios to SQ_WAVE_CYCLES or,
with GRBM_GUI_ACTIVE present, to SIMD-cycles (1024 * GRBM_GUI_ACTIVE / 8).   python tools/pmc_ratio.py DIR [substring ...]"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    if sys.argv[2:] and not any(s in k for s in sys.argv[2:]):
        continue
    if c.get("GRBM_GUI_ACTIVE", 0) > 0:
        den, what = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0, "SIMD-cycles"
    else:
        den, what = c.get("SQ_WAVE_CYCLES", 1.0), "wave-cycles"
    print(k.replace("void ", "")[:60], "(per %s)" % what, " ".join("%s=%.3f" % (n.replace("SQ_", ""), v / den) for n, v in sorted(c.items())))
