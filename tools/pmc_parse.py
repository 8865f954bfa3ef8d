#!/usr/bin/env python3
"""Developer helper: print per-dispatch counters of one kernel from a rocprofv3 --pmc csv directory."""
import collections, csv, glob, sys
d = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            e = d[(f, r["Dispatch_Id"])]
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(d.items()):
    print(k[1], {a: (b if a == "_ns" else "%.3g" % b) for a, b in sorted(v.items())})
