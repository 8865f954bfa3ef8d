#!/usr/bin/env python3
"""Developer helper: from a rocprofv3 --kernel-trace CSV, how much kernels overlap in time (sum of durations vs union), per
phase of the run, and the gaps between consecutive kernels.   python tools/trace_overlap.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
print("kernels", len(rows), "queues", sorted({r[3] for r in rows}), "streams", sorted({r[4] for r in rows}))
n = len(rows)
for lo, hi, label in ((0, n // 2, "first half (warm-up + capture)"), (n // 2, n, "second half")):
    seg = rows[lo:hi]
    tot = sum(e - s for s, e, *_ in seg)
    union, cur_s, cur_e = 0, None, None
    overl = 0
    for s, e, *_ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            overl += 1
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = seg[-1][1] - seg[0][0]
    print("%s: %d kernels, sum %.2f ms, union %.2f ms, span %.2f ms, %d start before the previous ended" % (label, len(seg), tot / 1e6, union / 1e6, span / 1e6, overl))
