#!/usr/bin/env python3
"""Developer helper: per training step (delimited by the loss kernel) of a rocprofv3 --kernel-trace CSV: wall span, sum of
kernel durations, idle time between kernels.   python tools/trace_steps.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "ce_fwd_grad_k" in r[2]]
for a, b in zip(marks, marks[1:]):
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0] if len(seg) > 1 else 0
    span = rows[b][0] - rows[a][0]
    tot = sum(e - s for s, e, _ in seg)
    idle, cur_e, n_gap = 0, seg[0][1], 0
    for s, e, _ in seg[1:]:
        if s > cur_e:
            idle += s - cur_e
            n_gap += 1
        cur_e = max(cur_e, e)
    print("step: %4d kernels  span %.3f ms  kernel sum %.3f ms  idle %.3f ms in %d gaps (%.2f us each)" %
          (len(seg), span / 1e6, tot / 1e6, idle / 1e6, n_gap, idle / 1e3 / max(n_gap, 1)))
