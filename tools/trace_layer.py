#!/usr/bin/env python3
"""Developer helper: the kernels of ONE layer's forward and ONE layer's backward inside a replayed step, from a rocprofv3
--kernel-trace CSV, with start / end relative to the window and the queue each ran on (two branch streams -> two queues).
    python tools/trace_layer.py DIR [layer index in the last full step, default 12]"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Workgroup_Size", ""), r.get("Grid_Size", "")))
rows.sort()
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
marks = [i for i, r in enumerate(rows) if "ce_fwd_grad_k" in r[2]]
# the shortest CE ... CE span = a graph-replayed step (the bench's eager probe regions follow it and are longer): backward of
# step k, optimizer, forward of step k+1
a, b = min(zip(marks, marks[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
seg = rows[a:b]
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("mm::", ""))[:64]


def window(first_pat, nth, count_pat):
    idx = [i for i, r in enumerate(seg) if first_pat in r[2]]
    i0 = idx[nth]
    i1 = idx[nth + 1] if nth + 1 < len(idx) else len(seg)
    return seg[i0:i1]


for title, pat in (("backward of one layer (from its adapter-combine backward)", "bottleneck_dma_k<mm::BF16, 1024, 128, 1"),
                   ("forward of one layer (from its dual LayerNorm)", "layernorm_fwd_k<mm::BF16, 2, true>")):
    idx = [i for i, r in enumerate(seg) if pat in r[2]]
    step = 2 if "bottleneck" in pat else 1     # two adapter backward kernels per layer
    if len(idx) < (L + 1) * step + 1:
        continue
    w = seg[idx[L * step]:idx[(L + 1) * step]]
    t0 = w[0][0]
    print("== %s: %d kernels, %.1f us wall, %.1f us kernel sum" % (title, len(w), (max(r[1] for r in w) - t0) / 1e3, sum(r[1] - r[0] for r in w) / 1e3))
    for s, e, n, q, wg, grid in w:
        print("  %8.1f -> %8.1f  (%6.1f us)  q%-3s grid %-8s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, grid, short(n)))
