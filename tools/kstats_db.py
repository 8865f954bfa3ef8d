#!/usr/bin/env python3
"""Developer helper: per-kernel call counts and average / min durations from a rocprofv3 results database (rocpd sqlite,
what `rocprofv3 --kernel-trace` writes by default on ROCm 7.2).   python tools/kstats_db.py <results.db> [substring ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
namecol = "display_name" if "display_name" in cols else "kernel_name"
q = f"select s.{namecol}, count(*), avg(d.end - d.start), min(d.end - d.start), sum(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id group by s.{namecol} order by 5 desc"
tot = 0
rows = list(cur.execute(q))
tot = sum(r[4] for r in rows)
for name, n, avg, mn, sm in rows:
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print("%5.1f%% %9.1f us avg %6d calls  min %9.1f  %s" % (100.0 * sm / tot, avg / 1e3, n, mn / 1e3, name[:120]))
print("total kernel time %.2f ms" % (tot / 1e6))
