#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace output directory (rocpd sqlite).
    python tools/kernel_totals.py <dir> [substring ...]"""
import glob
import sqlite3
import sys

db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
want = sys.argv[2:]
print(f"{'calls':>6} {'total_us':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel")
for name, n, tot, avg, lo, hi in rows:
  if want and not any(w in name for w in want):
    continue
  short = name.replace("(anonymous namespace)::", "").replace("void ", "")
  print(f"{n:6d} {tot / 1e3:10.1f} {avg / 1e3:9.1f} {lo / 1e3:9.1f} {hi / 1e3:9.1f}  {short[:110]}")
