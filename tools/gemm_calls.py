#!/usr/bin/env python3
"""GEMM dispatches of the last mi355q_gptq_hinv_f64 call in a rocprofv3 kernel trace, grouped by
(kernel, grid): calls, total and mean duration.   python tools/gemm_calls.py <trace dir>"""
import glob
import sqlite3
import sys

db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, grid_x, grid_y, grid_z, duration, start from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "copy_damped_lower" in r[0])
rows = rows[first:]
groups = {}
for name, gx, gy, gz, dur, _ in rows:
  short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("mi355q::", "")
  short = short.split("(")[0][:60]
  k = (short, gx // 256, gy, gz)
  g = groups.setdefault(k, [0, 0])
  g[0] += 1
  g[1] += dur
total = sum(r[4] for r in rows)
print(f"# one hinv call: {total / 1e3:.1f} us of kernels; blocks = grid / 256 threads")
print(f"{'calls':>6} {'total_us':>10} {'mean_us':>9} {'pct':>5}  kernel  blocks(x,y,z)")
for (short, bx, gy, gz), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
  print(f"{n:6d} {t / 1e3:10.1f} {t / n / 1e3:9.1f} {100 * t / total:5.1f}  {short}  ({bx},{gy},{gz})")
