#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (or CSV dir) into the text summaries kept
under profiles/: per-kernel call count / total / average / min / max duration,
and, when PMC counters were collected, the per-dispatch counter means.

    python tools/rocprof_summary.py gpurun_out/prof1 > profiles/r01_xxx.txt
"""
import glob
import os
import sqlite3
import sys


def short(name: str, n: int = 110) -> str:
  name = name.replace("(anonymous namespace)::", "")
  return name if len(name) <= n else name[: n - 3] + "..."


def summarize_db(path: str) -> None:
  c = sqlite3.connect(path)
  print(f"# source: {os.path.basename(path)} (rocprofv3 rocpd sqlite)")
  rows = c.execute(
      "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
      "from kernels group by name order by sum(duration) desc").fetchall()
  total = sum(r[2] for r in rows) or 1
  print("# kernel-trace stats (durations in microseconds)")
  print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
  for name, calls, tot, avg, mn, mx in rows:
    print(f"{calls:6d} {tot/1e3:12.2f} {avg/1e3:10.3f} {mn/1e3:10.3f} {mx/1e3:10.3f} "
          f"{100*tot/total:6.2f}  {short(name)}")
  try:
    pmc = c.execute(
        "select k.name, p.counter_name, count(*), avg(p.value), sum(p.value) "
        "from counters_collection p join kernels k on k.dispatch_id = p.dispatch_id "
        "group by k.name, p.counter_name order by k.name").fetchall()
  except sqlite3.Error:
    pmc = []
  if pmc:
    print("# PMC counters (mean per dispatch)")
    for name, counter, n, avg, tot in pmc:
      print(f"{counter:>20} n={n:5d} mean={avg:18.3f} sum={tot:20.1f}  {short(name, 80)}")


def main():
  root = sys.argv[1]
  dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)) if os.path.isdir(root) else [root]
  for db in dbs:
    summarize_db(db)


if __name__ == "__main__":
  main()
