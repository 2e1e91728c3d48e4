#!/usr/bin/env python3
"""HBM traffic per dispatch of selected kernels from two rocprofv3 PMC passes (FETCH_SIZE,
WRITE_SIZE; separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
    python tools/pmc_traffic.py <fetch dir> <write dir> <kernel substring> [...]
gfx950: FETCH_SIZE counts half the bytes of wide (16 B / lane) streaming reads -> x2; both in KiB."""
import glob
import sqlite3
import sys


def rows(path, counter, like):
  db = sqlite3.connect(sorted(glob.glob(path + "/**/*.db", recursive=True))[-1])
  return db.execute(
      "select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=?"
      " and kernel_name like ? group by kernel_name order by count(*) desc", (counter, f"%{like}%")).fetchall()


for like in sys.argv[3:]:
  fe, wr = rows(sys.argv[1], "FETCH_SIZE", like), rows(sys.argv[2], "WRITE_SIZE", like)
  for (name, n, fv, dur), w in zip(fe, wr):
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    rd, wb = 2 * fv * 1024, w[2] * 1024
    print(f"{short}\n  dispatches {n}  mean duration {dur / 1e3:.1f} us  read {rd / 1e6:.2f} MB  write {wb / 1e6:.2f} MB"
          f"  total {(rd + wb) / 1e6:.2f} MB  -> {(rd + wb) / dur:.1f} GB/s")
