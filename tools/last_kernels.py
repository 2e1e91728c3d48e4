"""The last N kernels of a rocprofv3 --kernel-trace database (start / duration / gap, microseconds).
usage: python tools/last_kernels.py <dir> [N=12]"""
import glob, sqlite3, sys
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
for name, s, e in rows:
  short = name.replace("mi355q::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]
  print(f"start {(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {short}")
  prev_end = e
