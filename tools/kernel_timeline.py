import glob, sqlite3, sys
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "copy_damped_lower" in r[0])
rows = rows[first:first + 40]
t0 = rows[0][1]
prev_end = t0
for name, s, e in rows:
  short = name.replace("mi355q::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:50]
  print(f"start {(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {short}")
  prev_end = e
