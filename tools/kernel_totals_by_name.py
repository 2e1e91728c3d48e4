"""Per-kernel totals of the LAST mi355q_gptq_hinv_f64 call in a rocprofv3 --kernel-trace database.
usage: python tools/kernel_totals_by_name.py <dir>"""
import glob, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "diag_sum" in r[0])
rows = rows[first:]
tot = defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
  k = name.replace("mi355q::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
  tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
print(f"wall {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
  print(f"{t:10.1f} us  {n:5d} x  {k}")
