"""Kernel totals of the LAST mi355q_gptq_apply_f32 call in a rocprofv3 --kernel-trace database of tools/apply_profile.py.
usage: python tools/apply_kernel_totals.py <dir>"""
import glob, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "gptq_rows_kernel" in r[0] or "gptq_block_kernel" in r[0]]
# the last call: walk back from the last chain kernel to the copy / plane kernels in front of its first group
last = starts[-1]
first = last
while first > 0 and ("gptq_" in rows[first - 1][0] or "upd_" in rows[first - 1][0] or "gemm" in rows[first - 1][0]):
  first -= 1
rows = rows[first:last + 1]
tot = defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
  k = n.replace("mi355q::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
  tot[k][0] += 1; tot[k][1] += (e - s) / 1e3
print(f"wall {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
  print(f"{t:10.1f} us {n:5d} x  {k}")
upd = [(e - s) / 1e3 for n, s, e in rows if "upd_bf16x3_kernel" in n or "gemm" in n]
if upd:
  print("update kernels, first three / last three (us):", [round(x, 1) for x in upd[:3]], [round(x, 1) for x in upd[-3:]])
