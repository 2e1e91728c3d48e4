"""The kernels of one mi355q_gptq_hinv_f64 call outside the Cholesky chain, from a rocprofv3 --kernel-trace database of
tools/hinv_profile.py: python tools/hinv_tail.py <trace dir>"""
import glob, re, sqlite3, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = c.execute("select name, start, end, duration from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "copy_damped_lower" in r[0])
rows = rows[first:]
t0 = rows[0][1]
lastp = max(i for i, r in enumerate(rows) if "potf2" in r[0] or "chol_step" in r[0])
short = lambda n: re.sub(r"\(anonymous namespace\)::|mi355q::|void ", "", n).split("(")[0][:70]
print(f"# whole call {(rows[-1][2] - t0) / 1e6:.2f} ms, {len(rows)} kernels; the Cholesky's last potf2 ends at {(rows[lastp][2] - t0) / 1e6:.2f} ms")
print("# before the first potf2:")
firstp = min(i for i, r in enumerate(rows) if "potf2" in r[0] or "chol_step" in r[0])
for r in rows[:firstp]:
  print(f"  start {(r[1] - t0) / 1e3:9.1f} us  dur {r[3] / 1e3:9.1f} us  {short(r[0])}")
print("# after the last potf2 (start, duration, gap to the end of the kernel before):")
prev = rows[lastp][2]
for r in rows[lastp + 1:]:
  print(f"  start {(r[1] - t0) / 1e3:9.1f} us  dur {r[3] / 1e3:9.1f} us  gap {(r[1] - prev) / 1e3:7.1f}  {short(r[0])}")
  prev = max(prev, r[2])
