"""Per outer block of the d >= 4096 Cholesky: wall span, time of the chain's kernels (potf2 / solve / update / small GEMMs)
and of the look-ahead GEMMs, from a rocprofv3 kernel trace of tools/hinv_profile.py.
    rocprofv3 --kernel-trace -d /tmp/prof_h -o p -- python tools/hinv_profile.py 16384 ; python tools/hinv_chain_timeline.py /tmp/prof_h"""
import glob
import sqlite3
import sys

db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
first = max(i for i, r in enumerate(rows) if "copy_damped_lower" in r[0])
rows = rows[first:]
t0 = rows[0][1]
end = next((i for i, r in enumerate(rows) if "diag_inverse" in r[0]), len(rows))
chol = rows[:end]
print(f"# Cholesky: {len(chol)} kernels, wall {(chol[-1][2] - t0) / 1e6:.2f} ms")
potf2 = [r for r in chol if "potf2" in r[0]]
print("# block  first-potf2-start(ms)  span(ms)  potf2 us(avg)  solve us(avg)  update us(avg)  big-gemm ms (overlapping)")
for b in range(0, len(potf2), 8):
  grp = potf2[b:b + 8]
  s, e = grp[0][1], (potf2[b + 8][1] if b + 8 < len(potf2) else chol[-1][2])
  inside = [r for r in chol if s <= r[1] < e]

  def avg(tag):
    v = [(r[2] - r[1]) / 1e3 for r in inside if tag in r[0]]
    return sum(v) / len(v) if v else 0.0
  big = sum((r[2] - r[1]) for r in inside if "gemm" in r[0] and (r[2] - r[1]) > 200e3) / 1e6
  print(f"{b // 8:5d}  {(s - t0) / 1e6:9.2f}  {(e - s) / 1e6:8.3f}  {avg('potf2'):8.1f}  {avg('solve_batched') or avg('trsm_panel'):8.1f}"
        f"  {avg('update_batched'):8.1f}  {big:8.2f}")
