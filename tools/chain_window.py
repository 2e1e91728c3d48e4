"""Kernels of a rocprofv3 --kernel-trace database in a time window: start, duration, queue, name -- to see what runs beside
the Hessian inverse's serial chain.   python tools/chain_window.py <dir> [nth copy_damped_lower=last] [skip us=20000] [count=60]"""
import glob, sqlite3, sys
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "copy_damped_lower_kernel<float" in r[0] or ("copy_damped_lower" in r[0] and "float" in r[0])]
marks = marks or [i for i, r in enumerate(rows) if "copy_damped_lower" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 20000.0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
t0 = rows[marks[which]][1]
sel = [r for r in rows if r[1] >= t0 + skip * 1e3][:count]
for name, s, e, qid in sel:
  short = name.replace("mi355q::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:60]
  print(f"start {(s - t0) / 1e3:10.1f}  dur {(e - s) / 1e3:8.1f}  q {qid}  {short}")
