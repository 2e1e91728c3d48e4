"""Per-kernel PMC sums from a rocprofv3 --pmc database.   usage: python tools/pmc_kernel.py <dir> <kernel substring>"""
import glob, sqlite3, sys
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
  print("tables:", pmc); sys.exit(0)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = db.execute(f"select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from {view} where kernel_name like ? group by kernel_name, counter_name", ("%" + sys.argv[2] + "%",)).fetchall()
for r in rows:
  print(f"{r[0][:60]:60s} {r[1]:32s} total {r[2]:.4g}  dispatches {r[3]}  per dispatch {r[2] / max(1, r[3]):.4g}")
