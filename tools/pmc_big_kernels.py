import glob, sqlite3, sys
db = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1])
rows = db.execute("select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE' and duration > 500000 order by duration desc limit 30").fetchall()
for name, v, dur in rows:
  short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
  print(f"{dur/1e3:10.1f} us  FETCH_SIZE {v*1024/1e9:8.2f} GB raw (x2 = {2*v*1024/1e9:8.2f} GB)  -> {2*v*1024/dur:7.1f} GB/s  {short}")
