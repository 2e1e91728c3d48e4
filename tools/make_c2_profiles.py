#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of the C2 bench passes (gpurun_out/prof_c2, pmc_fetch,
pmc_write + the un-profiled bench line) into the summaries kept under profiles/."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
bench_log = sys.argv[2] if len(sys.argv) > 2 else os.path.join(G, "bench5.log")


def newest(d):
  return sorted(glob.glob(os.path.join(G, d, "**", "*.db"), recursive=True), key=os.path.getmtime)[-1]


def counter(db, name):
  c = sqlite3.connect(db)
  return c.execute(
      "select kernel_name, count(*), avg(value), min(value), max(value), avg(duration) from counters_collection"
      " where counter_name=? and kernel_name like '%requant_rows_kernel%' group by kernel_name order by count(*) desc",
      (name,)).fetchone()


bench = json.loads(open(bench_log).read())
trace = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), newest("prof_c2")],
                       capture_output=True, text=True).stdout
with open(os.path.join(ROOT, "profiles", f"{tag}_c2_rowwise_int8_kernel_trace.txt"), "w") as f:
  f.write(trace)
  f.write("# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --extras 0\n")
  f.write(f"# un-profiled bench.py (same build): roofline.launch_ms = {bench['roofline']['launch_ms']} ms,"
          f" frac = {bench['roofline']['frac']}, value = {bench['value']} GB/s\n")

fe, wr = counter(newest("pmc_fetch"), "FETCH_SIZE"), counter(newest("pmc_write"), "WRITE_SIZE")
fb, wb = fe[2] * 1024 * 2, wr[2] * 1024
alg = 16 * (4096 * 4096 * 4 + 4096 * 4096 + 4096 * 4 + 4096)
kname = fe[0].split("(")[0].replace("void ", "")
txt = f"""# C2 batched kernel {kname}: HBM traffic from PMC counters
# commands (separate passes, as MI355X_MICROARCH.md section HBM prescribes):
#   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --extras 0
#   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --extras 0
counter      dispatches  mean_per_dispatch      min              max           mean_duration_ns
FETCH_SIZE   {fe[1]:9d}  {fe[2]:18.3f} {fe[3]:16.3f} {fe[4]:16.3f}  {fe[5]:12.1f}
WRITE_SIZE   {wr[1]:9d}  {wr[2]:18.3f} {wr[3]:16.3f} {wr[4]:16.3f}  {wr[5]:12.1f}
# units: KiB. gfx950 correction: FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
# (16 B/lane) streaming read -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is.
read_bytes_per_launch   = 2 * {fe[2]:.3f} * 1024 = {fb:.0f}   (input: 16 x 64 MiB = {16*4096*4096*4})
write_bytes_per_launch  = {wr[2]:.3f} * 1024     = {wb:.0f}   (int8 + scales: {16*(4096*4096+4096*4)})
hbm_bytes_per_launch    = {fb+wb:.0f}
algorithmic_bytes       = {alg}   (SURVEY 8d: 83 906 560 B x 16 buffers)
traffic / algorithmic   = {(fb+wb)/alg:.4f}
"""
open(os.path.join(ROOT, "profiles", f"{tag}_c2_pmc_traffic.txt"), "w").write(txt)
json.dump({"kernel": kname, "hbm_bytes_per_launch": round(fb + wb), "read_bytes": round(fb),
           "write_bytes": round(wb), "algorithmic_bytes": alg,
           "source": f"profiles/{tag}_c2_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 fetch correction)"},
          open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w").write(json.dumps(bench) + "\n")
print(txt)
print(trace[:600])
