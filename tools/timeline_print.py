"""Pretty-prints the MI355Q_TIMELINE=1 marks of a tools/c5_model.py / tools/file_bench.py JSON line read from stdin."""
import json, sys
for line in sys.stdin:
  line = line.strip()
  if not line.startswith("{"):
    continue
  d = json.loads(line)
  tl = d.get("timeline")
  if not tl:
    if "workload" in d:
      print("# " + json.dumps({k: d[k] for k in ("workload", "seconds", "gbps") if k in d}))
    continue
  head = {k: d[k] for k in ("seconds", "calibrate_s", "quantize_and_write_s", "gpu_busy_total_s", "gpu_busy_frac", "gpu_busy_s_rank0", "idle_gaps") if k in d}
  print("# " + json.dumps(head))
  print("#   ms since the call began (host)   [ms with the GPU drained, where waited for]   hipMallocs so far   mark")
  prev = None
  for row in tl:
    label, a = row[0], row[1]
    if a < 0:
      continue            # marks of the untimed preparation
    b = row[2] if len(row) > 2 else a
    n = row[3] if len(row) > 3 else ""
    step = "" if prev is None else f"(+{a - prev:7.1f})"
    drained = f"[{b:8.1f}]" if abs(b - a) > 0.05 else " " * 10
    print(f"  {a:9.1f} {step:>11} {drained} {str(n):>5}  {label}")
    prev = b
