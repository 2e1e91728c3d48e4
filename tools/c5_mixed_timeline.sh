#!/bin/bash
# BASELINE config 5, mixed recipe: three calls in one process with the phase marks of each (MI355Q_TIMELINE=1) and the idle
# accounting (MI355Q_C5_GAPS=1). Output: gpurun_out/r05_mixed/.
export TMPDIR=/tmp
out=gpurun_out/r05_mixed
mkdir -p $out
MI355Q_TIMELINE=1 MI355Q_C5_GAPS=1 timeout 900 env $EXTRA python tools/c5_second_call.py --variant mixed --calls 3 > $out/calls.jsonl 2> $out/calls.err
python - <<'PY'
import json
for l in open("gpurun_out/r05_mixed/calls.jsonl"):
  if not l.startswith("{"): continue
  r = json.loads(l)
  print("call", r.get("call"), "seconds", r["seconds"], "calibrate", r["calibrate_s"], "quantize+write", r["quantize_and_write_s"], "busy", r["gpu_busy_total_s"], r["gpu_busy_frac"], "section", r.get("section_bytes"), "expected", r.get("expected_section_bytes"))
  for t in r.get("timeline", []): print("    ", t)
  print("    gaps", r.get("idle_gaps"))
PY
tail -3 $out/calls.err
