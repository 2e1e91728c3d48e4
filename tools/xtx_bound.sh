#!/bin/bash
# Round 5: what bounds the Hessian product -- tools/xtx_bound.py plainly, with amd-smi's full metric dump taken mid-run (throttle /
# violation status, temperatures, power limit), run ON THE GPU BOX from the repo root:  gpurun -- 'bash tools/xtx_bound.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05d
mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
python tools/xtx_bound.py 6 > "$O/xtx_plain3.json" 2> "$O/xtx_plain3.err" &
BG=$!
sleep 22
for i in 1 2 3; do amd-smi metric -g 0 --json > "$O/amd_smi_full_under_load_$i.json" 2>&1; sleep 1; done
amd-smi static -g 0 --limit --json > "$O/amd_smi_limits.json" 2>&1
wait $BG
amd-smi metric -g 0 --json > "$O/amd_smi_full_idle.json" 2>&1
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
def flat(o, p=""):
  if isinstance(o, dict):
    for k, v in o.items(): yield from flat(v, p + k + ".")
  elif isinstance(o, list):
    for i, v in enumerate(o): yield from flat(v, p + str(i) + ".")
  else: yield p[:-1], o
for name in ("amd_smi_full_under_load_2", "amd_smi_full_idle", "amd_smi_limits"):
  print("##", name)
  try:
    d = json.load(open(f"{O}/{name}.json"))
  except Exception as e:
    print("unreadable:", e, open(f"{O}/{name}.json").read()[:300]); continue
  for k, v in flat(d):
    lk = k.lower()
    if any(w in lk for w in ("power", "throttl", "violation", "temper", "ppt", "limit", "gfx_0", "hotspot", "acc", "usage", "activity", "voltage")) and v != "N/A":
      print(" ", k, v)
print(open(f"{O}/xtx_plain3.json").read()[:600])
PY
