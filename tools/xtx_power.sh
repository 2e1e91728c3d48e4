#!/bin/bash
# Socket power and shader clock while the Hessian product runs in a loop (GPU box):  bash tools/xtx_power.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for pr in 0 66 1; do
  echo "== MI355Q_XTX_PROBE=$pr (0: the kernel; 66: MFMAs only; 1: staging only)"
  MI355Q_XTX_PROBE=$pr python - <<PY &
import os, sys, time, torch
sys.path.insert(0, "$R/ai-edge-quantizer_amd"); sys.path.insert(0, "$R")
import __graft_entry__ as g; g.build()
from mi355q import ops
x = torch.randn((16384, 16384), device="cuda")
t0 = time.time(); n = 0
while time.time() - t0 < 9:
  for _ in range(10): p = ops.gptq_xtx_accum(x, None)
  torch.cuda.synchronize(); n += 10
print("calls/s", n / (time.time() - t0))
PY
  PID=$!
  sleep 5
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk|Power" | tr '\n' ' '; echo; sleep 0.7; done
  wait $PID
done
