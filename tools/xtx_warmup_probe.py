"""Is the first heavy kernel work on a fresh box slower? The d = 16384 Hessian product (16384 tokens per call) timed call
by call for `seconds`, from the very first call of the process: python tools/xtx_warmup_probe.py [seconds=12]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
import torch
from mi355q import ops
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
d, n = 16384, 16384
x = torch.randn((n, d), device="cuda")
torch.cuda.synchronize()
prod = None
t_start = time.perf_counter()
rows = []
while time.perf_counter() - t_start < seconds:
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  prod = ops.gptq_xtx_accum(x, prod)
  e1.record()
  torch.cuda.synchronize()
  rows.append((time.perf_counter() - t_start, e0.elapsed_time(e1)))
for k in (0, 1, 2, 3, 5, 10, 20, 50, 100, 200, 400):
  if k < len(rows):
    print(f"call {k:4d} at {rows[k][0]:6.2f} s: {rows[k][1]:7.2f} ms")
print(f"calls {len(rows)}, first 10 mean {sum(r[1] for r in rows[:10]) / 10:.2f} ms, last 10 mean {sum(r[1] for r in rows[-10:]) / 10:.2f} ms")
