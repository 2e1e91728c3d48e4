"""Is the d = 16384 inverse slower right after seconds of Hessian products at the socket's power limit (as inside the 18-layer
run: 76 ms against 56)?   python tools/hinv_after_load_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d = 16384
x = torch.randn((16384, d), device="cuda")
prod = ops.gptq_xtx_accum(x, None)

def hinv_ms(reps):
  out = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gptq_hinv_from_product(prod, 2.0 / 128, 0.01); e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1), 1))
  return out
ops.gptq_hinv_from_product(prod, 2.0 / 128, 0.01); torch.cuda.synchronize()
print("cold:", hinv_ms(3))
for seconds in (0.5, 2.0):
  t0 = time.time(); n = 0
  while time.time() - t0 < seconds:
    for _ in range(4): p2 = ops.gptq_xtx_accum(x, None)
    torch.cuda.synchronize()
  print(f"right after {seconds} s of Hessian products:", hinv_ms(6))
time.sleep(2.0)
print("after 2 s idle:", hinv_ms(3))
