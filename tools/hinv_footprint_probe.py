"""Is the d = 16384 inverse slower when most of the HBM is allocated (inside the 18-layer run, with 106 GB of calibration
activations resident, it takes 76 ms against 56)?  python tools/hinv_footprint_probe.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d = 16384
x = torch.randn((32768, d), device="cuda")
prod = ops.gptq_xtx_accum(x, None); del x

def ms(p, reps=3):
  out = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gptq_hinv_from_product(p, 2.0 / 128, 0.01); e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1), 1))
  return out
print("small footprint:", ms(prod), "allocated GiB", round(torch.cuda.memory_allocated() / 2**30, 1))
ballast = []
for total in (64, 128, 192, 240):
  while len(ballast) < total:
    ballast.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
  torch.cuda.empty_cache()           # the inverse's 4.3 GB workspace is allocated afresh
  print(f"{total} GiB of ballast:", ms(prod), "allocated GiB", round(torch.cuda.memory_allocated() / 2**30, 1))
# the ballast written (touched) rather than only reserved
for b in ballast[:128]:
  b.zero_()
torch.cuda.synchronize(); torch.cuda.empty_cache()
print("128 GiB of it written:", ms(prod))
