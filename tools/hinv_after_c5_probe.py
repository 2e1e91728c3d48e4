"""The d = 16384 inverse inside a 3-layer C5 run: first run of the process, second run, and after inverses were run alone."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g; g.build()
from mi355q import ops
import c5_model
os.environ["MI355Q_C5_TRACE"] = "hinv"
for i in range(2):
  out = c5_model.run(layers=3, variant="gptq")
  print(f"run {i + 1}:", out["trace"]["hinv"], out["hbm"])
d = 16384
x = torch.randn((16384, d), device="cuda")
prod = ops.gptq_xtx_accum(x, None); del x
def hinv_ms(reps=3):
  out = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gptq_hinv_from_product(prod, 2.0 / 128, 0.01); e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1), 1))
  return out
print("alone:", hinv_ms(4))
out = c5_model.run(layers=3, variant="gptq")
print("run 3:", out["trace"]["hinv"], out["hbm"])
