"""Fused symmetric requantization (mi355q_requant_sym_f32_batched, 16 buffers per launch) over row lengths and block sizes:
fraction of the HBM peak of SURVEY 8d's algorithmic bytes. Finds shapes that fall off the fast kernels."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import torch, __graft_entry__ as g
g.build()
from mi355q import ops
def run(rows, cols, block, bits, packed):
  xs = [torch.randn((rows, cols), device="cuda") * 0.02 for _ in range(8)]
  b = ops.RequantBatch(xs, block=block, bits=bits, want_q=not packed, want_packed=packed, want_scale_f16=bool(block))
  for _ in range(20): b.run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): b.run()
  e1.record(); e1.synchronize()
  ms = e0.elapsed_time(e1) / 50
  n = rows * cols
  nscale = rows * (cols // block if block else 1)
  alg = 8 * (n * 4 + (n * bits // 8 if packed else n) + nscale * (2 if block else 4))
  print(json.dumps(dict(rows=rows, cols=cols, block=block, bits=bits, packed=packed, us_per_buffer=round(ms * 1e3 / 8, 2), hbm_frac=round(alg / ms / 1e6 / 8000, 3))), flush=True)
for cols in (384, 512, 768, 1000, 1024, 2048, 3072, 4096, 5120, 8192, 11008, 14336, 16384):
  run((1 << 24) // cols, cols, 0, 8, False)
for cols in (768, 4096, 11008):
  run((1 << 24) // cols, cols, 0, 4, True)
for block in (32, 64, 128, 256):
  for cols in (768, 4096, 11008):
    run((1 << 24) // cols, cols, block, 4, True)
