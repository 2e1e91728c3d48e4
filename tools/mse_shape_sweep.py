"""MSE scale + quantize (mi355q_mse_requant_f32) over row lengths: time per 2^24 elements and fraction of the HBM peak of one read + one int8
write; which lengths take the one-launch kernel (complete pairwise tree) and which the two kernels."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import torch, __graft_entry__ as g
g.build()
from mi355q import ops
for cols in (384, 512, 768, 1000, 1024, 2048, 3072, 4096, 5120, 8192, 11008, 12288, 14336, 16384):
  rows = (1 << 24) // cols
  w = torch.randn((rows, cols), device="cuda") * 0.02
  ops.mse_requant(w.view(-1), rows, cols, 0.37755, 4, False); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): ops.mse_requant(w.view(-1), rows, cols, 0.37755, 4, False)
  e1.record(); e1.synchronize()
  ms = e0.elapsed_time(e1) / 20
  print(json.dumps(dict(cols=cols, rows=rows, us=round(ms * 1e3, 1), hbm_frac=round(rows * cols * 5 / ms / 1e6 / 8000, 3))), flush=True)
