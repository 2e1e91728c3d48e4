"""The three GPTQ stages over other models' layer orders (d = 768 .. 14336, not only Gemma-2B's 2048 / 16384): Hessian product
of 16384 tokens, damped inverse, OBS apply of 4096 rows -- time and rate, to find orders that fall off the fast kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import torch, __graft_entry__ as g
g.build()
from mi355q import ops
def timed(fn, reps=3):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps
gen = torch.Generator(device="cuda").manual_seed(3)
for d in [int(a) for a in sys.argv[1:]] or (768, 1024, 1536, 3072, 4096, 5120, 8192, 11008, 14336):
  tokens = 16384
  x = torch.randn((tokens, d), generator=gen, device="cuda")
  t_x = timed(lambda: ops.gptq_xtx_accum(x, None))
  prod = ops.gptq_xtx_accum(x, None)
  t_h = timed(lambda: ops.gptq_hinv_from_product(prod, 2.0 / tokens), 2)
  hinv, info = ops.gptq_hinv_from_product(prod, 2.0 / tokens)
  rows = 4096
  w = torch.randn((rows, d), generator=gen, device="cuda") * 0.02
  scale = (w.abs().amax(dim=1) / 7).contiguous()
  t_a = timed(lambda: ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8), 2)
  print(json.dumps(dict(d=d, info=int(info.item()), xtx_ms=round(t_x * 1e3, 3), xtx_TF_bf16_equiv=round(6 * tokens * d * d / 2 * 2 / t_x / 1e12, 1),
                        hinv_ms=round(t_h * 1e3, 3), hinv_TF=round(d ** 3 / t_h / 1e12, 2),
                        apply_ms=round(t_a * 1e3, 3), apply_TF=round(2 * rows * d * d / t_a / 1e12, 1))), flush=True)
  del x, prod, hinv, w
