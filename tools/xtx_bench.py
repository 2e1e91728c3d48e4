"""X^T X (mi355q_gptq_xtx_accum_f32) on one GPU: the two-way float16 split against the three-way bfloat16 split.
usage: python tools/xtx_bench.py [d tokens]...   (default: the Hessians of a Gemma-2B layer, 65 536 tokens)"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops

args = [int(a) for a in sys.argv[1:]]
shapes = list(zip(args[0::2], args[1::2])) or [(2048, 65536), (16384, 16384), (16384, 65536)]
for d, n in shapes:
  gen = torch.Generator(device="cuda").manual_seed(d + n)
  x = torch.randn((n, d), generator=gen, device="cuda") * torch.exp2(torch.randint(-6, 6, (1, d), generator=gen, device="cuda").float()) + 0.1
  if os.environ.get("XTX_BENCH_ZERO"): x.zero_()   # (power probe: no data toggling in the matrix cores)
  strip = slice(d // 2, d // 2 + 128)
  ref = x.double().T @ x[:, strip].double()
  mag = x.double().abs().T @ x[:, strip].double().abs()
  row = {"d": d, "tokens": n, "triangle_TFLOP": round(n * d * (d + 128) / 1e12, 2)}
  for name, env in (("f16x2", "MI355Q_XTX_F16X2"), ("bf16x3", None)):
    if env: os.environ[env] = "1"
    prod = ops.gptq_xtx_accum(x, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
      prod = ops.gptq_xtx_accum(x, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    h = ops.gptq_xtx_finish(prod, 1.0)
    err = float(((h[:, strip] - ref).abs() / mag).max())
    row[name] = {"ms": round(ms, 3), "max_err_over_sum_abs": err}
    if env: del os.environ[env]
    del prod, h
  print(json.dumps(row), flush=True)
  del x, ref, mag
