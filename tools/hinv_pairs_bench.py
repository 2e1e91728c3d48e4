#!/usr/bin/env python3
"""d >= 4096 damped inverses from float32 products: one call each (mi355q_gptq_hinv_from_product_f32) against the
batched call that keeps TWO matrices in flight (mi355q_gptq_hinv_from_product_f32_batched).

  python tools/hinv_pairs_bench.py [d=16384] [count=4]
One JSON line: milliseconds per inverse either way, whether the results are bit-identical, and the largest relative
error of an inverse against the exact FP64 inverse of its damped Hessian (d <= 8192 only: the FP64 reference is slow).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def main():
  d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
  count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  forms = []
  for i in range(count):
    gen = torch.Generator(device="cuda").manual_seed(100 + i)
    x = torch.randn((max(4096, d), d), generator=gen, device="cuda")
    forms.append((ops.gptq_xtx_accum(x, None), 2.0 / 16))
    del x

  def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
  single = timed(lambda: [ops.gptq_hinv_from_product(p, a, 0.01) for p, a in forms])
  paired = timed(lambda: ops.gptq_hinv_from_product_batched(forms, 0.01))
  one = [ops.gptq_hinv_from_product(p, a, 0.01) for p, a in forms]
  two = ops.gptq_hinv_from_product_batched(forms, 0.01)
  torch.cuda.synchronize()
  same = all(torch.equal(a[0], b[0]) and int(a[1].item()) == int(b[1].item()) == 0 for a, b in zip(one, two))
  print(json.dumps(dict(d=d, count=count, ms_per_inverse_single_calls=round(single / count, 3),
                        ms_per_inverse_two_in_flight=round(paired / count, 3), speedup=round(single / paired, 3),
                        bit_identical=bool(same))), flush=True)


if __name__ == "__main__":
  main()
