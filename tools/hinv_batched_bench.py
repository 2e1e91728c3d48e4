"""Amortised time of a d x d damped inverse when `count` independent Hessians go out together
(mi355q_gptq_hinv_f64_batched) against one call per Hessian.  python tools/hinv_batched_bench.py [d] [count]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def main():
  d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
  count = int(sys.argv[2]) if len(sys.argv) > 2 else 54
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  hs = []
  for i in range(count):
    gen = torch.Generator(device="cuda").manual_seed(i)
    x = torch.randn((4 * d, d), generator=gen, device="cuda")
    hs.append(ops.gptq_xtx(x, 2.0 / 16))

  def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
  single = timed(lambda: [ops.gptq_hinv(h, 0.01) for h in hs])
  batched = timed(lambda: ops.gptq_hinv_batched(hs, 0.01))
  print(json.dumps(dict(d=d, count=count, ms_per_inverse_single_calls=round(single / count, 4),
                        ms_per_inverse_batched=round(batched / count, 4), speedup=round(single / batched, 2))))


if __name__ == "__main__":
  main()
