"""GPTQ OBS apply (mi355q_gptq_apply_f32) on the Gemma-2B layer shapes.
    python tools/gptq_apply_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  gen = torch.Generator(device="cuda")
  gen.manual_seed(7)
  for rows, d in ((256, 2048), (2048, 2048), (16384, 2048), (2048, 16384)):
    x = torch.randn((4096, d), generator=gen, device="cuda")
    hinv, info = ops.gptq_hinv(ops.gptq_xtx(x, 2.0 / 4096))
    assert int(info.item()) == 0
    w = torch.randn((rows, d), generator=gen, device="cuda") * 0.02
    scale = (w.abs().amax(dim=1) / 7).contiguous()
    for _ in range(2):
      ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
      ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8)
    torch.cuda.synchronize()
    print(json.dumps(dict(op="gptq_apply int4 channelwise", rows=rows, d=d,
                          ms=round((time.perf_counter() - t0) / reps * 1e3, 3))))
    del x, hinv, w


if __name__ == "__main__":
  main()
