#!/usr/bin/env python3
"""OSCAR stage timings on one MI355X, HIP events on the launch stream, weights
resident in HBM. Shapes: FULLY_CONNECTED weights [out_ch, in_ch] of a Gemma-2B-like layer.

    python tools/oscar_bench.py [--shapes 16384x2048,2048x16384] [--api]
Prints one JSON object per line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

g.build()
from mi355q import ops, qtyping  # noqa: E402
from mi355q.algorithms.uniform_quantize import oscar  # noqa: E402


def timed(fn, iters=5, warm=2):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  b.synchronize()
  return a.elapsed_time(b) / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--shapes", default="16384x2048,2048x16384,4096x4096")
  ap.add_argument("--api", action="store_true", help="also time get_tensor_quant_params end to end")
  args = ap.parse_args()
  gen = torch.Generator(device="cuda").manual_seed(0)
  for shape in args.shapes.split(","):
    n, d = (int(v) for v in shape.split("x"))
    w = torch.randn((n, d), generator=gen, device="cuda")
    rng = np.random.default_rng(1)
    mu2 = np.exp(rng.normal(size=d) * 1.5)
    s = ops._f64_dev(np.exp(rng.normal(size=d) * 0.2))
    m = ops._f64_dev(mu2)
    fbytes = n * d * 4
    ms = timed(lambda: ops.oscar_col_sumsq(w, False))
    print(json.dumps(dict(stage="col_sumsq", shape=shape, ms=round(ms, 4), gbps=round(fbytes / ms / 1e6, 1))), flush=True)
    for gran, gsz in (("CHANNELWISE", d), ("BLOCKWISE_32", 32), ("BLOCKWISE_128", 128)):
      ms = timed(lambda: ops.oscar_group_terms(w, s, gsz))
      print(json.dumps(dict(stage="group_terms", shape=shape, g=gsz, ms=round(ms, 4), gbps=round(fbytes / ms / 1e6, 1))), flush=True)
      _, winner, wsq = ops.oscar_group_terms(w, s, gsz)
      ms = timed(lambda: ops.oscar_winner_energy(winner, wsq, d, gsz))
      print(json.dumps(dict(stage="winner_energy", shape=shape, g=gsz, ms=round(ms, 4))), flush=True)
      groups = d // gsz
      u = ops._f64_dev(np.full(groups, 0.01))
      noise = ops._f64_dev(np.full(groups, 0.005))
      ms = timed(lambda: ops.oscar_clip_bounds(w, s, m, gsz, u, noise, 7, gsz != d, True, True), iters=3, warm=1)
      print(json.dumps(dict(stage="clip_bounds", shape=shape, g=gsz, ms=round(ms, 4), gbps=round(fbytes / ms / 1e6, 1))), flush=True)
      scale = ops._f64_dev(np.full(n * groups, 0.05))
      ms = timed(lambda: ops.oscar_quantize(w, s, scale, gsz, -8, 7))
      print(json.dumps(dict(stage="quantize", shape=shape, g=gsz, ms=round(ms, 4), gbps=round(fbytes / ms / 1e6, 1))), flush=True)
      if args.api:
        wh = w.cpu().numpy()
        cfg = qtyping.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=qtyping.QuantGranularity[gran])
        info = qtyping.OpInfo(op=qtyping.OperatorT(), op_name=qtyping.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                              op_quant_config=qtyping.OpQuantizationConfig(weight_tensor_config=cfg))
        best = best_host = None
        for _ in range(4):          # the first calls grow the caching allocator (GiB-sized workspaces)
          t = time.perf_counter()
          res = oscar.get_tensor_quant_params(info, cfg, wh, {"mu2": mu2})
          torch.cuda.synchronize()
          dt = time.perf_counter() - t
          np.asarray(res.quantized_data)          # large results stay in HBM until somebody asks
          dt_host = time.perf_counter() - t
          best = dt if best is None else min(best, dt)
          best_host = dt_host if best_host is None else min(best_host, dt_host)
        print(json.dumps(dict(stage="api_get_tensor_quant_params", shape=shape, granularity=gran,
                              seconds=round(best, 4), seconds_with_int8_copied_to_host=round(best_host, 4))),
              flush=True)


if __name__ == "__main__":
  main()
