#!/usr/bin/env python3
"""OCTAV clip search: the exact (NumPy-order) kernels against the opt-in one-read kernel (ops.octav_mode("fast")).

  python tools/octav_fast_bench.py
One JSON line per shape: microseconds per call of either kernel (HIP events over 20 calls), the fraction of one read of
the tensor at 8 TB/s, iterations used, and how far the fast kernel's clipping constants are from the exact kernel's
(which equal the reference's bit for bit): largest relative difference and the share of units beyond 1e-6.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def main():
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  gen = torch.Generator(device="cuda").manual_seed(7)
  shapes = [(4096, 4096, 0.02), (4096, 4096, 1.0), (2048, 2048, 0.02), (16384, 2048, 0.02), (4096, 11008, 0.02),
            (2048, 16384, 0.02), (4096 * 32, 128, 0.02), (4096 * 128, 32, 0.02), (4096 * 16, 256, 0.02), (1024, 65536, 0.02)]
  for units, unit_len, sigma in shapes:
    x = (torch.randn((units, unit_len), generator=gen, device="cuda") * sigma).contiguous()
    out = {"units": units, "unit_len": unit_len, "sigma": sigma}
    clips = {}
    for mode in ("exact", "fast"):
      with ops.octav_mode(mode):
        for _ in range(3):
          clip, iters = ops.octav_clip(x.view(-1), units, unit_len, 4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
          clip, iters = ops.octav_clip(x.view(-1), units, unit_len, 4)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
      clips[mode] = clip.double()
      out[mode] = {"us": round(us, 1), "frac_of_one_read": round(x.numel() * 4 / (us * 1e-6) / 8e12, 4), "iterations": int(iters.item())}
    rel = ((clips["fast"] - clips["exact"]).abs() / clips["exact"].abs().clamp_min(1e-30))
    out["max_rel_diff"] = float(rel.max())
    out["units_beyond_1e-6"] = float((rel > 1e-6).double().mean())
    out["speedup"] = round(out["exact"]["us"] / out["fast"]["us"], 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
  main()
