"""OCTAV on blockwise units (octav_unit_lanes_kernel): time of one call against the number of Newton iterations.
    python tools/octav_unit_iter_bench.py [unit_len] [sigma]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, __graft_entry__ as g
g.build()
from mi355q import ops
unit = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
w = torch.from_numpy((np.random.default_rng(0).standard_normal((4096, 4096)) * sigma).astype(np.float32)).cuda()
units = w.numel() // unit
prev = 0.0
for it in range(1, 11):
  ops.octav_clip(w, units, unit, 4, it, 3.0, False); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): ops.octav_clip(w, units, unit, 4, it, 3.0, False)
  e1.record(); e1.synchronize()
  us = e0.elapsed_time(e1) * 100
  print(f"unit {unit} sigma {sigma} max_iter={it:2d} {us:8.1f} us (+{us - prev:6.1f})")
  prev = us
