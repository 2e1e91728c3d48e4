"""OCTAV clip search, cost per Newton iteration (the masks get sparser as the guess converges).
    python tools/octav_iter_bench.py [rows cols sigma]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
  rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
  sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  rng = np.random.default_rng(12)
  w = torch.from_numpy(rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(sigma)).cuda()
  x, prev = w.view(-1), 0.0
  for it in range(1, 11):
    for _ in range(3):
      ops.octav_clip(x, rows, cols, 4, it, 3.0, False, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      clip, _ = ops.octav_clip(x, rows, cols, 4, it, 3.0, False, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e6
    dens = float((w.abs() >= clip.view(-1, 1)).float().mean())
    print(f"max_iter={it:2d}  {dt:8.1f} us  (+{dt - prev:6.1f})  clip[0]={float(clip[0]):.5f}  selected after={dens:.4f}")
    prev = dt


if __name__ == "__main__":
  main()
