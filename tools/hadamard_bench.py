"""Block-diagonal Hadamard rotation (FWHT) timing at the Gemma / Llama shapes: HBM read + write
against the 8 TB/s peak.   python tools/hadamard_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  for rows, cols, h in ((4096, 4096, 4096), (2048, 2048, 2048), (16384, 2048, 2048), (4096, 8192, 8192),
                        (2048, 16384, 16384), (4096, 11008, 256)):
    pool = [torch.randn((rows, cols), device="cuda") for _ in range(max(2, (1 << 30) // (rows * cols * 4)))]
    for x in pool[:2]:
      ops.hadamard_rotate(x, h)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps):
      for x in pool:
        ops.hadamard_rotate(x, h)
    b.record()
    b.synchronize()
    ms = a.elapsed_time(b) / (reps * len(pool))
    print(json.dumps({"shape": [rows, cols], "h": h, "us": round(ms * 1e3, 2),
                      "read_write_GBps": round(2 * rows * cols * 4 / ms / 1e6, 1),
                      "hbm_frac": round(2 * rows * cols * 4 / ms / 1e6 / 8000, 4)}))


if __name__ == "__main__":
  main()
