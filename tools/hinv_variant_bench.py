#!/usr/bin/env python3
"""The damped inverse with a given build of libmi355q.so (tools/build_variant.sh), one process per library:

  python tools/hinv_variant_bench.py <lib.so | default> [d=16384] [count=3]
One JSON line: milliseconds per inverse (HIP events, 3 repetitions of `count` single calls), and a checksum of the first
inverse (equal checksums across libraries = the same bits)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def main():
  which = sys.argv[1] if len(sys.argv) > 1 else "default"
  d = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
  count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
  from mi355q import _ffi
  if which != "default":
    _ffi.LIB_PATH = os.path.abspath(which)
  else:
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
  run = lambda: [ops.gptq_hinv_from_product(p, a, 0.01) for p, a in forms]
  run()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3):
    out = run()
  e1.record()
  torch.cuda.synchronize()
  first = out[0][0].contiguous().cpu().numpy()
  print(json.dumps(dict(lib=os.path.basename(which), d=d, count=count, ms_per_inverse=round(e0.elapsed_time(e1) / 3 / count, 3),
                        info=int(out[0][1].item()), sha256_16=hashlib.sha256(first.tobytes()).hexdigest()[:16])), flush=True)


if __name__ == "__main__":
  main()
