"""OCTAV clipping search on a 4096 x 4096 weight for every unit length (HIP events).
    python tools/octav_block_bench.py            # MI355Q_OCTAV_WAVE_KERNEL=1 for the one-wave-per-unit kernel"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, __graft_entry__ as g
g.build()
from mi355q import ops
rng = np.random.default_rng(0)
w = torch.from_numpy((rng.standard_normal((4096, 4096)) * 0.02).astype(np.float32)).cuda()
for block in (32, 64, 128, 256, 512, 1024, 4096):
  units = w.numel() // block
  ops.octav_clip(w, units, block, 4); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): ops.octav_clip(w, units, block, 4)
  e1.record(); e1.synchronize()
  print(json.dumps(dict(op="octav_clip int4, 4096 x 4096, sigma 0.02", unit_len=block, ms=round(e0.elapsed_time(e1) / 10, 4),
                        kernel="wave per unit" if os.environ.get("MI355Q_OCTAV_WAVE_KERNEL") else "default")))
