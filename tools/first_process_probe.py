"""Does a fresh box read HBM slower at first? 100 GB of float32 in 128 tensors (what a C5 calibration set is), then one
HBM-bound pass over a tensor after the other (act_minmax: 0.8 GB each), timed call by call for `seconds`:
python tools/first_process_probe.py [seconds=15]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
import torch
from mi355q import ops
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
t0 = time.perf_counter()
xs = [torch.randn((200 << 20,), device="cuda") for _ in range(128)]     # 128 x 0.8 GB
torch.cuda.synchronize()
print(f"allocated and filled {len(xs) * xs[0].numel() * 4 / 1e9:.0f} GB in {time.perf_counter() - t0:.2f} s", flush=True)
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # > 0: first read one float every `stride` bytes of every tensor (translations only)
if stride:
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  acc = sum(x[::stride // 4].sum() for x in xs)
  e1.record()
  torch.cuda.synchronize()
  print(f"touched one float per {stride} bytes of every tensor in {e0.elapsed_time(e1):.2f} ms", flush=True)
rows = []
t_start = time.perf_counter()
k = 0
while time.perf_counter() - t_start < seconds:
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  ops.act_minmax([xs[k % len(xs)]], -3e38, 3e38)
  e1.record()
  torch.cuda.synchronize()
  rows.append((time.perf_counter() - t_start, e0.elapsed_time(e1)))
  k += 1
for i in (0, 1, 2, 5, 10, 50, 100, 128, 129, 200, 500, 1000, 2000, 5000, 10000):
  if i < len(rows):
    print(f"call {i:6d} at {rows[i][0]:6.2f} s: {rows[i][1]:7.3f} ms  ({0.8389 / rows[i][1]:.2f} TB/s)")
n = len(rows)
print(f"calls {n}: first pass over the 128 tensors {sum(r[1] for r in rows[:128]) / 128:.3f} ms each, second pass {sum(r[1] for r in rows[128:256]) / 128:.3f}, last 128 {sum(r[1] for r in rows[-128:]) / 128:.3f}")
