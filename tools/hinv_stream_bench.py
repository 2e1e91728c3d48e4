"""mi355q_gptq_hinv_f64 timed from a non-default (non-blocking) torch stream.
usage: python tools/hinv_stream_bench.py [d=16384] [reps=4]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x = torch.randn((4096, d), device="cuda")
H = (x.double().T @ x.double()) / 4096 + torch.eye(d, device="cuda", dtype=torch.float64)
del x
torch.cuda.synchronize()
ref = None
for label, stream in (("default stream", None), ("own stream", torch.cuda.Stream())):
  ts = []
  ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
  with ctx:
    for _ in range(reps):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      r, info = ops.gptq_hinv(H)
      e1.record(); e1.synchronize()
      ts.append(e0.elapsed_time(e1))
  if ref is None:
    ref = r.clone()
  print(label, "ms:", " ".join(f"{t:.2f}" for t in ts), " info", int(info.item()), " max|diff| vs first", float((r - ref).abs().max()))
