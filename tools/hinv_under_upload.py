"""Does an upload running on the copy stream slow the Hessian inverse (a chain of ~2000 small dependent kernels)?
python tools/hinv_under_upload.py [d=16384]"""
import ctypes, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import _ffi, ops
L = _ffi.lib()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = torch.randn((4 * d if d <= 4096 else 32768, d), device="cuda")
prod = ops.gptq_xtx_accum(x, None); del x
alpha = 2.0 / 128
w = torch.randn((2048, d), device="cuda") * 0.02
sc = (w.abs().amax(dim=1) / 7).contiguous()

def hinv_ms(reps=3):
  ops.gptq_hinv_from_product(prod, alpha, 0.01); torch.cuda.synchronize()
  out = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hinv, _ = ops.gptq_hinv_from_product(prod, alpha, 0.01); e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1), 1))
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); ops.gptq_apply(w, hinv, sc, None, 1, 0, 4, False, False, 8); e1.record(); torch.cuda.synchronize()
  return out, round(e0.elapsed_time(e1), 1)

print("alone: hinv ms", *hinv_ms())
n = 1 << 30
path = "/tmp/hinv_under_upload.bin"
if not os.path.exists(path) or os.path.getsize(path) != n:
  np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8).tofile(path)
fd = os.open(path, os.O_RDONLY)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for label, prio in (("normal-priority copy stream", 0), ("lowest-priority copy stream", lo)):
  st = torch.cuda.Stream(priority=prio)
  dev = torch.empty(n, dtype=torch.uint8, device="cuda")
  stop = threading.Event(); moved = [0]
  def pump():
    torch.cuda.set_device(0)
    while not stop.is_set():
      _ffi.check(L.mi355q_file_to_device(fd, 0, n, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(st.cuda_stream)))
      st.synchronize(); moved[0] += n
  t = threading.Thread(target=pump); t0 = time.perf_counter(); t.start()
  time.sleep(0.1)
  res = hinv_ms()
  stop.set(); t.join()
  print(f"under uploads ({label}, priority {prio}): hinv ms", *res, " upload GB/s", round(moved[0] / (time.perf_counter() - t0) / 1e9, 1))
