import ctypes, os, sys, time, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import _ffi
L = _ffi.lib()
n = 1 << 30
path = "/tmp/io_sweep.bin"
if not os.path.exists(path) or os.path.getsize(path) != n:
  np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8).tofile(path)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
fd = os.open(path, os.O_RDONLY)
for rep in range(3):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  _ffi.check(L.mi355q_file_to_device(fd, 0, n, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(st.cuda_stream)))
  st.synchronize(); dt = time.perf_counter() - t0
  print("threads", os.environ.get("MI355Q_IO_THREADS"), "upload GB/s", round(n / dt / 1e9, 1))
out = "/tmp/io_sweep_out.bin"
for rep in range(3):
  if rep == 0 and os.path.exists(out): os.remove(out)
  fo = os.open(out, os.O_RDWR | os.O_CREAT)
  os.ftruncate(fo, n)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  _ffi.check(L.mi355q_device_to_file(ctypes.c_void_p(dev.data_ptr()), n, fo, 0, ctypes.c_void_p(st.cuda_stream)))
  _ffi.check(L.mi355q_file_io_finish()); dt = time.perf_counter() - t0
  os.close(fo)
  print("threads", os.environ.get("MI355Q_IO_THREADS"), "download GB/s", round(n / dt / 1e9, 1), "(fresh file)" if rep == 0 else "(overwrite)")
