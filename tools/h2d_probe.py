"""How fast do the weights of an mmap'd model file reach HBM? pageable .to(), registered mapping, threaded staging."""
import mmap, os, sys, time, threading
import numpy as np, torch
n = 8 * 4096 * 11008 * 4
path = "/tmp/h2d_probe.bin"
if not os.path.exists(path) or os.path.getsize(path) != n:
  with open(path, "wb") as f:
    blk = np.random.default_rng(0).standard_normal(1 << 24, dtype=np.float32).tobytes()
    for _ in range(n // len(blk)): f.write(blk)
    f.write(blk[: n % len(blk)])
f = open(path, "rb")
mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
arr = np.frombuffer(mm, dtype=np.uint8)
_ = arr[:: 4096].sum()    # page cache warm
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
def timed(label, fn, reps=3):
  best = 1e9
  for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
  print(f"{label:50s} {best * 1e3:8.1f} ms  {n / best / 1e9:6.1f} GB/s")
# (a) what the product path does: one pageable copy per tensor
per = n // 8
def pageable():
  for i in range(8): dev[i * per:(i + 1) * per].copy_(torch.from_numpy(arr[i * per:(i + 1) * per]))
import warnings; warnings.simplefilter("ignore")
timed("pageable copy_, 8 tensors", pageable)
# (b) register the mapping
rt = torch.cuda.cudart()
addr = arr.ctypes.data
t0 = time.perf_counter()
for flags in (0, 8):   # 8 = cudaHostRegisterReadOnly
  rc = rt.cudaHostRegister(addr, n, flags)
  print("cudaHostRegister flags", flags, "->", rc, f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
  if int(rc) == 0: break
if int(rc) == 0:
  src = torch.from_numpy(arr)
  timed("registered mapping, one async copy", lambda: dev.copy_(src, non_blocking=True))
  t0 = time.perf_counter(); rt.cudaHostUnregister(addr); print(f"unregister {(time.perf_counter() - t0) * 1e3:.1f} ms")
# (c) threaded staging through pinned buffers
chunk = 32 << 20
nchunks = (n + chunk - 1) // chunk
for nthreads in (2, 4, 8):
  pins = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(2 * nthreads)]
  pin_np = [p.numpy() for p in pins]
  streams = [torch.cuda.Stream() for _ in range(nthreads)]
  def staged():
    def work(tid):
      with torch.cuda.stream(streams[tid]):
        evs = [None, None]
        for j, c in enumerate(range(tid, nchunks, nthreads)):
          slot = tid * 2 + (j & 1)
          if evs[j & 1] is not None: evs[j & 1].synchronize()
          lo, hi = c * chunk, min(n, (c + 1) * chunk)
          np.copyto(pin_np[slot][: hi - lo], arr[lo:hi])
          dev[lo:hi].copy_(pins[slot][: hi - lo], non_blocking=True)
          ev = torch.cuda.Event(); ev.record(streams[tid]); evs[j & 1] = ev
    ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in ts]; [t.join() for t in ts]
  timed(f"pinned staging, {nthreads} threads x 32 MiB chunks", staged)
