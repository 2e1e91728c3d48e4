"""Upload of a large model file to HBM: a fresh mmap + pageable copies (what Quantizer(path) does) against
pread() into pinned staging buffers on reader threads + asynchronous copies. python tools/h2d_big_probe.py [GiB] [dir]"""
import mmap, os, sys, time, threading, warnings
import numpy as np, torch
warnings.simplefilter("ignore")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 6
d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
n = int(gib * (1 << 30))
path = os.path.join(d, "h2d_big_probe.bin")
blk = np.random.default_rng(0).standard_normal(1 << 24, dtype=np.float32).tobytes()
with open(path, "wb") as f:
  for _ in range(n // len(blk)): f.write(blk)
n = n // len(blk) * len(blk)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
piece = 128 << 20

def mmap_pageable():
  f = open(path, "rb"); mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
  arr = np.frombuffer(mm, dtype=np.uint8)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for o in range(0, n, piece):
    dev[o:o + piece].copy_(torch.from_numpy(arr[o:o + piece]), non_blocking=True)
  torch.cuda.synchronize()
  return time.perf_counter() - t0

def pread_pinned(threads, chunk=32 << 20, slots=8):
  fd = os.open(path, os.O_RDONLY)
  bufs = [torch.empty(chunk, dtype=torch.uint8).pin_memory() for _ in range(slots)]
  views = [memoryview(b.numpy()) for b in bufs]
  free = [threading.Semaphore(1) for _ in range(slots)]
  ready = [threading.Semaphore(0) for _ in range(slots)]
  events = [torch.cuda.Event() for _ in range(slots)]
  nchunks = (n + chunk - 1) // chunk
  torch.cuda.synchronize(); t0 = time.perf_counter()
  def reader(k):
    for c in range(k, nchunks, threads):
      s = c % slots
      free[s].acquire()
      size = min(chunk, n - c * chunk)
      got = 0
      while got < size:
        got += os.preadv(fd, [views[s][got:size]], c * chunk + got)
      ready[s].release()
  ts = [threading.Thread(target=reader, args=(k,)) for k in range(threads)]
  [t.start() for t in ts]
  copy_stream = torch.cuda.Stream()
  pending = {}
  for c in range(nchunks):
    s = c % slots
    ready[s].acquire()
    size = min(chunk, n - c * chunk)
    with torch.cuda.stream(copy_stream):
      dev[c * chunk:c * chunk + size].copy_(bufs[s][:size], non_blocking=True)
      events[s].record()
    # release the slot once its copy is done (waiter thread-free: wait lazily before reuse)
    def rel(s=s):
      events[s].synchronize(); free[s].release()
    threading.Thread(target=rel).start()
  [t.join() for t in ts]
  torch.cuda.synchronize()
  os.close(fd)
  return time.perf_counter() - t0

for name, fn in (("mmap + pageable copies", mmap_pageable), ("pread x2 -> pinned -> async", lambda: pread_pinned(2)),
                 ("pread x4 -> pinned -> async", lambda: pread_pinned(4)), ("pread x8 -> pinned -> async", lambda: pread_pinned(8))):
  best = min(fn() for _ in range(2))
  print(f"{name:32s} {best * 1e3:8.1f} ms  {n / best / 1e9:6.1f} GB/s", flush=True)
os.remove(path)
