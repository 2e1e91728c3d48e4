"""Uploads from a FRESH mapping of the model file (what Quantizer(path) sees): one thread vs several."""
import mmap, os, sys, time, threading, warnings
import numpy as np, torch
warnings.simplefilter("ignore")
n = 8 * 4096 * 11008 * 4
path = "/tmp/h2d_probe.bin"
if not os.path.exists(path) or os.path.getsize(path) != n:
  with open(path, "wb") as f:
    blk = np.random.default_rng(0).standard_normal(1 << 24, dtype=np.float32).tobytes()
    for _ in range(n // len(blk)): f.write(blk)
    f.write(blk[: n % len(blk)])
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
per = n // 8
def fresh():
  f = open(path, "rb")
  mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
  return np.frombuffer(mm, dtype=np.uint8)
def run(nthreads):
  arr = fresh()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  def work(ids):
    for i in ids: dev[i * per:(i + 1) * per].copy_(torch.from_numpy(arr[i * per:(i + 1) * per]))
  if nthreads == 1:
    work(range(8))
  else:
    ts = [threading.Thread(target=work, args=(range(k, 8, nthreads),)) for k in range(nthreads)]
    [t.start() for t in ts]; [t.join() for t in ts]
  torch.cuda.synchronize()
  return time.perf_counter() - t0
for nt in (1, 4):
  best = min(run(nt) for _ in range(3))
  print(f"fresh mapping, {nt} thread(s): {best * 1e3:7.1f} ms  {n / best / 1e9:5.1f} GB/s")
# what runtime.to_device does: a float32 view at a 16-byte-aligned (not page-aligned) offset, .to() into a new tensor
def run_to(off):
  arr = fresh()
  per4 = (per - 4096) // 4
  torch.cuda.synchronize(); t0 = time.perf_counter()
  outs = []
  for i in range(8):
    v = arr[i * per + off: i * per + off + per4 * 4].view(np.float32)
    outs.append(torch.from_numpy(np.ascontiguousarray(v)).to("cuda", non_blocking=True))
  torch.cuda.synchronize()
  return time.perf_counter() - t0
for off in (0, 1232, 48):
  best = min(run_to(off) for _ in range(3))
  print(f".to() of float32 views at offset {off:5d}: {best * 1e3:7.1f} ms  {n / best / 1e9:5.1f} GB/s")
