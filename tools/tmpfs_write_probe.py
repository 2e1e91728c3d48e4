#!/usr/bin/env python3
"""What writing a fresh 1 GB output file costs on the memory-backed scratch directory (the tail of a C5 mixed call):
pwrite of 8 MiB pieces from T threads into (a) a new file, (b) a file whose pages were allocated first (posix_fallocate),
(c) a file that is overwritten in place; and the allocation alone. One JSON line each.

  python tools/tmpfs_write_probe.py [--dir /dev/shm] [--gib 1.0]
"""
import argparse
import json
import os
import threading
import time


def write_all(fd, total, piece, threads, buf):
  offs = list(range(0, total, piece))
  lock = threading.Lock()

  def work():
    while True:
      with lock:
        if not offs:
          return
        o = offs.pop()
      os.pwrite(fd, buf[:min(piece, total - o)], o)
  ts = [threading.Thread(target=work) for _ in range(threads)]
  t0 = time.perf_counter()
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  return time.perf_counter() - t0


def copy_all(view, total, piece, threads, src):
  """The same pieces copied into a shared mapping of the file (no inode lock: page faults and copies run in parallel)."""
  import numpy as np
  dst = np.frombuffer(view, dtype=np.uint8)
  s = np.frombuffer(src, dtype=np.uint8)
  offs = list(range(0, total, piece))
  lock = threading.Lock()

  def work():
    while True:
      with lock:
        if not offs:
          return
        o = offs.pop()
      n = min(piece, total - o)
      np.copyto(dst[o:o + n], s[:n])
  ts = [threading.Thread(target=work) for _ in range(threads)]
  t0 = time.perf_counter()
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  return time.perf_counter() - t0


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--dir", default="/dev/shm")
  ap.add_argument("--gib", type=float, default=1.0)
  a = ap.parse_args()
  total = int(a.gib * (1 << 30))
  piece = 8 << 20
  buf = memoryview(bytearray(os.urandom(1 << 20) * 8))
  path = os.path.join(a.dir, "mi355q_write_probe.bin")
  for threads in (1, 4, 8, 16):
    for mode in ("fresh", "fallocated", "in_place"):
      if mode != "in_place" and os.path.exists(path):
        os.remove(path)
      fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
      alloc_s = None
      if mode == "fallocated":
        t0 = time.perf_counter()
        os.posix_fallocate(fd, 0, total)
        alloc_s = time.perf_counter() - t0
      s = write_all(fd, total, piece, threads, buf)
      os.close(fd)
      print(json.dumps({"dir": a.dir, "bytes": total, "threads": threads, "mode": mode, "write_s": round(s, 4),
                        "GBps": round(total / s / 1e9, 2), "fallocate_s": None if alloc_s is None else round(alloc_s, 4)}), flush=True)
    os.remove(path)
    import mmap
    for mode in ("mmap_fresh", "mmap_fallocated", "mmap_in_place"):
      if mode != "mmap_in_place" and os.path.exists(path):
        os.remove(path)
      fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
      alloc_s = None
      t0 = time.perf_counter()
      if mode == "mmap_fallocated":
        os.posix_fallocate(fd, 0, total)
        alloc_s = time.perf_counter() - t0
      else:
        os.ftruncate(fd, total)
      mm = mmap.mmap(fd, total)
      s_ = copy_all(mm, total, piece, threads, buf)
      del mm
      os.close(fd)
      print(json.dumps({"dir": a.dir, "bytes": total, "threads": threads, "mode": mode, "write_s": round(s_, 4),
                        "GBps": round(total / s_ / 1e9, 2), "fallocate_s": None if alloc_s is None else round(alloc_s, 4)}), flush=True)
    os.remove(path)
  # unlink of a 1 GB file (what replacing the previous output costs)
  fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
  os.posix_fallocate(fd, 0, total)
  os.close(fd)
  t0 = time.perf_counter()
  os.remove(path)
  print(json.dumps({"unlink_s": round(time.perf_counter() - t0, 4)}))


if __name__ == "__main__":
  main()
