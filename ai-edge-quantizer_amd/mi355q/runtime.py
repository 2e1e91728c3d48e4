"""Host staging for the C ABI: torch-ROCm owns device memory and streams.

Nothing here computes; it moves NumPy buffers to HBM and back and hands raw
device pointers + the current HIP stream to libmi355q.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional
import warnings
import weakref

import numpy as np
import torch

from . import _ffi


_PREPARED: set = set()

# MI355Q_TIMELINE=1: wall-clock marks at the phase boundaries of a whole-model call (tools/c5_model.py prints them).
# `sync=True` marks wait for the GPU first, so the difference of two such marks is wall time with the GPU drained.
TIMELINE: list = []
_TIMELINE_ON = bool(os.environ.get("MI355Q_TIMELINE"))


def mark(label: str, sync: bool = False) -> None:
  if not _TIMELINE_ON:
    return
  import time
  t_host = time.perf_counter()
  if sync and torch.cuda.is_available():
    torch.cuda.synchronize()
  TIMELINE.append((label, t_host, time.perf_counter(), torch.cuda.memory_stats().get("num_device_alloc", 0) if torch.cuda.is_available() else 0))


def require_gpu() -> None:
  # (called by every op wrapper: 17 000 times in an 18-layer GPTQ calibration -- the prepared device is the fast path)
  if _PREPARED and torch.cuda.current_device() in _PREPARED:
    return
  if not torch.cuda.is_available():
    raise RuntimeError(
        "mi355q needs an AMD GPU (torch.cuda.is_available() is False); the product"
        " path has no CPU fallback.")
  dev = torch.cuda.current_device()
  if dev not in _PREPARED:
    # the library's look-ahead stream gets its hardware queue before this process creates stream pools
    # (mi355q_prepare_device in include/mi355q.h: 56 against 76 ms for a d = 16384 Hessian inverse)
    _ffi.check(_ffi.lib().mi355q_prepare_device())
    _PREPARED.add(dev)


def device() -> torch.device:
  return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> ctypes.c_void_p:
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> ctypes.c_void_p:
  return ctypes.c_void_p(0 if t is None else t.data_ptr())


def to_device(a, dtype=None) -> torch.Tensor:
  """NumPy (possibly a read-only mmap view) or torch tensor -> contiguous device tensor."""
  if isinstance(a, HbmArray):
    a = a.device_tensor
  if isinstance(a, torch.Tensor):
    t = a
    if dtype is not None and t.dtype != dtype:
      t = t.to(dtype)
    return t.to(device(), non_blocking=True).contiguous()
  arr = np.ascontiguousarray(a)
  if _FILE_MAPPINGS and (arr.nbytes >= _UPLOAD_MIN_TENSOR_BYTES or announced(arr)) and _file_range_of(arr) is not None:
    t = upload_overlapped(arr)        # a weight inside a large mapped model file: read + copied by the io ring
    return t if dtype is None or t.dtype == dtype else t.to(dtype)
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")  # non-writable buffer warning for mmap views
    t = torch.from_numpy(arr)
  if dtype is not None and t.dtype != dtype:
    t = t.to(dtype)
  return t.to(device(), non_blocking=True)


# ---- uploads that do not wait for the compute queue, downloads that do not fault the output in ------
# (the transfers themselves are csrc/file_io.hip: pread / pwrite on the library's io threads through a ring of
# three pinned 8 MiB slots -- page-locking costs ~1 ms per MiB here, once per process)
_UPLOAD_MIN_FILE_BYTES = 1 << 30  # only models large enough to earn the ring's 24 ms back
_UPLOAD_MIN_TENSOR_BYTES = 4 << 20
_COPY_STREAMS: dict = {}   # device index -> the upload stream of the io ring
_DOWNLOAD_STREAMS: dict = {}   # device index -> the download stream (the two directions do not queue behind each other)
_DOWNLOADS_IN_FLIGHT: list = []    # (payload tensor, ready event) of submitted downloads, kept until finish_downloads()
_FILE_MAPPINGS: list = []  # _FileMapping records of model files mapped by tfl_flatbuffer_utils
_OUT_MAPPINGS: list = []   # (base address, length, file descriptor) of output files being built
_OUT_PAGES_EXIST: dict = {}   # base address -> bytes from the start of the file whose pages are allocated and mapped in


class _FileMapping:
  """A mapped model file as the upload path knows it. The record lives exactly as long as the
  mapping does: it holds a weak reference to the mmap object whose finalizer removes the record and
  closes the descriptor, so an address range that the allocator hands out again after the munmap
  can never be mistaken for the file. The descriptor is a dup() of the one the mapping was made
  from -- the same inode whatever happens to the path afterwards."""
  __slots__ = ("base", "length", "fd", "ref", "__weakref__")

  def __init__(self, mapping, fd: int):
    self.base = np.frombuffer(mapping, dtype=np.uint8).ctypes.data
    self.length = len(mapping)
    self.fd = os.dup(fd)
    self.ref = weakref.ref(mapping, self._gone)

  def _gone(self, _ref=None) -> None:
    try:
      _FILE_MAPPINGS.remove(self)
    except ValueError:
      pass
    fd, self.fd = self.fd, -1
    if fd >= 0:
      try:
        os.close(fd)
      except OSError:
        pass

  def alive(self) -> bool:
    m = self.ref()
    return m is not None and not getattr(m, "closed", False) and self.fd >= 0


def register_file_mapping(mapping, fd: int) -> None:
  """Remembers where a model file is mapped, so that a weight that is a view of the mapping can also
  be fetched with pread() from the file itself (upload_overlapped). `fd`: the descriptor the
  mapping was made from (duplicated here; the caller may close its own)."""
  try:
    rec = _FileMapping(mapping, fd)
  except (ValueError, TypeError, OSError):
    return
  for old in [m for m in _FILE_MAPPINGS if m.base == rec.base or not m.alive()]:
    old._gone()  # pylint: disable=protected-access
  for old in _FILE_MAPPINGS[:-15]:
    old._gone()  # pylint: disable=protected-access
  _FILE_MAPPINGS.append(rec)


def register_output_mapping(mapping, fd: int, pages_exist: int = 0) -> None:
  """The writable mapping of an output file and its descriptor: device-resident buffers then reach
  the file by pwrite() from pinned staging (HbmArray.copy_into) instead of a pageable copy that
  faults every fresh page of the mapping in on one thread. `pages_exist`: the first so many bytes of the file have
  their pages already (allocated ahead of time and mapped in, LiteRTLMFile.prepare_output): payloads that lie inside are
  copied into the mapping by the io threads instead of pwritten (mi355q_file_io_submit_download_mapped)."""
  base = np.frombuffer(mapping, dtype=np.uint8).ctypes.data
  _OUT_MAPPINGS[:] = [m for m in _OUT_MAPPINGS if m[0] != base][-3:]
  _OUT_MAPPINGS.append((base, len(mapping), fd))
  _OUT_PAGES_EXIST[base] = int(pages_exist)
  for k in [k for k in _OUT_PAGES_EXIST if all(k != m[0] for m in _OUT_MAPPINGS)]:
    del _OUT_PAGES_EXIST[k]


def forget_output_mapping(mapping) -> None:
  try:
    base = np.frombuffer(mapping, dtype=np.uint8).ctypes.data
  except (ValueError, TypeError):
    return
  _OUT_MAPPINGS[:] = [m for m in _OUT_MAPPINGS if m[0] != base]
  _OUT_PAGES_EXIST.pop(base, None)


def _backing_mapping(arr):
  """The object at the end of `arr`'s chain of bases (ndarray.base / memoryview.obj): for a weight
  read from a mapped model file, the mmap itself."""
  obj = arr
  for _ in range(16):
    nxt = obj.base if isinstance(obj, np.ndarray) else obj.obj if isinstance(obj, memoryview) else None
    if nxt is None:
      return obj
    obj = nxt
  return obj


def _file_range_of(arr: np.ndarray):
  """(descriptor, file offset) when `arr` is a view of a registered, still mapped model file of at
  least 1 GiB; None otherwise. Both the address range AND the array's ownership chain must name
  the mapping: an array that merely landed in addresses a dead mapping used to occupy is copied
  as the ordinary array it is."""
  addr = arr.ctypes.data
  for rec in reversed(_FILE_MAPPINGS):
    if rec.base <= addr and addr + arr.nbytes <= rec.base + rec.length:
      if not rec.alive() or _backing_mapping(arr) is not rec.ref():
        return None
      return (rec.fd, addr - rec.base) if rec.length >= _UPLOAD_MIN_FILE_BYTES else None
  return None


def _copy_stream(dev) -> "torch.cuda.Stream":
  st = _COPY_STREAMS.get(dev.index)
  if st is None:
    st = _COPY_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
  return st


def _download_stream(dev) -> "torch.cuda.Stream":
  st = _DOWNLOAD_STREAMS.get(dev.index)
  if st is None:
    st = _DOWNLOAD_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
  return st


# ---- weights on their way to HBM before anybody asks for them ------------------------------------------------------
# mi355q_file_to_device holds its caller until the file has been read; a model's op walk -- the thread that launches
# the kernels -- stood still for 0.62 s of a Gemma-2B GPTQ run that way. prefetch_uploads() hands the weights a plan
# will read to the library's upload thread (mi355q_file_io_submit_upload) in plan order; upload_overlapped() then finds
# the tensor, waits only if its copies are not enqueued yet and orders the current stream behind them.
PREFETCH_WINDOW_BYTES = int(os.environ.get("MI355Q_PREFETCH_BYTES", 32 << 30))   # submitted and not yet consumed
_PREFETCHED: dict = {}     # (fd, offset, nbytes) -> (uint8 tensor, ticket)
_PREFETCH_WAITING: dict = {}   # (fd, offset, nbytes) -> None, in plan order: beyond the window, submitted as earlier ones are consumed
_PREFETCH_OUTSTANDING = [0]
_ARENA_MIN_BYTES = 64 << 20     # announced at once and at least this much: one allocation for all of it (_UploadArena)


# The upload tensors of a model are fresh device memory: ~30 ms of hipMalloc per GiB on whoever asks, and the framework's
# allocator holds its lock (and the interpreter's) meanwhile. A caller that announces a whole plan's weights at once gets ONE
# allocation for all of them, made by a helper thread through mi355q_device_alloc while the caller goes on; the uploads land in
# slices of it (torch tensors over foreign memory: __cuda_array_interface__), and the memory is given back when the last tensor
# that aliases it is gone AND release_upload_files() comes by (hipFree waits for the device: never in the middle of a walk).
_ARENA_OF: dict = {}        # announced key -> (_UploadArena, offset)
_ARENA_FREES: list = []     # (device pointer, bytes) nobody aliases any more
_ARENA_SPARE: list = []     # at most one of those, kept for the next call of the process (what a caching allocator would do:
                            # a second model of the same size starts its uploads at once); release_upload_staging() frees it


class _UploadArena:
  def __init__(self, nbytes: int):
    import threading
    self.nbytes = nbytes
    self._ptr = ctypes.c_void_p()
    self._status = None
    self._device = torch.cuda.current_device()
    if _ARENA_SPARE and _ARENA_SPARE[0][1] >= nbytes and _ARENA_SPARE[0][2] == self._device:
      spare = _ARENA_SPARE.pop()
      self._ptr.value, self.nbytes, self._status = spare[0], spare[1], 0

    def work():
      if self._status is None:
        torch.cuda.set_device(self._device)
        self._status = _ffi.lib().mi355q_device_alloc(nbytes, ctypes.byref(self._ptr))
    self._thread = threading.Thread(target=work, name="mi355q-upload-arena", daemon=True)
    self._thread.start()

  def ready(self) -> bool:
    return not self._thread.is_alive()

  def slice(self, offset: int, n: int):
    """uint8 tensor over [offset, offset + n) of the arena, or None when the allocation failed."""
    self._thread.join()
    if self._status != 0 or not self._ptr.value:
      return None
    return torch.as_tensor(_ArenaSlice(self, self._ptr.value + offset, n), device=torch.device("cuda", self._device))

  def __del__(self):
    try:
      self._thread.join()
      if self._status == 0 and self._ptr.value:
        _ARENA_FREES.append((self._ptr.value, self.nbytes, self._device))
    except Exception:  # noqa: BLE001 - interpreter exit
      pass


class _ArenaSlice:
  """What torch.as_tensor wraps (and keeps alive for as long as the tensor's storage lives)."""

  def __init__(self, arena: _UploadArena, ptr: int, n: int):
    self.arena = arena
    self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _free_dead_arenas(keep_spare: bool = True) -> None:
  """The arenas nobody aliases any more: the largest stays as the spare of the next call, the others go back (hipFree waits
  for the device: callers come here when a whole-model call is over)."""
  if not _ARENA_FREES and (keep_spare or not _ARENA_SPARE):
    return
  if torch.cuda.is_available():
    torch.cuda.synchronize()        # (whatever still read an arena's slices has run: the spare may be written again at once)
  dead = _ARENA_FREES[:] + _ARENA_SPARE[:]
  del _ARENA_FREES[:], _ARENA_SPARE[:]
  dead.sort(key=lambda e: e[1])
  if keep_spare and dead:
    _ARENA_SPARE.append(dead.pop())
  for ptr, _, _ in dead:
    _ffi.lib().mi355q_device_free(ctypes.c_void_p(ptr))


def _submit_upload(key) -> None:
  fd, offset, n = key
  dev = device()
  copy_stream = _copy_stream(dev)
  out = None
  held = _ARENA_OF.pop(key, None)
  if held is not None:
    out = held[0].slice(held[1], n)
  if out is None:
    with torch.cuda.stream(copy_stream):
      out = torch.empty((n,), dtype=torch.uint8, device=dev)
  ticket = ctypes.c_int64(0)
  _ffi.check(_ffi.lib().mi355q_file_io_submit_upload(fd, offset, n, ctypes.c_void_p(out.data_ptr()),
                                                     ctypes.c_void_p(copy_stream.cuda_stream), ctypes.byref(ticket)))
  _PREFETCHED[key] = (out, ticket.value)
  _PREFETCH_OUTSTANDING[0] += n


def _top_up_prefetch(at_most_bytes: Optional[int] = None, wait_for_arena: bool = True) -> int:
  done = 0
  while _PREFETCH_WAITING and _PREFETCH_OUTSTANDING[0] < PREFETCH_WINDOW_BYTES:
    key = next(iter(_PREFETCH_WAITING))
    held = _ARENA_OF.get(key)
    if held is None:                      # a tensor of its own: fresh memory on this thread's time, so a bounded amount per call
      if at_most_bytes is not None and done >= at_most_bytes:
        break
    elif not wait_for_arena and not held[0].ready():
      break                               # (the helper thread is still allocating: next time)
    del _PREFETCH_WAITING[key]
    _submit_upload(key)
    done += key[2]
  return done


def prefetch_uploads(arrays, submit: bool = True) -> int:
  """Announces the file-backed weights among `arrays` (views of a registered model-file mapping, 1 MiB and more) in
  the order they will be read, and (`submit`) starts their uploads without waiting for any of them; at most
  PREFETCH_WINDOW_BYTES are in HBM unconsumed at a time. With submit=False nothing starts yet: pump_prefetch() starts
  a bounded amount per call (a fresh GiB of HBM costs its caller ~30 ms of hipMalloc: a caller that has a GPU to keep
  fed spreads that over its loop). Returns the bytes announced. Whatever is not consumed is dropped by cancel_prefetch()."""
  if os.environ.get("MI355Q_NO_PREFETCH") or not torch.cuda.is_available():
    return 0
  total = 0
  fresh: list = []
  for a in arrays:
    if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or a.nbytes < (1 << 20):
      continue
    where = _file_range_of(a)
    if where is None:
      continue
    key = (where[0], where[1], a.nbytes)
    if key in _PREFETCHED or key in _PREFETCH_WAITING:
      continue
    _PREFETCH_WAITING[key] = None
    total += a.nbytes
    fresh.append(key)
  if total >= _ARENA_MIN_BYTES and total <= PREFETCH_WINDOW_BYTES and not os.environ.get("MI355Q_NO_UPLOAD_ARENA"):
    arena, at = _UploadArena(sum((k[2] + 255) & ~255 for k in fresh)), 0
    for k in fresh:
      _ARENA_OF[k] = (arena, at)
      at += (k[2] + 255) & ~255
  if submit:
    _top_up_prefetch()
  return total


def announced(a) -> bool:
  """Is `a` a weight whose upload prefetch_uploads() was told about (and nobody has consumed yet)?"""
  if not (_PREFETCHED or _PREFETCH_WAITING) or not isinstance(a, np.ndarray) or not a.flags.c_contiguous:
    return False
  where = _file_range_of(a)
  if where is None:
    return False
  key = (where[0], where[1], a.nbytes)
  return key in _PREFETCHED or key in _PREFETCH_WAITING


def pump_prefetch(at_most_bytes: int = 256 << 20) -> int:
  """Starts more of the announced uploads without holding the caller: everything whose memory the arena's helper thread has
  ready; of the uploads that need a tensor of their own, up to `at_most_bytes` (0: none of those)."""
  return _top_up_prefetch(at_most_bytes, wait_for_arena=False) if _PREFETCH_WAITING else 0


def cancel_prefetch() -> None:
  """Uploads nobody consumed: waited for (the upload thread writes into their tensors) and dropped."""
  _PREFETCH_WAITING.clear()
  _ARENA_OF.clear()
  entries = list(_PREFETCHED.values())
  _PREFETCHED.clear()
  _PREFETCH_OUTSTANDING[0] = 0
  for _, ticket in entries:
    _ffi.lib().mi355q_file_io_wait(ticket)     # (a read error of a tensor nobody reads is nobody's error)


def upload_overlapped(a: np.ndarray) -> torch.Tensor:
  """A weight that is a view of a mapped model file -> device tensor, without waiting for the compute
  already queued on the current stream.

  to_device's pageable copy is ordered behind everything queued on the current stream and blocks
  the calling thread until it ran: under GPTQ, where every FULLY_CONNECTED queues tens of
  milliseconds of inverse and update, the op walk then advances in lock step with the GPU and the
  model's 8 GB of weights cross PCIe while nothing computes; and it moves 24 GB/s (one staging
  thread inside the runtime). Here the bytes are read from the
  FILE (pread() on the library's io threads -- a kernel copy out of the page
  cache, no page-table population of the mapping -- into a ring of pinned slots) and travel on a
  copy stream of their own;
  the current stream waits for them only where it first uses the tensor. The tensor is allocated
  on the copy stream (the block it lands in cannot still be read by compute queued earlier) and
  recorded on the current one. A weight that prefetch_uploads() announced is already on its way (or there); the
  others are submitted here and waited for. Arrays that are not views of a registered mapping of at least 1 GiB
  take the pageable copy."""
  where = _file_range_of(a) if isinstance(a, np.ndarray) and a.flags.c_contiguous else None
  if where is None:
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")  # non-writable buffer warning for mmap views
      return torch.from_numpy(np.ascontiguousarray(a)).to(device(), non_blocking=True)
  key = (where[0], where[1], a.nbytes)   # (the transfer is over when mi355q_file_io_wait returns: `a` keeps the mapping, and so fd, alive)
  if key not in _PREFETCHED:
    _PREFETCH_WAITING.pop(key, None)
    _submit_upload(key)
  out, ticket = _PREFETCHED.pop(key)
  _PREFETCH_OUTSTANDING[0] -= a.nbytes
  _ffi.check(_ffi.lib().mi355q_file_io_wait(ticket))      # every copy of the transfer is enqueued on the copy stream
  done = torch.cuda.Event()
  done.record(_copy_stream(out.device))
  cur = torch.cuda.current_stream()
  cur.wait_event(done)
  out.record_stream(cur)
  _top_up_prefetch()
  t = torch.from_numpy(np.empty(0, a.dtype)).dtype
  return out.view(t).reshape(a.shape)


def _submit_download(src: torch.Tensor, fd: int, offset: int, ready=None) -> None:
  """`src` (flat uint8, device) -> fd at offset on the library's download thread. The copies wait for `ready` (an event
  recorded behind the payload's producer; default: everything queued on the current stream now), not for what
  is queued later; `src` and the event are kept until finish_downloads()."""
  if ready is None:
    ready = torch.cuda.Event()
    ready.record()
  st = _download_stream(src.device)
  _ffi.check(_ffi.lib().mi355q_file_io_submit_download(ctypes.c_void_p(src.data_ptr()), src.numel(), fd, int(offset),
                                                       ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(ready.cuda_event)))
  _DOWNLOADS_IN_FLIGHT.append((src, ready))


def _submit_download_mapped(src: torch.Tensor, address: int, ready=None) -> None:
  """`src` (flat uint8, device) -> host memory at `address` (inside an output file's mapping whose pages exist) through the
  download ring: staged in pinned slots, copied from there by the io threads. Same ordering and lifetime as _submit_download."""
  if ready is None:
    ready = torch.cuda.Event()
    ready.record()
  st = _download_stream(src.device)
  _ffi.check(_ffi.lib().mi355q_file_io_submit_download_mapped(ctypes.c_void_p(src.data_ptr()), src.numel(), ctypes.c_void_p(address),
                                                              ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(ready.cuda_event)))
  _DOWNLOADS_IN_FLIGHT.append((src, ready))


def download_into_file(t: torch.Tensor, dst: np.ndarray, ready=None) -> bool:
  """Device bytes -> the output file whose mapping `dst` is a view of (register_output_mapping):
  asynchronous copies into the pinned download ring and pwrite() from there on the io threads, driven by the
  library's download thread (mi355q_file_io_submit_download): the call returns at once. False when `dst` is not inside
  a registered output mapping (finish_downloads() before the file is handed back)."""
  addr = dst.ctypes.data
  hit = next(((base, fd) for base, length, fd in reversed(_OUT_MAPPINGS) if base <= addr and addr + dst.nbytes <= base + length), None)
  if hit is None or not t.is_cuda or dst.nbytes == 0:
    return False        # (every size goes through the download thread: a pageable copy of a 256 KB k_proj payload is ordered behind
                        # everything queued on the compute stream and held the writer's loop in lock step with the GPU)
  base, fd = hit
  flat = t.contiguous().reshape(-1).view(torch.uint8)
  if addr - base + dst.nbytes <= _OUT_PAGES_EXIST.get(base, 0) and not os.environ.get("MI355Q_DOWNLOADS_BY_PWRITE"):
    _submit_download_mapped(flat, addr, ready)
  else:
    _submit_download(flat, fd, addr - base, ready)
  return True


def write_to_file(t: torch.Tensor, fd: int, offset: int) -> None:
  """Device bytes -> `fd` at `offset` through the io ring (this rank's own descriptor of a file another rank laid out)."""
  _submit_download(t.contiguous().reshape(-1).view(torch.uint8), fd, offset)


def output_file_range_of(dst: np.ndarray):
  """(path, offset) of `dst` inside a registered output mapping, or None."""
  addr = dst.ctypes.data
  for base, length, fd in reversed(_OUT_MAPPINGS):
    if base <= addr and addr + dst.nbytes <= base + length:
      return os.readlink(f"/proc/self/fd/{fd}"), addr - base
  return None


# ---- quantized payloads that stay on the rank that made them (sharded runs that write a file) ----------------
# distributed.quantize_model_sharded gathers the per-op RESULTS on rank 0, which lays the output file out. The payloads --
# a gigabyte of quantized weights for a Gemma-2B -- need not make that trip as pickles through rank 0's host memory:
# while the results are pickled inside remote_payloads(), a device-resident payload is replaced by a RemoteBuffer (rank,
# key, sizes) and stays in its rank's HBM; rank 0's serializer "copies" a RemoteBuffer into the output mapping by noting
# the file offset it was given; afterwards every rank writes its own payloads to those offsets of the shared file through
# its io ring (distributed._write_remote_payloads).
_REMOTE_RANK: list = [None]
_REMOTE_LOCAL: dict = {}      # key -> HbmArray, on the rank that owns it
_REMOTE_WRITES: list = []     # rank 0: (rank, key, path, offset, nbytes) noted by RemoteBuffer.copy_into; path None: offset
                              # is an index into _REMOTE_HOST_SLOTS (a destination that is not inside a registered output file)
_REMOTE_HOST_SLOTS: list = [] # rank 0: writable uint8 views such payloads are copied into once their bytes have arrived
_REMOTE_SEQ = [0]
_REMOTE_KEYS: dict = {}       # id(HbmArray) -> key, while remote_payloads() is open: one array pickled twice is ONE payload


class remote_payloads:   # pylint: disable=invalid-name
  """Context: HbmArrays pickled inside it travel as RemoteBuffer records; the arrays stay registered here."""

  def __init__(self, rank: int):
    self.rank = int(rank)

  def __enter__(self):
    _REMOTE_LOCAL.clear()
    _REMOTE_KEYS.clear()
    _REMOTE_RANK[0] = self.rank
    return self

  def __exit__(self, *exc):
    _REMOTE_RANK[0] = None
    _REMOTE_KEYS.clear()


def take_remote_writes() -> tuple[list, list]:
  """(the writes noted on this rank, the host destinations the path-less ones refer to by index)."""
  out, slots = list(_REMOTE_WRITES), list(_REMOTE_HOST_SLOTS)
  _REMOTE_WRITES.clear()
  _REMOTE_HOST_SLOTS.clear()
  return out, slots


class RemoteBuffer:
  """What an HbmArray unpickles to when its bytes stayed behind (see remote_payloads)."""
  __array_priority__ = 100.0

  def __init__(self, rank: int, key: str, shape, dtype: str, nbytes: int, packed_nbytes: int):
    self.rank, self.key, self._shape, self._dtype = int(rank), key, tuple(int(v) for v in shape), np.dtype(dtype)
    self._nbytes, self._packed_nbytes = int(nbytes), int(packed_nbytes)

  shape = property(lambda self: self._shape)
  ndim = property(lambda self: len(self._shape))
  dtype = property(lambda self: self._dtype)
  nbytes = property(lambda self: self._nbytes)
  size = property(lambda self: int(np.prod(self._shape, dtype=np.int64)))

  @property
  def packed(self):
    if not self._packed_nbytes:
      return None
    return RemoteBuffer(self.rank, self.key + "/packed", (self._packed_nbytes,), "uint8", self._packed_nbytes, 0)

  def copy_into(self, dst: np.ndarray) -> None:
    if dst.nbytes != self._nbytes:
      raise RuntimeError(f"the quantized payload {self.key} of rank {self.rank} has {self._nbytes} bytes, its place {dst.nbytes}")
    where = output_file_range_of(dst)
    if where is None:
      # not a file this process laid out through a registered mapping (a caller's sink that hands out plain memory):
      # the owner sends the bytes and they are copied in before the call returns (distributed._write_remote_payloads)
      _REMOTE_HOST_SLOTS.append(dst)
      _REMOTE_WRITES.append((self.rank, self.key, None, len(_REMOTE_HOST_SLOTS) - 1, self._nbytes))
      return
    _REMOTE_WRITES.append((self.rank, self.key, where[0], int(where[1]), self._nbytes))

  def same_payload(self, other) -> bool:
    """Equality as qtyping's value comparisons need it (params_generator's sharing checks, ref params_generator.py:516-560),
    without the bytes: the SAME payload of the same rank. Ops that read one constant buffer are planned onto one rank
    (distributed.plan_op_shards), where the (buffer, config) cache hands both the same array -- pickled once, one key --
    so two records that name different payloads are different results; they are reported unequal (the conservative
    answer: the writer then keeps both)."""
    return isinstance(other, RemoteBuffer) and (self.rank, self.key) == (other.rank, other.key)

  def __eq__(self, other):
    if isinstance(other, RemoteBuffer):
      return self.same_payload(other)
    return NotImplemented

  __hash__ = object.__hash__

  def __array__(self, dtype=None, copy=None):
    raise RuntimeError(f"the quantized payload {self.key} lives in the HBM of rank {self.rank} (sharded run writing a file)")

  def __len__(self) -> int:
    return self._shape[0]

  def __reduce__(self):
    return (RemoteBuffer, (self.rank, self.key, self._shape, str(self._dtype), self._nbytes, self._packed_nbytes))

  def __repr__(self):
    return f"RemoteBuffer(rank={self.rank}, key={self.key}, shape={self._shape}, dtype={self._dtype})"


def finish_downloads() -> None:
  """Waits for the pwrite()s of download_into_file (before the output file is handed back)."""
  if _COPY_STREAMS or _DOWNLOAD_STREAMS:
    try:
      _ffi.check(_ffi.lib().mi355q_file_io_finish())
    finally:
      del _DOWNLOADS_IN_FLIGHT[:]


def release_upload_files() -> None:
  """Waits for the uploads in flight. (The descriptors belong to the mappings they were
  duplicated for and are closed when those are unmapped: _FileMapping.)"""
  cancel_prefetch()
  for st in _COPY_STREAMS.values():
    st.synchronize()
  _free_dead_arenas()
  for rec in [m for m in _FILE_MAPPINGS if not m.alive()]:
    rec._gone()  # pylint: disable=protected-access


def release_upload_staging() -> None:
  """Closes the files; the pinned ring (24 MB of page-locked host memory per device) and the io
  threads go with mi355q_shutdown()."""
  release_upload_files()
  _free_dead_arenas(keep_spare=False)
  _COPY_STREAMS.clear()
  _DOWNLOAD_STREAMS.clear()


_NP_DTYPE: dict = {}     # torch dtype -> NumPy dtype


class LateVector:
  """The flat view of a device-resident result (per-channel scales) whose VALUES the flatbuffer writer may take last:
  size and dtype are known now, `np.asarray()` waits for the values (utils/tflite_flatbuffer.py: `late_values`).
  Every other consumer sees an ndarray-like: reading it completes the producer first, as reading the HbmArray would."""
  late_values = True

  def __init__(self, source: "HbmArray"):
    self._source = source
    self.dtype = source.dtype
    self.size = source.size
    self.nbytes = source.size * source.dtype.itemsize
    self.shape = (self.size,)
    self.ndim = 1

  def __array__(self, dtype=None, copy=None):
    a = np.ravel(np.asarray(self._source))
    return a if dtype is None or a.dtype == dtype else a.astype(dtype)

  def __len__(self) -> int:
    return self.size

  def __getitem__(self, idx):
    return np.asarray(self)[idx]

  def __iter__(self):
    return iter(np.asarray(self))

  def tolist(self):
    return np.asarray(self).tolist()

  def __eq__(self, other):
    return np.asarray(self) == other

  def __ne__(self, other):
    return np.asarray(self) != other

  __hash__ = None

  def __getattr__(self, name):      # anything else NumPy offers: on the values
    if name.startswith("_"):
      raise AttributeError(name)
    return getattr(np.asarray(self), name)

  def __repr__(self):
    return f"LateVector(size={self.size}, dtype={self.dtype})"


_LATE_CONSTANTS = [0]      # > 0 while a writer that verifies them is at work (model_modifier.ModelModifier.modify_model)


def late_constants_allowed() -> bool:
  return _LATE_CONSTANTS[0] > 0


def late_vector(values, dtype):
  """`values` as the flat `dtype` vector a flatbuffer table stores: a LateVector when they are still in HBM with that
  dtype and nobody has read them on the host, else the ndarray."""
  if isinstance(values, HbmArray) and getattr(values, "_host", None) is None and values.dtype == dtype:
    return LateVector(values)
  return np.ravel(values).astype(dtype, copy=False)


class HbmArray:
  """A result that lives in HBM and reaches the host only if somebody asks for it.

  QSV entries are `Any` in the reference's interface (qtyping.QSV); a GPTQ Hessian is d x d
  float64 (2 GiB at d = 16384) that the calibration loop merges once per sample and the weight
  update consumes on the GPU again, so copying it out and back for every step would make
  PCIe the bottleneck of the whole algorithm. NumPy consumers still work: `np.asarray(h)`,
  `h.shape`, `h.dtype`, indexing and arithmetic all go through a cached host copy made on
  first use.
  """
  __array_priority__ = 100.0
  ready = None          # torch.cuda.Event recorded behind the kernels that produced the values, when the producer offers one

  def __init__(self, tensor: torch.Tensor):
    self.device_tensor = tensor
    self._host = None
    self.cache: dict = {}          # derived device results (e.g. the damped inverse)
    self.packed = None             # quantized weights: the packed bytes of the same launch

  @property
  def nbytes(self) -> int:
    return self.device_tensor.numel() * self.device_tensor.element_size()

  def copy_into(self, dst: np.ndarray) -> None:
    """D2H straight into `dst` (a writable uint8 view of, e.g., the output file's mapping)."""
    if self._host is not None:
      dst[:] = np.ravel(self._host).view(np.uint8)
      return
    t = self.device_tensor                  # (a placeholder's last launch leaves here and sets `ready`)
    ready = getattr(self, "ready", None)    # the event behind this payload's own producer, when the producer recorded one
    if download_into_file(t, dst) if ready is None else download_into_file(t, dst, ready):
      return
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      torch.from_numpy(dst).copy_(self.device_tensor.contiguous().reshape(-1).view(torch.uint8))

  @property
  def shape(self):
    return tuple(self.device_tensor.shape)

  @property
  def ndim(self) -> int:
    return self.device_tensor.dim()

  @property
  def dtype(self):
    t = self.device_tensor.dtype
    d = _NP_DTYPE.get(t)
    if d is None:
      d = _NP_DTYPE[t] = np.dtype(str(t).replace("torch.", ""))
    return d

  @property
  def size(self) -> int:
    return self.device_tensor.numel()

  def reshape(self, *shape):
    """Another view of the same device memory (no copy, no host visit)."""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
      shape = tuple(shape[0])
    out = HbmArray(self.device_tensor.contiguous().reshape(tuple(int(d) for d in shape)))
    if self._host is not None:
      out._host = self._host.reshape(shape)  # pylint: disable=protected-access
    return out

  def numpy(self) -> np.ndarray:
    if self._host is None:
      self._host = self.device_tensor.cpu().numpy()
    return self._host

  def __array__(self, dtype=None, copy=None):
    a = self.numpy()
    return a if dtype is None else a.astype(dtype, copy=False)

  def __len__(self) -> int:
    return self.shape[0]

  def __getitem__(self, idx):
    return self.numpy()[idx]

  def tolist(self):
    return self.numpy().tolist()

  def __getattr__(self, name):
    # anything else an ndarray offers (.T, .reshape, .sum, ...) is answered by the host copy
    if name.startswith("__"):
      raise AttributeError(name)
    return getattr(self.numpy(), name)

  def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
    host = [x.numpy() if isinstance(x, HbmArray) else x for x in inputs]
    return getattr(ufunc, method)(*host, **kwargs)

  def __repr__(self):
    return f"HbmArray(shape={self.shape}, dtype={self.dtype})"

  def __reduce__(self):
    # crosses process boundaries (gather of sharded results) as host data. A sub-byte quantized
    # weight travels as the packed bytes the model file stores -- a third of what the int8
    # containers plus the packed bytes would be; the containers are unpacked on arrival only if read.
    # (integer payloads only: what the serializer copies into the file byte for byte. Blockwise scales, float32 arrays the
    # transformation layer reads as values, travel as host data like before.)
    # From 256 KiB on: one such payload alone puts the model over the serializer's inline limit (model_modifier), so a
    # RemoteBuffer always meets the external-buffer layout, which copies payloads with copy_into().
    if _REMOTE_RANK[0] is not None and self.nbytes >= (1 << 18) and self.dtype in (np.int8, np.uint8):
      key = _REMOTE_KEYS.get(id(self))
      if key is None:             # (an array two results share -- a weight two ops read -- is one payload under one key)
        _REMOTE_SEQ[0] += 1
        key = _REMOTE_KEYS[id(self)] = f"r{_REMOTE_RANK[0]}/{_REMOTE_SEQ[0]}"
      _REMOTE_LOCAL[key] = self
      packed_nbytes = 0
      if isinstance(self.packed, HbmArray):
        _REMOTE_LOCAL[key + "/packed"] = self.packed
        packed_nbytes = self.packed.nbytes
      return (RemoteBuffer, (_REMOTE_RANK[0], key, self.shape, str(self.dtype), self.nbytes, packed_nbytes))
    if self.packed is not None and self.dtype == np.int8 and self.size:
      bits = self.packed.size * 8 // self.size
      if bits in (2, 4):
        return (PackedCarrier, (np.asarray(self.packed), self.shape, bits))
    return (_host_carrier, (self.numpy(), None if self.packed is None else np.asarray(self.packed)))


class PackedCarrier:
  """What the HbmArray of an int4 / int2 quantized weight unpickles to on another rank: the packed
  bytes (`.packed`, what transformations/quantize_tensor stores) and, on demand, the sign-extended
  int8 containers (ref transformations/transformation_utils.py:293-353 read backwards)."""
  __array_priority__ = 100.0

  def __init__(self, packed: np.ndarray, shape, bits: int):
    self.packed = packed
    self._shape = tuple(int(v) for v in shape)
    self._bits = int(bits)
    self._values = None

  shape = property(lambda self: self._shape)
  ndim = property(lambda self: len(self._shape))
  dtype = property(lambda self: np.dtype(np.int8))
  size = property(lambda self: int(np.prod(self._shape, dtype=np.int64)))
  nbytes = property(lambda self: self.size)

  def numpy(self) -> np.ndarray:
    if self._values is None:
      per = 8 // self._bits
      b = np.asarray(self.packed, dtype=np.uint8)
      parts = [(b >> (self._bits * k)) & ((1 << self._bits) - 1) for k in range(per)]
      v = np.stack(parts, axis=-1).reshape(-1)[:self.size].astype(np.int8)
      half = 1 << (self._bits - 1)
      self._values = np.where(v >= half, v - (1 << self._bits), v).astype(np.int8).reshape(self._shape)
    return self._values

  def __array__(self, dtype=None, copy=None):
    a = self.numpy()
    return a if dtype is None else a.astype(dtype, copy=False)

  def __len__(self) -> int:
    return self._shape[0]

  def __getitem__(self, idx):
    return self.numpy()[idx]

  def reshape(self, *shape):
    return self.numpy().reshape(*shape)

  def tolist(self):
    return self.numpy().tolist()

  def __reduce__(self):
    return (PackedCarrier, (self.packed, self._shape, self._bits))

  def __repr__(self):
    return f"PackedCarrier(shape={self._shape}, int{self._bits})"


KEEP_IN_HBM_BYTES = 4 << 20     # quantized weights of float tensors this large stay in HBM


def quantized_result(q: torch.Tensor, num_bits: int, source_nbytes: int, shape=None):
  """What a weight algorithm hands back as `quantized_data`: a host ndarray for ordinary
  tensors; for large ones the device tensor itself (HbmArray, with the packed bytes of sub-byte
  types made by one more launch), so that the model writer copies it straight into the output
  file and NumPy consumers still get a host copy on demand."""
  if shape is not None:
    q = q.reshape(shape)
  if source_nbytes < KEEP_IN_HBM_BYTES:
    return to_numpy(q)
  out = HbmArray(q)
  if num_bits in (2, 4):
    from . import ops
    out.packed = HbmArray(ops.pack_bits(q, num_bits))
  return out


class HostCarrier(np.ndarray):
  """What an HbmArray unpickles to: a plain ndarray (+ the packed bytes that rode along)."""
  packed = None

  def __array_finalize__(self, obj):
    self.packed = None


def _host_carrier(values: np.ndarray, packed):
  out = values.view(HostCarrier)
  out.packed = packed
  return out


def _host_operator(name):
  def op(self, other):
    other = other.numpy() if isinstance(other, HbmArray) else other
    return getattr(self.numpy(), name)(other)
  op.__name__ = name
  return op


for _name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__",
              "__rtruediv__", "__matmul__", "__rmatmul__", "__eq__", "__ne__", "__lt__", "__le__",
              "__gt__", "__ge__"):
  setattr(HbmArray, _name, _host_operator(_name))
HbmArray.__hash__ = object.__hash__
HbmArray.__neg__ = lambda self: -self.numpy()
HbmArray.__abs__ = lambda self: abs(self.numpy())


def resident_sample(value):
  """A calibration sample entry as the calibrator keeps it: device tensors become HbmArray (no
  copy), host torch tensors become ndarrays, everything else is passed through."""
  if isinstance(value, torch.Tensor):
    value = value.detach()
    if value.dtype == torch.bfloat16:     # NumPy has no bfloat16; widening to float32 is exact
      value = value.float()
    return HbmArray(value) if value.is_cuda else value.numpy()
  return value


def on_device(a, dtype=None) -> torch.Tensor:
  """Device tensor of `a` without a copy when `a` already lives in HBM."""
  if isinstance(a, HbmArray):
    t = a.device_tensor
    return t if dtype is None or t.dtype == dtype else t.to(dtype)
  return to_device(a, dtype)


# ---- calibration-step staging ----------------------------------------------------------------
# While the calibrator walks the ops of one sample, every activation it will touch is already in
# HBM with its (min, max): staged once, reduced in one batched launch. Keyed by the identity of
# the host array the caller supplied (the entry keeps that array alive, so ids cannot be reused).
_STEP_STAGE: dict[int, dict] = {}


def stage_calibration_step(entries: dict[int, dict]) -> None:
  _STEP_STAGE.clear()
  _STEP_STAGE.update(entries)


def clear_calibration_step() -> None:
  _STEP_STAGE.clear()


def staged(host_array):
  """The staging record of `host_array` ({"host", "dev", "lo", "hi", "minmax"}) or None."""
  rec = _STEP_STAGE.get(id(host_array))
  return rec if rec is not None and rec["host"] is host_array else None


def empty(shape, dtype) -> torch.Tensor:
  return torch.empty(shape, dtype=dtype, device=device())


def to_numpy(t: torch.Tensor) -> np.ndarray:
  return t.cpu().numpy()


def ptr_table(tensors) -> torch.Tensor:
  """Device table of device pointers (int64) for the *_batched entry points."""
  host = torch.tensor([0 if t is None else t.data_ptr() for t in tensors], dtype=torch.int64)
  return host.to(device())
