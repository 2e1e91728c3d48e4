"""Model-level batching of the per-op weight loop (ref: params_generator.py:110-183).

The reference walks the ops of a model and quantizes one weight per call of
`get_tensor_quant_params`. On MI355X one 4096 x 4096 buffer is 14 us of kernel time: a launch
per tensor leaves the part ramping up and draining most of the time (63 % of the HBM roofline),
and a scale read-back per tensor adds a host synchronisation to every call (22 %). Inside
`batching()` the fused symmetric min/max path therefore only *enqueues* its tensor; equally
shaped tensors leave together through mi355q_requant_sym_f32_batched (one launch over the whole
group, 76 % of the roofline at C2). A flush never waits for the GPU: per-row scales come back
through ONE asynchronous copy per wave into pinned memory, completed when the block exits (or
when somebody reads a value), so the kernels of a wave run while the host walks the next ops.

What a caller sees does not change:
  * per-channel / per-tensor `UniformQuantParams.scale` is a float32 ndarray once the block has
    exited (the queue swaps the placeholder for the array); before that the placeholder behaves
    like one (a read completes the wave first). Blockwise scales (one per 32..256 weights, 1.4 MB
    per C3 layer) stay in HBM as an array-like `runtime.HbmArray` with the IEEE-half patterns the
    model file stores (`.f16`, written by the same launch) beside them; they reach the host only
    if somebody reads them;
  * `quantized_data` is a `runtime.HbmArray` as for every large weight: int8 values on demand, the
    bytes the model file stores (`.packed` for sub-byte types) copied straight from HBM by the
    writer. For int4 / int2 targets the launch writes ONLY the packed bytes (4.5 instead of 5.5
    bytes of traffic per weight); the int8 containers are unpacked from them if somebody looks.

`ParamsGenerator.generate_quantization_parameters` runs inside `batching()`; a direct call of
`get_tensor_quant_params` outside of it computes immediately as before.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Optional

import numpy as np
import torch

from . import _ffi
from . import runtime as rt

# Pending FP32 input bytes after which the queue flushes on its own (a model larger than this
# goes out in several waves; 288 GB of HBM hold the inputs and outputs of a wave many times over).
DEFAULT_BUDGET_BYTES = int(os.environ.get("MI355Q_BATCH_BYTES", 48 << 30))
# ... and the number of pending tensors: the GPU starts on a wave while the host walks on
DEFAULT_BUDGET_TENSORS = int(os.environ.get("MI355Q_BATCH_TENSORS", 64))
# A shape group that has collected this many tensors leaves at once (the batched kernel is at its
# roofline fraction with 16 buffers per launch): the GPU works on them while Python enqueues the next
# ones, instead of idling until the whole model has been walked.
GROUP_LAUNCH_TENSORS = int(os.environ.get("MI355Q_GROUP_LAUNCH_TENSORS", 16))
# The FIRST group of a walk leaves at 8 tensors (MI355Q_GROUP_RAMP, a comma-separated list of sizes for the first launches):
# until the first launch the GPU has nothing to do. tools/api_resident_timeline.py, 64 resident 4096 x 4096 weights, total us:
# no ramp 1126, "8" 1081, "4,8" 1119, "4,12" 1102, "2,6,8" 1142 -- smaller first launches run further below the roofline
# fraction of 16-tensor ones than the earlier start buys. The walk is GPU-bound behind it (the host is done at 0.49 of 1.08 ms).
GROUP_RAMP = tuple(int(v) for v in os.environ.get("MI355Q_GROUP_RAMP", "8").split(",") if v.strip())

# Pointer tables of a launch: in the kernel arguments (mi355q_requant_sym_f32_batched_hostptrs), or -- MI355Q_REQUANT_TABLES=device,
# for A / B timing -- copied to HBM in front of it (mi355q_requant_sym_f32_batched)
_HOST_TABLES = os.environ.get("MI355Q_REQUANT_TABLES", "host") != "device"

_I8, _U8, _F32, _F16 = np.dtype(np.int8), np.dtype(np.uint8), np.dtype(np.float32), np.dtype(np.float16)


class PendingArray(rt.HbmArray):
  """An HbmArray whose device tensor does not exist yet: shape and dtype are known, the values
  arrive when its wave has been issued (`source` = (group tensor, row)) or when its packed
  sibling has been unpacked."""

  # (class-level defaults: three of these are made per queued tensor, and only what differs is set)
  _source = None        # (tensor [n, ...], index) once the wave is out
  _wave = None          # _Wave whose pinned copy carries the host values (scales)
  _host_at = None       # (offset, count) into the wave's host block
  _unpack = None        # (packed sibling, element count, bits)
  _tensor = None
  _host = None
  _slot = None          # the queue slot whose launch produces the values (RequantQueue.resolve)
  packed = None
  f16 = None

  def __init__(self, shape, dtype, queue):  # pylint: disable=super-init-not-called
    self._shape = shape
    self._dtype = dtype
    self._queue = queue
    self.cache = {}

  @property
  def device_tensor(self) -> torch.Tensor:
    if self._tensor is None:
      if self._unpack is not None:
        from . import ops
        packed, n, bits = self._unpack
        self._tensor = ops.unpack_bits(packed.device_tensor, n, bits).reshape(self._shape)
      else:
        if self._source is None:
          self._queue.resolve(self._slot)
        group, i = self._source
        self._tensor = group[i].reshape(self._shape)
    return self._tensor

  @device_tensor.setter
  def device_tensor(self, value) -> None:
    self._tensor = value

  @property
  def resolved(self) -> bool:
    """Values exist on the device (issued, for queue outputs; unpacked, for int8 containers)."""
    return self._tensor is not None or self._source is not None

  @property
  def shape(self):
    return self._shape

  @property
  def ndim(self) -> int:
    return len(self._shape)

  @property
  def dtype(self):
    return self._dtype

  @property
  def size(self) -> int:
    n = 1
    for d in self._shape:
      n *= d
    return n

  @property
  def nbytes(self) -> int:
    return self.size * self._dtype.itemsize

  def numpy(self) -> np.ndarray:
    if self._host is None:
      if self._unpack is None and self._source is None:
        self._queue.resolve(self._slot)
      if self._wave is None and self._queue is not None:
        self._queue._send_scales()               # (a per-row scale of a group that left early: its copy leaves now)  # pylint: disable=protected-access
      if self._wave is not None:                 # rode along in the wave's pinned scale copy
        self._host = self._wave.host_values(*self._host_at).reshape(self._shape)
      else:
        self._host = self.device_tensor.cpu().numpy()
    return self._host

  def __repr__(self):
    state = "resolved" if self.resolved else "pending"
    return f"PendingArray(shape={self._shape}, dtype={self._dtype}, {state})"


class _Slot:
  __slots__ = ("x", "scale", "out", "params", "key", "counted")

  def __init__(self, x, scale, out, key=None):
    self.x, self.scale, self.out, self.key = x, scale, out, key
    self.params = None
    self.counted = not isinstance(x, np.ndarray)   # takes part in the wave budgets: its FP32 copy is in HBM on the queue's account


class _Wave:
  """The asynchronous device-to-host copy of one flush's per-row scales."""

  def __init__(self, device_flat: torch.Tensor):
    self._pinned = torch.empty(device_flat.shape, dtype=device_flat.dtype, pin_memory=True)
    self._pinned.copy_(device_flat, non_blocking=True)
    self._event = torch.cuda.Event()
    self._event.record()
    self._values = None
    self._landing = None
    self.slots: list[_Slot] = []

  def host_values(self, offset: int, count: int) -> np.ndarray:
    if self._values is None:
      self._event.synchronize()
      self._values = self._pinned.numpy().copy()    # ordinary memory; the pinned block is recycled
      self._pinned = None
    return self._values[offset:offset + count]

  def hand_over(self) -> None:
    """Every per-row scale of the wave becomes the ndarray the reference returns -- a view of ONE block of ordinary
    memory that land() fills. The 64 small host steps per wave (a view, a reshape, a field swap per tensor) happen HERE,
    while the wave's last launch is still running; after the GPU has finished only one wait and one copy remain
    (done the other way round, those steps were a quarter of the wall time of a 64-tensor walk: 2-3 us per tensor of
    host work with the GPU idle)."""
    if self._values is not None:                     # somebody read a scale earlier: the values are here already
      for s in self.slots:
        host = s.scale.numpy()
        if s.params is not None and s.params.scale is s.scale:
          object.__setattr__(s.params, "scale", host)
      self.slots = []
      return
    # (NaN until land() has filled it: a scale read after a failed wait must not look like one)
    self._landing = np.full(tuple(self._pinned.shape), np.nan, np.float32)
    for s in self.slots:
      off, n = s.scale._host_at                      # pylint: disable=protected-access
      host = self._landing[off:off + n].reshape(s.scale._shape)   # pylint: disable=protected-access
      s.scale._host = host                           # pylint: disable=protected-access
      if s.params is not None and s.params.scale is s.scale:
        object.__setattr__(s.params, "scale", host)
    self.slots = []

  def land(self) -> None:
    """The wave's copy has arrived: its values fill the block the scales are views of."""
    if self._landing is not None:
      if self._values is not None:      # host_values() was asked between hand_over() and here: the values are its copy
        np.copyto(self._landing, self._values)
      else:
        self._event.synchronize()
        np.copyto(self._landing, self._pinned.numpy())
      self._values, self._landing, self._pinned = self._landing, None, None

  def complete(self) -> None:
    self.hand_over()
    self.land()


class RequantQueue:
  """Collects fused symmetric requantization requests and issues them group by group."""

  def __init__(self, budget_bytes: int = DEFAULT_BUDGET_BYTES,
               budget_tensors: int = DEFAULT_BUDGET_TENSORS):
    self.budget_bytes = budget_bytes
    self.budget_tensors = budget_tensors
    self._groups: dict[tuple, list[_Slot]] = {}
    self._pending_bytes = 0
    self._pending = 0
    self._waves: list[_Wave] = []
    self._row_scales: list = []         # (slots, scale tensor) of launches whose per-row scales have not been sent to the host yet
    self._deferred: list = []           # other algorithms' own queues: completed with this one
    self.stats = {"tensors": 0, "launches": 0, "flushes": 0, "scale_copies": 0}

  def defer(self, complete) -> None:
    """`complete()` runs when the block exits (before the waits): an algorithm that queues work of
    its own inside `batching()` (GPTQ's applies that share a Hessian inverse) finishes there."""
    self._deferred.append(complete)

  # ------------------------------------------------------------------------------- submit
  def submit(self, tensor_content, layout, num_bits: int, scale_shape, packable: bool):
    """Enqueues one [rows, cols] weight; returns (scale, quantized_data, slot) placeholders."""
    rows, cols, block = layout
    if isinstance(tensor_content, rt.HbmArray):
      x = tensor_content.device_tensor
      if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    elif rt.announced(tensor_content):
      x = tensor_content                 # on its way to HBM already (runtime.prefetch_uploads): met again where it is launched
    else:
      x = rt.to_device(tensor_content)
    if not isinstance(x, np.ndarray) and x.data_ptr() % 16:
      x = x.clone()                      # the batched kernel takes 16-byte aligned buffers
    shape = tuple(tensor_content.shape)
    scale = PendingArray(tuple(scale_shape), _F32, self)
    if block:
      scale.f16 = PendingArray(scale.shape, _F16, self)
    if packable and num_bits != 8:
      out = PendingArray((rows * cols * num_bits // 8,), _U8, self)
      q = PendingArray(shape, _I8, self)
      q._unpack = (out, rows * cols, num_bits)   # pylint: disable=protected-access
      q.packed = out
      sub_byte = True
    else:
      q = out = PendingArray(shape, _I8, self)
      sub_byte = False
    key = (rows, cols, block, num_bits, sub_byte)
    slot = _Slot(x, scale, out, key)
    scale._slot = out._slot = q._slot = slot     # pylint: disable=protected-access
    if block:
      scale.f16._slot = slot                     # pylint: disable=protected-access
    group = self._groups.get(key)
    if group is None:
      group = self._groups[key] = []
    group.append(slot)
    if slot.counted:      # (a weight still arriving from the model file is on the prefetch window's account, runtime.prefetch_uploads)
      self._pending_bytes += rows * cols * 4
      self._pending += 1
    return scale, q, slot

  def attach(self, slot: _Slot, params) -> None:
    """`params.scale` (a frozen dataclass field holding the placeholder) becomes the float32
    ndarray itself when the slot's wave has completed (per-row scales)."""
    slot.params = params
    if self._pending >= self.budget_tensors or self._pending_bytes >= self.budget_bytes:
      self.flush()
      return
    group = self._groups.get(slot.key)
    # (a group whose weights are still arriving from the model file does not leave at 16: its launch would hold the walk until
    # those uploads are in, and the writer that follows the walk launches them as they arrive, payload by payload: resolve())
    ramp = self.stats["launches"]
    leave_at = GROUP_RAMP[ramp] if ramp < len(GROUP_RAMP) and _HOST_TABLES else GROUP_LAUNCH_TENSORS
    if group is not None and len(group) >= min(leave_at, GROUP_LAUNCH_TENSORS) and not isinstance(slot.x, np.ndarray):
      del self._groups[slot.key]
      rows, cols = slot.key[0], slot.key[1]
      counted = sum(1 for s in group if s.counted)
      self._pending -= counted
      self._pending_bytes -= counted * rows * cols * 4
      self._launch({slot.key: group}, send_scales=False)

  # ------------------------------------------------------------------------------ resolve
  def resolve(self, slot) -> None:
    """Somebody needs the values of `slot` now. Tensors already in HBM leave all together (flush). A group whose
    sources are still arriving from the model file (submit kept the host view: the upload thread is at work) sends
    only the slots up to and including this one -- uploads arrive in the order they were announced, so those are
    there or next -- and the later ones keep waiting: the writer asks payload by payload, in that same order, and
    each one's bytes start for the output file while the following weights are still on their way in."""
    group = self._groups.get(slot.key) if slot is not None else None
    at = next((i for i, s in enumerate(group) if s is slot), None) if group else None
    if at is None or not any(isinstance(s.x, np.ndarray) for s in group[:at + 1]):
      self.flush()
      return
    head, rest = group[:at + 1], group[at + 1:]
    if rest:
      self._groups[slot.key] = rest
    else:
      del self._groups[slot.key]
    rows, cols = slot.key[0], slot.key[1]
    counted = sum(1 for s in head if s.counted)
    self._pending -= counted
    self._pending_bytes -= counted * rows * cols * 4
    self._launch({slot.key: head})

  # -------------------------------------------------------------------------------- flush
  def flush(self) -> None:
    """Issues everything pending: one launch per shape group, one asynchronous copy of the
    per-row scales. Does not wait for the GPU."""
    if not self._groups:
      self._send_scales()
      return
    groups, self._groups = self._groups, {}
    self._pending_bytes = self._pending = 0
    self.stats["flushes"] += 1
    self._launch(groups)

  def _launch(self, groups, send_scales: bool = True) -> None:
    L = _ffi.lib()
    dev = rt.device()
    stream = rt.stream_ptr()
    for (rows, cols, block, bits, sub_byte), slots in groups.items():
      n = len(slots)
      for s in slots:
        if isinstance(s.x, np.ndarray):        # announced weight: waits here if its copies are not enqueued yet
          s.x = rt.to_device(s.x)
          if s.x.data_ptr() % 16:
            s.x = s.x.clone()
      nscale = rows * (cols // block) if block else rows
      out_bytes = rows * cols * bits // 8 if sub_byte else rows * cols
      scale_all = torch.empty((n, nscale), dtype=torch.float32, device=dev)
      out_all = torch.empty((n, out_bytes), dtype=torch.uint8 if sub_byte else torch.int8, device=dev)
      f16_all = torch.empty((n, nscale), dtype=torch.float16, device=dev) if block else None
      steps = np.arange(n, dtype=np.int64)
      xs_at = [s.x.data_ptr() for s in slots]
      if _HOST_TABLES:
        # the tables travel in the kernel arguments (16 buffers per dispatch): no copy in front of the launch
        arr = ctypes.c_void_p * n
        outs = arr(*(out_all.data_ptr() + steps * out_bytes).tolist())
        _ffi.check(L.mi355q_requant_sym_f32_batched_hostptrs(
            arr(*xs_at), n, rows, cols, block, bits, None if sub_byte else outs, outs if sub_byte else None,
            arr(*(scale_all.data_ptr() + steps * (nscale * 4)).tolist()),
            arr(*(f16_all.data_ptr() + steps * (nscale * 2)).tolist()) if block else None, stream))
        self.stats["launches"] += -(-n // 16)
      else:
        # one H2D for the pointer tables of the group
        # (pinned staging: a pageable source would make the copy wait for the previous wave)
        pinned = torch.empty((4, n), dtype=torch.int64, pin_memory=True)
        table = pinned.numpy()
        table[0] = xs_at
        table[1] = out_all.data_ptr() + steps * out_bytes
        table[2] = scale_all.data_ptr() + steps * (nscale * 4)
        table[3] = f16_all.data_ptr() + steps * (nscale * 2) if block else 0
        tab = pinned.to(dev, non_blocking=True)
        base = tab.data_ptr()
        for first in range(0, n, 65535):       # blockIdx.y limit of one launch
          cnt = min(65535, n - first)
          ptr = lambda row: ctypes.c_void_p(base + (row * n + first) * 8)   # noqa: E731
          _ffi.check(L.mi355q_requant_sym_f32_batched(
              ptr(0), cnt, rows, cols, block, bits, None if sub_byte else ptr(1),
              ptr(1) if sub_byte else None, ptr(2), ptr(3) if block else None, stream))
          self.stats["launches"] += 1
        del tab                                # (stream-ordered allocator: freed after the launch)
      self.stats["tensors"] += n
      produced = torch.cuda.Event()            # behind this group's outputs (runtime.HbmArray.copy_into: the file writer's gate)
      produced.record()
      for i, s in enumerate(slots):
        s.scale._source = (scale_all, i)       # pylint: disable=protected-access
        s.out._source = (out_all, i)           # pylint: disable=protected-access
        s.out.ready = s.scale.ready = produced
        if block:
          s.scale.f16._source = (f16_all, i)   # pylint: disable=protected-access
          s.scale.f16.ready = produced
        s.x = None                             # the FP32 copy in HBM is no longer needed
      if not block:
        self._row_scales.append((slots, scale_all))
    if send_scales:
      self._send_scales()

  def _send_scales(self) -> None:
    """ONE asynchronous copy of the per-row scales of every launch since the last one. A group that leaves early (16
    tensors of a shape, attach()) does not send its own: a device-to-host copy between two launches holds the second
    one back on the in-order stream until the copy's completion has been signalled (~10 us per launch); the scales
    leave with the next flush, or when somebody reads one."""
    row_scales, self._row_scales = self._row_scales, []
    if not row_scales:
      return
    flat = (torch.cat([g[1].reshape(-1) for g in row_scales]) if len(row_scales) > 1
            else row_scales[0][1].reshape(-1))
    wave = _Wave(flat)
    self.stats["scale_copies"] += 1
    pos = 0
    for slots, scale_all in row_scales:
      nscale = scale_all.shape[1]
      for s in slots:
        s.scale._wave, s.scale._host_at = wave, (pos, nscale)   # pylint: disable=protected-access
        pos += nscale
      wave.slots.extend(slots)
    self._waves.append(wave)

  def finish(self) -> None:
    """Flush + wait for the scale copies + swap the placeholders of per-row scales."""
    deferred, self._deferred = self._deferred, []
    for complete in deferred:
      complete()
    self.flush()
    waves, self._waves = self._waves, []
    for w in waves:          # (the host's share first, under the last launches; then the waits)
      w.hand_over()
    for w in waves:
      w.land()


# the queue of the thread that opened `batching()`: another thread's calls neither join it nor
# have their launches issued on its stream
_LOCAL = threading.local()
# False: `batching()` is a no-op and every tensor is quantized by its own launch (the behaviour
# the batched path is tested against; also MI355Q_NO_BATCH=1)
ENABLED = os.environ.get("MI355Q_NO_BATCH", "") in ("", "0")


def active() -> Optional[RequantQueue]:
  return getattr(_LOCAL, "queue", None)


def complete_active() -> None:
  """The open block's launches out and its host values in (what leaving the block does), with the block still open: a
  writer that has handed every payload its place in the output file takes the values (per-channel scales) last."""
  queue = active()
  if queue is not None:
    queue.finish()


@contextlib.contextmanager
def batching(budget_bytes: int = DEFAULT_BUDGET_BYTES, budget_tensors: int = DEFAULT_BUDGET_TENSORS):
  """Defers fused requantization launches issued inside the block; completes them on exit."""
  outer = active()
  if outer is not None:                       # nested use joins the outer queue
    yield outer
    return
  queue = RequantQueue(budget_bytes, budget_tensors)
  if not ENABLED:
    yield queue
    return
  _LOCAL.queue = queue
  try:
    yield queue
    queue.finish()
  finally:
    _LOCAL.queue = None
