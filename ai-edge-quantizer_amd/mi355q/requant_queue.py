"""Model-level batching of the per-op weight loop (ref: params_generator.py:110-183).

The reference walks the ops of a model and quantizes one weight per call of
`get_tensor_quant_params`. On MI355X one 4096 x 4096 buffer is 14 us of kernel time: a launch
per tensor leaves the part ramping up and draining most of the time (63 % of the HBM roofline),
and a scale read-back per tensor adds a host synchronisation to every call (22 %). Inside
`batching()` the fused symmetric min/max path therefore only *enqueues* its tensor; equally
shaped tensors leave together through mi355q_requant_sym_f32_batched (one launch over the whole
group, 76 % of the roofline at C2) and all scales of a flush come back in ONE device-to-host copy.

What a caller sees does not change:
  * `UniformQuantParams.scale` is a float32 ndarray once the queue has been flushed (the queue
    swaps the placeholder for the array); before that the placeholder behaves like one (any read
    flushes first);
  * `quantized_data` is a `runtime.HbmArray` as for every large weight: int8 values on demand, the
    bytes the model file stores (`.packed` for sub-byte types) copied straight from HBM by the
    writer. For int4 / int2 targets the launch writes ONLY the packed bytes (4.5 instead of 5.5
    bytes of traffic per weight); the int8 containers are unpacked from them if somebody looks.

`ParamsGenerator.generate_quantization_parameters` runs inside `batching()`; a direct call of
`get_tensor_quant_params` outside of it computes immediately as before.
"""
from __future__ import annotations

import contextlib
import os
from typing import Callable, Optional

import numpy as np
import torch

from . import _ffi
from . import runtime as rt

# Pending FP32 input bytes after which the queue flushes on its own (a model larger than this
# goes out in several waves; 288 GB of HBM hold the inputs and outputs of a wave many times over).
DEFAULT_BUDGET_BYTES = int(os.environ.get("MI355Q_BATCH_BYTES", 48 << 30))
# ... and the number of pending tensors: the GPU starts on a wave while the host walks on
DEFAULT_BUDGET_TENSORS = int(os.environ.get("MI355Q_BATCH_TENSORS", 256))

_TORCH_OF = {np.dtype(np.int8): torch.int8, np.dtype(np.uint8): torch.uint8,
             np.dtype(np.float32): torch.float32}


class PendingArray(rt.HbmArray):
  """An HbmArray whose device tensor does not exist yet: shape and dtype are known, the values
  arrive when `resolve` runs (a queue flush, or the unpacking of packed bytes)."""

  def __init__(self, shape, dtype, resolve: Callable[[], None]):  # pylint: disable=super-init-not-called
    self._shape = tuple(int(d) for d in shape)
    self._dtype = np.dtype(dtype)
    self._resolve = resolve
    self._tensor = None
    self._host = None
    self.cache = {}
    self.packed = None

  @property
  def device_tensor(self) -> torch.Tensor:
    if self._tensor is None:
      self._resolve()
      if self._tensor is None:
        raise RuntimeError("a pending result was not produced by its flush")
    return self._tensor

  @device_tensor.setter
  def device_tensor(self, value) -> None:
    self._tensor = value

  def fill(self, tensor: torch.Tensor, host: Optional[np.ndarray] = None) -> None:
    self._tensor = tensor.reshape(self._shape)
    if host is not None:
      self._host = host.reshape(self._shape)
    self._resolve = None

  @property
  def resolved(self) -> bool:
    return self._tensor is not None

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
    return int(np.prod(self._shape, dtype=np.int64))

  @property
  def nbytes(self) -> int:
    return self.size * self._dtype.itemsize

  def numpy(self) -> np.ndarray:
    if self._host is None:
      _ = self.device_tensor      # resolve (a flush may fill the host copy directly)
    return super().numpy()

  def __repr__(self):
    state = "resolved" if self.resolved else "pending"
    return f"PendingArray(shape={self._shape}, dtype={self._dtype}, {state})"


class _Slot:
  __slots__ = ("x", "scale", "q", "packed", "params", "small")

  def __init__(self, x, scale, q, packed, small):
    self.x, self.scale, self.q, self.packed, self.small = x, scale, q, packed, small
    self.params = None


class RequantQueue:
  """Collects fused symmetric requantization requests and issues them group by group."""

  def __init__(self, budget_bytes: int = DEFAULT_BUDGET_BYTES,
               budget_tensors: int = DEFAULT_BUDGET_TENSORS):
    self.budget_bytes = budget_bytes
    self.budget_tensors = budget_tensors
    self._groups: dict[tuple, list[_Slot]] = {}
    self._pending_bytes = 0
    self._pending = 0
    self.stats = {"tensors": 0, "launches": 0, "flushes": 0, "scale_copies": 0}

  # ------------------------------------------------------------------------------- submit
  def submit(self, tensor_content, layout, num_bits: int, scale_shape, packable: bool):
    """Enqueues one [rows, cols] weight; returns (scale, quantized_data) placeholders."""
    rows, cols, block = layout
    x = rt.to_device(tensor_content.reshape(rows, cols))
    if x.data_ptr() % 16:
      x = x.clone()                      # the batched kernel takes 16-byte aligned buffers
    shape = tuple(tensor_content.shape)
    sub_byte = packable and num_bits in (2, 4)
    scale = PendingArray(scale_shape, np.float32, self.flush)
    packed = None
    if sub_byte:
      packed = PendingArray((rows * cols * num_bits // 8,), np.uint8, self.flush)
      q = PendingArray(shape, np.int8, None)
      q._resolve = _unpacker(q, packed, rows * cols, num_bits)   # pylint: disable=protected-access
      q.packed = packed
    else:
      q = PendingArray(shape, np.int8, self.flush)
    nbytes = rows * cols * 4
    slot = _Slot(x, scale, q, packed, nbytes < rt.KEEP_IN_HBM_BYTES)
    self._groups.setdefault((rows, cols, block, num_bits, sub_byte), []).append(slot)
    self._pending_bytes += nbytes
    self._pending += 1
    return scale, q, slot

  def attach(self, slot: _Slot, params) -> None:
    """`params.scale` (a frozen dataclass field holding the placeholder) becomes the float32
    ndarray itself when the slot's group has run."""
    slot.params = params
    if self._pending_bytes >= self.budget_bytes or self._pending >= self.budget_tensors:
      self.flush()

  # -------------------------------------------------------------------------------- flush
  def flush(self) -> None:
    if not self._groups:
      return
    groups, self._groups = self._groups, {}
    self._pending_bytes = self._pending = 0
    self.stats["flushes"] += 1
    L = _ffi.lib()
    dev = rt.device()
    issued = []
    for (rows, cols, block, bits, sub_byte), slots in groups.items():
      n = len(slots)
      nscale = rows * (cols // block) if block else rows
      out_bytes = rows * cols * bits // 8 if sub_byte else rows * cols
      scale_all = torch.empty((n, nscale), dtype=torch.float32, device=dev)
      out_all = torch.empty((n, out_bytes), dtype=torch.uint8 if sub_byte else torch.int8, device=dev)
      # one H2D for the three pointer tables of the group
      table = np.empty((3, n), np.int64)
      table[0] = [s.x.data_ptr() for s in slots]
      table[1] = out_all.data_ptr() + np.arange(n, dtype=np.int64) * out_bytes
      table[2] = scale_all.data_ptr() + np.arange(n, dtype=np.int64) * (nscale * 4)
      tab = torch.from_numpy(table).to(dev)
      base = tab.data_ptr()
      for first in range(0, n, 65535):       # blockIdx.y limit of one launch
        cnt = min(65535, n - first)
        ptr = lambda row: rt.ctypes.c_void_p(base + (row * n + first) * 8)   # noqa: E731
        _ffi.check(L.mi355q_requant_sym_f32_batched(
            ptr(0), cnt, rows, cols, block, bits, None if sub_byte else ptr(1),
            ptr(1) if sub_byte else None, ptr(2), None, rt.stream_ptr()))
        self.stats["launches"] += 1
      self.stats["tensors"] += n
      issued.append((slots, scale_all, out_all, sub_byte, tab))
    # every scale of the flush in one device-to-host copy (the only synchronisation)
    flat = torch.cat([g[1].reshape(-1) for g in issued]) if len(issued) > 1 else issued[0][1].reshape(-1)
    host_scales = flat.cpu().numpy()
    self.stats["scale_copies"] += 1
    pos = 0
    for slots, scale_all, out_all, sub_byte, _ in issued:
      nscale = scale_all.shape[1]
      host_out = None
      if all(s.small for s in slots):        # small results: one copy for the whole group
        host_out = out_all.cpu().numpy()
      for i, s in enumerate(slots):
        h = host_scales[pos:pos + nscale]
        pos += nscale
        s.scale.fill(scale_all[i], h)
        (s.packed if sub_byte else s.q).fill(out_all[i], None if host_out is None else host_out[i])
        if s.params is not None and s.params.scale is s.scale:
          object.__setattr__(s.params, "scale", s.scale._host)   # pylint: disable=protected-access
        s.x = None                            # the FP32 copy in HBM is no longer needed


def _unpacker(q: PendingArray, packed: PendingArray, n: int, bits: int):
  def resolve():
    from . import ops
    q.fill(ops.unpack_bits(packed.device_tensor, n, bits))
  return resolve


_ACTIVE: Optional[RequantQueue] = None
# False: `batching()` is a no-op and every tensor is quantized by its own launch (the behaviour
# the batched path is tested against; also MI355Q_NO_BATCH=1)
ENABLED = os.environ.get("MI355Q_NO_BATCH", "") in ("", "0")


def active() -> Optional[RequantQueue]:
  return _ACTIVE


@contextlib.contextmanager
def batching(budget_bytes: int = DEFAULT_BUDGET_BYTES, budget_tensors: int = DEFAULT_BUDGET_TENSORS):
  """Defers fused requantization launches issued inside the block; flushes on exit."""
  global _ACTIVE
  if _ACTIVE is not None:                     # nested use joins the outer queue
    yield _ACTIVE
    return
  queue = RequantQueue(budget_bytes, budget_tensors)
  if not ENABLED:
    yield queue
    return
  _ACTIVE = queue
  try:
    yield queue
    queue.flush()
  finally:
    _ACTIVE = None
