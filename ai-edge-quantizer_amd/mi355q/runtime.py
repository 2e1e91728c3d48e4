"""Host staging for the C ABI: torch-ROCm owns device memory and streams.

Nothing here computes; it moves NumPy buffers to HBM and back and hands raw
device pointers + the current HIP stream to libmi355q.
"""
from __future__ import annotations

import ctypes
import warnings

import numpy as np
import torch

from . import _ffi


def require_gpu() -> None:
  if not torch.cuda.is_available():
    raise RuntimeError(
        "mi355q needs an AMD GPU (torch.cuda.is_available() is False); the product"
        " path has no CPU fallback.")
  _ffi.lib()


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
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")  # non-writable buffer warning for mmap views
    t = torch.from_numpy(arr)
  if dtype is not None and t.dtype != dtype:
    t = t.to(dtype)
  return t.to(device(), non_blocking=True)


_NP_DTYPE: dict = {}     # torch dtype -> NumPy dtype


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
    # crosses process boundaries (gather of sharded results) as host data
    return (_host_carrier, (self.numpy(), None if self.packed is None else np.asarray(self.packed)))


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
