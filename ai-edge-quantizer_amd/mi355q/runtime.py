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


def empty(shape, dtype) -> torch.Tensor:
  return torch.empty(shape, dtype=dtype, device=device())


def to_numpy(t: torch.Tensor) -> np.ndarray:
  return t.cpu().numpy()


def ptr_table(tensors) -> torch.Tensor:
  """Device table of device pointers (int64) for the *_batched entry points."""
  host = torch.tensor([0 if t is None else t.data_ptr() for t in tensors], dtype=torch.int64)
  return host.to(device())
