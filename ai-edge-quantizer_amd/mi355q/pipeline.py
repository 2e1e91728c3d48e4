"""Streaming requantization of a model's weight buffers: host -> HBM -> host.

The reference's ParamsGenerator walks the ops one by one and quantizes each
weight where it lies in host memory (ref: params_generator.py:110-183). On an
MI355X the kernels need ~15 us per 64 MiB buffer while PCIe needs ~1.3 ms for the
same bytes, so a model-level driver is a copy pipeline around the kernels:

    host tensor --memcpy--> pinned slot --H2D (copy stream)--> HBM
        --mi355q_requant_sym_f32 (compute stream)--> HBM
        --D2H (copy-back stream)--> pinned slot --> NumPy result

with a ring of slots so that the three stages of consecutive tensors overlap
(HIP streams + events; no device synchronisation inside the loop). Results are the
same arrays `get_tensor_quant_params` returns (scale, int8 `quantized_data`) plus,
optionally, the packed bytes `quantize_tensor` would store.
"""
from __future__ import annotations

import dataclasses
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from . import _ffi
from . import runtime as rt


@dataclasses.dataclass
class RequantResult:
  scale: np.ndarray                 # float32, [rows] or [rows, cols/block]
  quantized_data: Optional[np.ndarray]   # int8 [rows, cols] (None when only packed bytes were asked for)
  packed: Optional[np.ndarray]      # uint8 bytes as stored in the flatbuffer
  scale_f16: Optional[np.ndarray]   # float16 blockwise scales (the `<name>_scales` tensor)


class _Slot:
  def __init__(self, max_elems: int, max_scales: int, want_q: bool, want_packed: bool, block: bool):
    pin = dict(pin_memory=True)
    self.h_in = torch.empty(max_elems, dtype=torch.float32, **pin)
    self.d_in = rt.empty((max_elems,), torch.float32)
    self.d_q = rt.empty((max_elems,), torch.int8) if want_q else None
    self.h_q = torch.empty(max_elems, dtype=torch.int8, **pin) if want_q else None
    self.d_p = rt.empty((max_elems,), torch.uint8) if want_packed else None
    self.h_p = torch.empty(max_elems, dtype=torch.uint8, **pin) if want_packed else None
    self.d_s = rt.empty((max_scales,), torch.float32)
    self.h_s = torch.empty(max_scales, dtype=torch.float32, **pin)
    self.d_s16 = rt.empty((max_scales,), torch.float16) if block else None
    self.h_s16 = torch.empty(max_scales, dtype=torch.float16, **pin) if block else None
    self.done = torch.cuda.Event()     # D2H of the previous occupant finished
    self.meta = None


def _threaded_copy(pool: ThreadPoolExecutor, dst: np.ndarray, src: np.ndarray, parts: int = 8) -> None:
  """memcpy into the pinned slot with several threads (NumPy releases the GIL)."""
  n = src.size
  if n < (1 << 20):
    np.copyto(dst[:n], src)
    return
  step = -(-n // parts)
  list(pool.map(lambda k: np.copyto(dst[k:min(n, k + step)], src[k:min(n, k + step)]),
                range(0, n, step)))


def requantize_weights(tensors: Iterable[np.ndarray], block: int, bits: int, want_q: bool = True,
                       want_packed: bool = False, slots: int = 3,
                       max_elems: Optional[int] = None) -> list[RequantResult]:
  """Symmetric min/max requantization (CHANNELWISE dim 0 when block == 0, else
  BLOCKWISE along the last dim) of every float32 tensor in `tensors`, pipelined.

  ref: naive_min_max_quantize.py:34-110 per tensor; transformation_utils.py:293-353 (packed).
  """
  rt.require_gpu()
  tensors = list(tensors)
  if not tensors:
    return []
  for t in tensors:
    if t.dtype != np.float32 or t.ndim < 2:
      raise TypeError("requantize_weights expects float32 tensors of rank >= 2")
    if block and t.shape[-1] % block:
      raise ValueError(f"Quantized dimension {t.shape[-1]} in tensor shape {t.shape} is not"
                       f" divisible by block size {block}.")
  cap = max_elems or max(t.size for t in tensors)
  cap_scales = max((t.size // block) if block else t.shape[0] for t in tensors)
  ring = [_Slot(cap, cap_scales, want_q, want_packed, bool(block)) for _ in range(slots)]
  s_in, s_run, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
  lib = _ffi.lib()
  results: list[Optional[RequantResult]] = [None] * len(tensors)
  per = 8 // bits
  pool = ThreadPoolExecutor(max_workers=8)

  def collect(slot: _Slot):
    if slot.meta is None:
      return
    idx, shape, n, ns = slot.meta
    slot.done.synchronize()
    sshape = (shape[0],) if not block else tuple(shape[:-1]) + (shape[-1] // block,)
    results[idx] = RequantResult(
        scale=slot.h_s[:ns].numpy().reshape(sshape).copy(),
        quantized_data=slot.h_q[:n].numpy().reshape(shape).copy() if want_q else None,
        packed=slot.h_p[: n // per].numpy().copy() if want_packed else None,
        scale_f16=slot.h_s16[:ns].numpy().reshape(sshape).copy() if block else None)
    slot.meta = None

  try:
    for i, t in enumerate(tensors):
      slot = ring[i % slots]
      collect(slot)  # the slot's previous tensor has left the GPU
      n = t.size
      rows = t.shape[0] if not block else n // t.shape[-1]
      cols = n // rows
      ns = n // block if block else rows
      if want_packed and n % per:
        raise ValueError("packed output needs numel divisible by values-per-byte")
      _threaded_copy(pool, slot.h_in.numpy(), np.ascontiguousarray(t).reshape(-1))
      with torch.cuda.stream(s_in):
        slot.d_in[:n].copy_(slot.h_in[:n], non_blocking=True)
        ev_in = torch.cuda.Event()
        ev_in.record(s_in)
      with torch.cuda.stream(s_run):
        s_run.wait_event(ev_in)
        _ffi.check(lib.mi355q_requant_sym_f32(
            rt.ptr(slot.d_in), rows, cols, block, bits, None, rt.ptr(slot.d_q), rt.ptr(slot.d_p),
            rt.ptr(slot.d_s), rt.ptr(slot.d_s16), rt.stream_ptr()))
        ev_run = torch.cuda.Event()
        ev_run.record(s_run)
      with torch.cuda.stream(s_out):
        s_out.wait_event(ev_run)
        slot.h_s[:ns].copy_(slot.d_s[:ns], non_blocking=True)
        if want_q:
          slot.h_q[:n].copy_(slot.d_q[:n], non_blocking=True)
        if want_packed:
          slot.h_p[: n // per].copy_(slot.d_p[: n // per], non_blocking=True)
        if block:
          slot.h_s16[:ns].copy_(slot.d_s16[:ns], non_blocking=True)
        slot.done.record(s_out)
      slot.meta = (i, t.shape, n, ns)
    for slot in ring:
      collect(slot)
  finally:
    pool.shutdown(wait=True)
  return results  # type: ignore[return-value]
