"""Device-level operators: one function per libmi355q entry point.

Inputs and outputs are torch tensors resident in HBM (torch is only the
allocator / stream owner). Every function enqueues on the current HIP stream and
returns without synchronizing. Shapes follow the [outer, channels, inner] view of
include/mi355q.h.
"""
from __future__ import annotations

import contextlib as _contextlib
import os
import threading

import numpy as np
import torch

from . import _ffi
from . import runtime as rt


def _f32(x: torch.Tensor) -> torch.Tensor:
  if x.dtype != torch.float32 or not x.is_cuda:
    raise TypeError(f"expected a float32 device tensor, got {x.dtype} on {x.device}")
  return x.contiguous()


def device_info() -> dict:
  import ctypes
  rt.require_gpu()
  cu, wave = ctypes.c_int32(0), ctypes.c_int32(0)
  name = ctypes.create_string_buffer(64)
  _ffi.check(_ffi.lib().mi355q_device_info(ctypes.byref(cu), ctypes.byref(wave), name, 64))
  return {"cu_count": cu.value, "wavefront": wave.value, "arch": name.value.decode()}


def minmax(x: torch.Tensor, outer: int, channels: int, inner: int):
  """K1. Returns (min, max) float32[channels]. ref: common_quantize.py:1311-1359."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != outer * channels * inner:
    raise ValueError("shape view does not match numel")
  if channels and (outer == 0 or inner == 0):
    raise ValueError("zero-size array to reduction operation minimum which has no identity")
  mn = rt.empty((channels,), torch.float32)
  mx = rt.empty((channels,), torch.float32)
  L = _ffi.lib()
  nbytes = L.mi355q_minmax_workspace_bytes(outer, channels, inner)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_minmax_f32(rt.ptr(x), outer, channels, inner, rt.ptr(mn), rt.ptr(mx),
                                 rt.ptr(ws), nbytes, rt.stream_ptr()))
  return mn, mx


def requant_sym(x: torch.Tensor, block: int, bits: int, clip: torch.Tensor | None = None,
                want_q: bool = True, want_packed: bool = False, want_scale_f16: bool = False):
  """Fused K1+K2+K3(+K4) for a [rows, cols] float32 weight buffer.

  Returns dict(q=int8[rows,cols]|None, packed=uint8[...]|None,
               scale=float32[rows] or [rows, cols/block], scale_f16=float16|None).
  ref: naive_min_max_quantize.py:34-110; transformation_utils.py:293-353.
  """
  rt.require_gpu()
  x = _f32(x)
  if x.dim() != 2:
    raise ValueError("requant_sym expects a 2-D [rows, cols] tensor")
  rows, cols = x.shape
  if block and cols % block != 0:
    raise ValueError(f"Quantized dimension {cols} in tensor shape {tuple(x.shape)} is not"
                     f" divisible by block size {block}.")
  nscale_shape = (rows, cols // block) if block else (rows,)
  scale = rt.empty(nscale_shape, torch.float32)
  q = rt.empty((rows, cols), torch.int8) if want_q else None
  packed = None
  if want_packed:
    per = 8 // bits
    if (rows * cols) % per:
      raise ValueError("packed output needs numel divisible by values-per-byte")
    packed = rt.empty((rows * cols // per,), torch.uint8)
  s16 = rt.empty(nscale_shape, torch.float16) if (want_scale_f16 and block) else None
  if clip is not None:
    clip = _f32(clip)
    if clip.numel() != scale.numel():
      raise ValueError("clip must have one entry per scale")
  _ffi.check(_ffi.lib().mi355q_requant_sym_f32(
      rt.ptr(x), rows, cols, block, bits, rt.ptr(clip), rt.ptr(q), rt.ptr(packed),
      rt.ptr(scale), rt.ptr(s16), rt.stream_ptr()))
  return {"q": q, "packed": packed, "scale": scale, "scale_f16": s16}


class RequantBatch:
  """Pre-staged pointer tables for mi355q_requant_sym_f32_batched.

  Holds `count` equally shaped weight buffers and their outputs in HBM so one
  launch requantizes all of them (used by bench.py and the model-level driver).
  """

  def __init__(self, xs, block: int, bits: int, want_q=True, want_packed=False,
               want_scale_f16=False):
    rt.require_gpu()
    self.xs = [_f32(x) for x in xs]
    self.rows, self.cols = self.xs[0].shape
    if any(x.shape != self.xs[0].shape for x in self.xs):
      raise ValueError("all tensors of a batch must share one shape")
    self.block, self.bits = block, bits
    n = len(self.xs)
    sshape = (self.rows, self.cols // block) if block else (self.rows,)
    per = 8 // bits
    self.q = [rt.empty((self.rows, self.cols), torch.int8) for _ in range(n)] if want_q else None
    self.packed = ([rt.empty((self.rows * self.cols // per,), torch.uint8) for _ in range(n)]
                   if want_packed else None)
    self.scale = [rt.empty(sshape, torch.float32) for _ in range(n)]
    self.scale_f16 = ([rt.empty(sshape, torch.float16) for _ in range(n)]
                      if (want_scale_f16 and block) else None)
    self._tx = rt.ptr_table(self.xs)
    self._tq = rt.ptr_table(self.q) if self.q else None
    self._tp = rt.ptr_table(self.packed) if self.packed else None
    self._ts = rt.ptr_table(self.scale)
    self._t16 = rt.ptr_table(self.scale_f16) if self.scale_f16 else None

  def run(self) -> None:
    _ffi.check(_ffi.lib().mi355q_requant_sym_f32_batched(
        rt.ptr(self._tx), len(self.xs), self.rows, self.cols, self.block, self.bits,
        rt.ptr(self._tq), rt.ptr(self._tp), rt.ptr(self._ts), rt.ptr(self._t16),
        rt.stream_ptr()))


_OUT_DTYPE = {8: torch.int8, 16: torch.int16, 32: torch.int32}


def quantize(x: torch.Tensor, outer: int, channels: int, inner: int, scale: torch.Tensor,
             zero_point: torch.Tensor | None, bits: int, narrow: bool,
             zp_via_f64: bool = False) -> torch.Tensor:
  """K3 with given parameters. scale is float32 or float64 [channels]; zero_point
  int32 [channels] or None. ref: uniform_quantize_tensor.py:273-362."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != outer * channels * inner:
    raise ValueError("shape view does not match numel")
  if scale.numel() != channels:
    raise ValueError("scale must have one entry per channel")
  if scale.dtype not in (torch.float32, torch.float64):
    raise TypeError("scale must be float32 or float64")
  out_bits = 8 if bits <= 8 else 16 if bits <= 16 else 32
  q = rt.empty(tuple(x.shape), _OUT_DTYPE[out_bits])
  if zero_point is not None:
    zero_point = zero_point.to(torch.int32).contiguous()
  _ffi.check(_ffi.lib().mi355q_quantize_f32(
      rt.ptr(x), outer, channels, inner, rt.ptr(scale.contiguous()),
      1 if scale.dtype == torch.float64 else 0, rt.ptr(zero_point), 1 if zp_via_f64 else 0,
      bits, 1 if narrow else 0, out_bits, rt.ptr(q), rt.stream_ptr()))
  return q


def dequantize(q: torch.Tensor, outer: int, channels: int, inner: int, scale: torch.Tensor,
               zero_point: torch.Tensor | None, diff_bits: int) -> torch.Tensor:
  """(q - zp) * scale. ref: uniform_quantize_tensor.py:365-409."""
  rt.require_gpu()
  in_bits = {torch.int8: 8, torch.int16: 16, torch.int32: 32}[q.dtype]
  out_f64 = in_bits == 32 or diff_bits == 32
  out = rt.empty(tuple(q.shape), torch.float64 if out_f64 else torch.float32)
  if zero_point is not None:
    zero_point = zero_point.to(torch.int32).contiguous()
  _ffi.check(_ffi.lib().mi355q_dequantize_f32(
      rt.ptr(q.contiguous()), in_bits, outer, channels, inner, rt.ptr(_f32(scale)),
      rt.ptr(zero_point), diff_bits, 1 if out_f64 else 0, rt.ptr(out), rt.stream_ptr()))
  return out


def pack_bits(q: torch.Tensor, bits: int) -> torch.Tensor:
  """K4. int8 values -> packed bytes. ref: transformation_utils.py:293-353."""
  rt.require_gpu()
  if q.dtype not in (torch.int8, torch.uint8) or not q.is_cuda:
    raise TypeError("pack_bits expects an int8/uint8 device tensor")
  q = q.contiguous().view(-1)
  n = q.numel()
  n_out = n if bits not in (2, 4) else -(-n * bits // 8)
  out = rt.empty((n_out,), torch.uint8)
  if bits not in (2, 4):
    out.copy_(q.view(torch.uint8))
    return out
  _ffi.check(_ffi.lib().mi355q_pack_bits(rt.ptr(q), n, bits, rt.ptr(out), rt.stream_ptr()))
  return out


def unpack_bits(packed: torch.Tensor, n: int, bits: int) -> torch.Tensor:
  """Inverse of K4: n sign-extended int8 values from the packed bytes of one fused launch."""
  rt.require_gpu()
  if packed.dtype != torch.uint8 or not packed.is_cuda:
    raise TypeError("unpack_bits expects a uint8 device tensor")
  packed = packed.contiguous().view(-1)
  if packed.numel() != -(-n * bits // 8):
    raise ValueError("packed size does not match the element count")
  out = rt.empty((n,), torch.int8)
  _ffi.check(_ffi.lib().mi355q_unpack_bits(rt.ptr(packed), n, bits, rt.ptr(out), rt.stream_ptr()))
  return out


def cast_f16(x: torch.Tensor) -> torch.Tensor:
  """float32 -> float16 (RNE), same shape. ref: nonlinear_quantize/float_casting.py:157-160."""
  rt.require_gpu()
  x = _f32(x)
  out = rt.empty(tuple(x.shape), torch.float16)
  _ffi.check(_ffi.lib().mi355q_cast_f32_to_f16(rt.ptr(x), x.numel(), rt.ptr(out), rt.stream_ptr()))
  return out


# How far the host may run ahead of the GPU in tables (one per calibration sample): with 8 the op walk of an 18-layer GPTQ
# calibration stood still during every burst of Hessian products (one per 32 samples, 0.4 s each) and the GPU then waited for the
# walk of the next 24 samples (4 ms each); 64 x 128 KB of pinned memory let the walk finish while the products run.
_TABLE_SLOTS = int(os.environ.get("MI355Q_TABLE_SLOTS", 64))
_TABLE_CAPACITY = 8192          # entries per row of a slot (2 rows of int64: 128 KB of pinned memory)
_TABLE_RING: dict = {}          # device index -> {"slots": [(pinned int64 [2, capacity], event)], "next": 0}
_TABLE_LOCK = threading.Lock()


def _table_to_device(pointers, lengths) -> torch.Tensor:
  """int64 [2, n] on the device, copied asynchronously out of a ring of pinned slots."""
  n = len(pointers)
  dev = rt.device()
  if n > _TABLE_CAPACITY:
    return torch.tensor([pointers, lengths], dtype=torch.int64).to(dev)
  with _TABLE_LOCK:                         # (slot choice, fill and enqueue as one step: threads share the ring)
    ring = _TABLE_RING.setdefault(dev.index, {"slots": [], "next": 0})
    k = ring["next"] % _TABLE_SLOTS
    ring["next"] += 1
    if len(ring["slots"]) <= k:
      ring["slots"].append((torch.empty((2, _TABLE_CAPACITY), dtype=torch.int64, pin_memory=True), torch.cuda.Event()))
    pinned, event = ring["slots"][k]
    event.synchronize()                     # the copy that last read this slot (eight tables ago)
    pinned[0, :n] = torch.tensor(pointers, dtype=torch.int64)
    pinned[1, :n] = torch.tensor(lengths, dtype=torch.int64)
    out = torch.empty((2, n), dtype=torch.int64, device=dev)
    out.copy_(pinned[:, :n], non_blocking=True)
    event.record()
  return out


class ActMinMaxBatch:
  """Pre-staged K7 launch over a fixed list of float32 device tensors (<= 65535).

  Pointer / length tables, workspace and the [count, 2] output live in HBM, so
  `run()` is just the two kernel launches (used per calibration sample).
  """

  def __init__(self, tensors, lo: float | None = -3e38, hi: float | None = 3e38):
    rt.require_gpu()
    self.tensors = [_f32(t) for t in tensors]
    if len(self.tensors) > 65535:
      raise ValueError("at most 65535 tensors per batch")
    if any(t.numel() == 0 for t in self.tensors):
      raise ValueError("zero-size array to reduction operation minimum which has no identity")
    if (lo is None) != (hi is None):
      raise ValueError("lo and hi must both be given or both be None")
    self.use = lo is not None
    self.lo = np.float32(lo if self.use else 0.0)
    self.hi = np.float32(hi if self.use else 0.0)
    n = len(self.tensors)
    self.out = rt.empty((n, 2), torch.float32)
    if n:
      # pointer and length tables travel in one copy -- from a pinned slot, so that the copy is queued
      # like a kernel (a pageable copy holds the calling thread until everything queued before it on
      # the stream has run: one such wait per calibration sample was 0.5 s of an 18-layer GPTQ
      # calibration, the op walk and the Hessian products taking turns instead of overlapping)
      both = _table_to_device([t.data_ptr() for t in self.tensors], [t.numel() for t in self.tensors])
      self._tab, self._numel = both[0], both[1]
      self._nbytes = _ffi.lib().mi355q_act_minmax_workspace_bytes(n)
      self._ws = rt.empty((self._nbytes,), torch.uint8)

  def run(self) -> torch.Tensor:
    n = len(self.tensors)
    if n:
      _ffi.check(_ffi.lib().mi355q_act_minmax_f32(
          rt.ptr(self._tab), rt.ptr(self._numel), n, self.lo, self.hi, 1 if self.use else 0,
          rt.ptr(self.out), rt.ptr(self._ws), self._nbytes, rt.stream_ptr()))
    return self.out


def act_minmax(tensors, lo: float | None = -3e38, hi: float | None = 3e38) -> torch.Tensor:
  """K7 over a list of float32 device tensors -> float32[count, 2] (min, max).

  ref: common_quantize.py:1362-1413. lo/hi None disables the range masks.
  """
  ts = list(tensors)
  if len(ts) <= 65535:
    return ActMinMaxBatch(ts, lo, hi).run()
  return torch.cat([ActMinMaxBatch(ts[i:i + 65535], lo, hi).run() for i in range(0, len(ts), 65535)])


def act_minmax_entries(pointers, lengths, lo: float = -3e38, hi: float = 3e38) -> torch.Tensor:
  """K7 over (device pointer, float32 element count) pairs whose memory the caller keeps alive until the launch has
  run -> float32 [count, 2]. The calibrator's K-samples-per-launch path: K x T activations, one table, one launch."""
  rt.require_gpu()
  n = len(pointers)
  out = rt.empty((n, 2), torch.float32)
  L = _ffi.lib()
  for i in range(0, n, _TABLE_CAPACITY):
    m = min(_TABLE_CAPACITY, n - i)
    both = _table_to_device(pointers[i:i + m], lengths[i:i + m])
    nbytes = L.mi355q_act_minmax_workspace_bytes(m)
    ws = rt.empty((nbytes,), torch.uint8)
    _ffi.check(L.mi355q_act_minmax_f32(rt.ptr(both[0]), rt.ptr(both[1]), m, np.float32(lo), np.float32(hi), 1,
                                       rt.ptr(out[i:i + m]), rt.ptr(ws), nbytes, rt.stream_ptr()))
  return out


_OCTAV_MODE = [None]          # None: MI355Q_OCTAV_FAST decides; "exact" / "fast" inside octav_mode()


def _octav_fast() -> bool:
  if _OCTAV_MODE[0] is not None:
    return _OCTAV_MODE[0] == "fast"
  return os.environ.get("MI355Q_OCTAV_FAST", "") not in ("", "0")


@_contextlib.contextmanager
def octav_mode(mode: str = "exact"):
  """The kernel OCTAV's clip search (ref octav.py:30-112) runs on inside this block.

  "exact" (the default everywhere): the masked float32 sums in NumPy 2's summation order -- the clipping constants, and
  with them scales and integers, are the reference's bit for bit; 0.045 of one read of the tensor.
  "fast" (also MI355Q_OCTAV_FAST=1): one pass over HBM, a row / block stays in registers for all ten iterations, the
  sums as float32 partials combined by a tree -- tolerance class T2 (scales within 1e-6 relative, integers
  +-1 on <= 1e-5 of the elements; the reference pins neither NumPy nor its summation order). Units of 4 .. 65536
  elements (a multiple of 4) with an axis given; anything else runs on the exact kernels in either mode."""
  if mode not in ("exact", "fast"):
    raise ValueError("octav_mode must be 'exact' or 'fast'")
  before, _OCTAV_MODE[0] = _OCTAV_MODE[0], mode
  try:
    yield
  finally:
    _OCTAV_MODE[0] = before


def octav_clip(x: torch.Tensor, units: int, unit_len: int, bits: int, max_iter: int = 10,
               exponent_divisor: float = 3.0, early_stop: bool = True, axis_given: bool = True):
  """K5. Returns (clip float32[units], iterations int). ref: octav.py:30-112.

  `axis_given=False` is the TENSORWISE form (axis=None in the reference).
  """
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != units * unit_len:
    raise ValueError("shape view does not match numel")
  clip = rt.empty((units,), torch.float32)
  iters = rt.empty((1,), torch.int32)
  L = _ffi.lib()
  if axis_given and _octav_fast() and unit_len % 4 == 0 and 4 <= unit_len <= 65536 and x.data_ptr() % 16 == 0:
    # the opt-in one-read kernel (octav_mode("fast")): tolerance class T2 instead of NumPy's summation order
    nbytes = L.mi355q_octav_workspace_bytes(units, max_iter)
    ws = rt.empty((max(nbytes, 1),), torch.uint8)
    _ffi.check(L.mi355q_octav_clip_fast_f32(
        rt.ptr(x), units, unit_len, bits, max_iter, np.float32(exponent_divisor), 1 if early_stop else 0,
        rt.ptr(clip), rt.ptr(iters), rt.ptr(ws), nbytes, rt.stream_ptr()))
    return clip, iters
  nbytes = L.mi355q_octav_rows_workspace_bytes(units, unit_len, max_iter)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_octav_clip_f32(
      rt.ptr(x), units, unit_len, bits, max_iter, np.float32(exponent_divisor),
      1 if early_stop else 0, 1 if axis_given else 0, rt.ptr(clip), rt.ptr(iters), rt.ptr(ws),
      nbytes, rt.stream_ptr()))
  return clip, iters


def octav_clip_nd(x: torch.Tensor, outer: int, channels: int, inner: int, bits: int,
                  max_iter: int = 10, exponent_divisor: float = 3.0, early_stop: bool = True):
  """K5 over the [outer, channels, inner] view: one clipping constant per channel."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != outer * channels * inner:
    raise ValueError("shape view does not match numel")
  clip = rt.empty((channels,), torch.float32)
  iters = rt.empty((1,), torch.int32)
  L = _ffi.lib()
  nbytes = L.mi355q_octav_workspace_bytes(channels, max_iter)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_octav_clip_nd_f32(
      rt.ptr(x), outer, channels, inner, bits, max_iter, np.float32(exponent_divisor),
      1 if early_stop else 0, rt.ptr(clip), rt.ptr(iters), rt.ptr(ws), nbytes, rt.stream_ptr()))
  return clip, iters


def mse_scale_nd(x: torch.Tensor, outer: int, channels: int, inner: int,
                 multiplier: float) -> torch.Tensor:
  """a14 over the [outer, channels, inner] view: scale[c] = multiplier * sqrt(mean(x[:, c, :]**2))."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != outer * channels * inner:
    raise ValueError("shape view does not match numel")
  scale = rt.empty((channels,), torch.float32)
  _ffi.check(_ffi.lib().mi355q_mse_scale_nd_f32(rt.ptr(x), outer, channels, inner,
                                                np.float32(multiplier), rt.ptr(scale), rt.stream_ptr()))
  return scale


def mse_requant(x: torch.Tensor, units: int, unit_len: int, multiplier: float, bits: int,
                narrow: bool) -> tuple[torch.Tensor, torch.Tensor]:
  """a14 whole, contiguous units: (scale float32 [units], q int8 [units * unit_len]) -- the MSE scale in NumPy's order
  and clip(rint(x / scale)) with a zero zero point, one kernel where the unit length allows. ref: mse.py:100-128."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != units * unit_len:
    raise ValueError("shape view does not match numel")
  scale = rt.empty((units,), torch.float32)
  q = rt.empty(tuple(x.shape), torch.int8)
  _ffi.check(_ffi.lib().mi355q_mse_requant_f32(rt.ptr(x), units, unit_len, np.float32(multiplier), bits,
                                               1 if narrow else 0, rt.ptr(scale), rt.ptr(q), rt.stream_ptr()))
  return scale, q


def mse_scale(x: torch.Tensor, units: int, unit_len: int, multiplier: float) -> torch.Tensor:
  """a14. scale[u] = multiplier * sqrt(mean(x_u**2)). ref: mse.py:100-109."""
  rt.require_gpu()
  x = _f32(x)
  if x.numel() != units * unit_len:
    raise ValueError("shape view does not match numel")
  scale = rt.empty((units,), torch.float32)
  _ffi.check(_ffi.lib().mi355q_mse_scale_f32(rt.ptr(x), units, unit_len, np.float32(multiplier),
                                             rt.ptr(scale), rt.stream_ptr()))
  return scale


def hadamard_rotate(x: torch.Tensor, h: int) -> torch.Tensor:
  """K6. reshape(x, (-1, h)) @ (H_h / sqrt(h)), same shape as x. ref: hadamard_rotation.py:93-134."""
  rt.require_gpu()
  x = _f32(x)
  if h <= 0 or h & (h - 1):
    raise ValueError("Hadamard matrix size must be a power of 2. ")
  if x.numel() % h:
    raise ValueError("tensor size is not a multiple of the Hadamard size")
  out = torch.empty_like(x)
  _ffi.check(_ffi.lib().mi355q_hadamard_rotate_f32(rt.ptr(x), x.numel() // h, h, rt.ptr(out),
                                                   rt.stream_ptr()))
  return out


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
         lower_only: bool = False) -> torch.Tensor:
  """MFMA GEMM on 2-D float32 / float64 device tensors: op(a) @ op(b)."""
  rt.require_gpu()
  if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.float64):
    raise TypeError("gemm expects matching float32 or float64 tensors")
  a, b = a.contiguous(), b.contiguous()
  m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
  k2, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
  if k != k2:
    raise ValueError("inner dimensions do not match")
  c = torch.zeros((m, n), dtype=a.dtype, device=a.device)
  a_i, a_k = (1, a.shape[1]) if trans_a else (a.shape[1], 1)
  b_k, b_j = (1, b.shape[1]) if trans_b else (b.shape[1], 1)
  fn = _ffi.lib().mi355q_gemm_f32 if a.dtype == torch.float32 else _ffi.lib().mi355q_gemm_f64
  _ffi.check(fn(rt.ptr(a), a_i, a_k, rt.ptr(b), b_k, b_j, rt.ptr(c), n, 1, m, n, k, 1.0, 0.0,
                1 if lower_only else 0, rt.stream_ptr()))
  return c


def gptq_xtx(x: torch.Tensor, alpha: float) -> torch.Tensor:
  """K8. float64 [d, d] = alpha * (x^T x) for float32 x [n, d]. ref: gptq.py:100-107."""
  rt.require_gpu()
  x = _f32(x)
  n, d = x.shape
  h = rt.empty((d, d), torch.float64)
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_xtx_workspace_bytes(n, d)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_gptq_xtx_f32(rt.ptr(x), n, d, float(alpha), rt.ptr(h), rt.ptr(ws), nbytes,
                                   rt.stream_ptr()))
  return h


_XTX_SCRATCH: dict = {}      # (device index, d) -> uint8 scratch for products of <= 16384-token slabs


def gptq_xtx_accum(x: torch.Tensor, product: torch.Tensor | None) -> torch.Tensor:
  """product (float32 [d, d], lower-triangular part valid) (+)= x^T x for float32 x [n, d];
  None starts a new product. The scratch (bfloat16 planes of a slab / split-K partials, up to
  1.5 GiB at d = 16384) is kept per device and width: calibration calls this once per slab and
  Hessian, back to back on one stream."""
  rt.require_gpu()
  x = _f32(x)
  n, d = x.shape
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_xtx_accum_workspace_bytes(n, d)
  key = (x.device.index, d)
  ws = _XTX_SCRATCH.get(key)
  if ws is None or ws.numel() < nbytes:
    ws = _XTX_SCRATCH[key] = rt.empty((max(nbytes, L.mi355q_gptq_xtx_accum_workspace_bytes(16384, d), 1),), torch.uint8)
  fresh = product is None
  if fresh:
    product = rt.empty((d, d), torch.float32)
  _ffi.check(L.mi355q_gptq_xtx_accum_f32(rt.ptr(x), n, d, rt.ptr(product), 0 if fresh else 1, rt.ptr(ws),
                                         ws.numel(), rt.stream_ptr()))
  return product


def gptq_xtx_finish(product: torch.Tensor, alpha: float) -> torch.Tensor:
  """float64 [d, d] = alpha * product, both triangles (ref gptq.py:100-107's scaling)."""
  rt.require_gpu()
  d = product.shape[0]
  h = rt.empty((d, d), torch.float64)
  _ffi.check(_ffi.lib().mi355q_gptq_xtx_finish_f64(rt.ptr(product), d, float(alpha), rt.ptr(h), rt.stream_ptr()))
  return h


def release_scratch() -> None:
  """Gives the cached product scratch back (ParamsGenerator / Calibrator call this when done)."""
  _XTX_SCRATCH.clear()


def gptq_hessian_merge(h_cur: torch.Tensor, n_cur: float, h_new: torch.Tensor, n_new: float):
  """(h_cur*n_cur + h_new*n_new)/(n_cur+n_new), float64. ref: qsv_utils.py:71-88."""
  rt.require_gpu()
  out = torch.empty_like(h_cur)
  _ffi.check(_ffi.lib().mi355q_gptq_hessian_merge_f64(
      rt.ptr(h_cur.contiguous()), float(n_cur), rt.ptr(h_new.contiguous()), float(n_new),
      h_cur.shape[0], rt.ptr(out), rt.stream_ptr()))
  return out


def gptq_hinv(hessian: torch.Tensor, damp_factor: float = 0.01):
  """K9. Returns (hinv float32 [d,d], info int32[1]). ref: gptq.py:111-128."""
  rt.require_gpu()
  if hessian.dtype != torch.float64:
    hessian = hessian.to(torch.float64)
  hessian = hessian.contiguous()
  d = hessian.shape[0]
  hinv = rt.empty((d, d), torch.float32)
  info = rt.empty((1,), torch.int32)
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_hinv_workspace_bytes(d)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_gptq_hinv_f64(rt.ptr(hessian), d, float(damp_factor), rt.ptr(hinv),
                                    rt.ptr(info), rt.ptr(ws), nbytes, rt.stream_ptr()))
  return hinv, info


def gptq_hinv_from_product(product: torch.Tensor, alpha: float, damp_factor: float = 0.01):
  """K9 on hessian = alpha * product (float32 X^T X, lower triangle valid) without materializing the
  float64 Hessian; same (hinv, info) as gptq_hinv(gptq_xtx_finish(product, alpha))."""
  rt.require_gpu()
  d = product.shape[0]
  hinv = rt.empty((d, d), torch.float32)
  info = rt.empty((1,), torch.int32)
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_hinv_workspace_bytes(d)
  held = HinvWorkspace.current.pointer(nbytes) if HinvWorkspace.current is not None else None
  ws = None if held is not None else rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check(L.mi355q_gptq_hinv_from_product_f32(rt.ptr(_f32(product)), d, float(alpha), float(damp_factor), rt.ptr(hinv),
                                                 rt.ptr(info), held if held is not None else rt.ptr(ws), nbytes, rt.stream_ptr()))
  return hinv, info


class HinvWorkspace:
  """The workspace of the d >= 4096 inverses (5 GiB at d = 16384), allocated ahead of their first call on a helper
  thread: a fresh GiB of HBM costs its caller ~30 ms of hipMalloc, and whoever starts the first large inverse does so
  on a GPU that has just run dry (the end of calibration). mi355q_device_alloc runs without the interpreter lock and
  outside the framework's allocator (whose lock would stall every allocation of the feeding thread meanwhile).
  gptq_hinv_from_product() uses it while one is installed (`with`, or install() / release()); calls on one stream
  follow each other, so one workspace serves them all."""
  current = None

  def __init__(self, d: int, lanes: int = 1):
    """lanes = 2: room for the pair of large matrices gptq_hinv_from_product_batched keeps in flight."""
    import ctypes
    import threading
    rt.require_gpu()
    self.nbytes = int(_ffi.lib().mi355q_gptq_hinv_from_product_batched_workspace_bytes(lanes, d)) if lanes > 1 \
        else int(_ffi.lib().mi355q_gptq_hinv_workspace_bytes(d))
    self.d = d
    self._ptr = ctypes.c_void_p()
    self._status = 0
    self._device = torch.cuda.current_device()

    def work():
      torch.cuda.set_device(self._device)
      self._status = _ffi.lib().mi355q_device_alloc(self.nbytes, ctypes.byref(self._ptr))
    self._thread = threading.Thread(target=work, name="mi355q-hinv-workspace", daemon=True)
    self._thread.start()

  def pointer(self, nbytes: int):
    """Device pointer when the workspace is there and large enough, else None (the caller allocates its own)."""
    self._thread.join()
    return self._ptr if self._status == 0 and self._ptr.value and nbytes <= self.nbytes else None

  def install(self):
    HinvWorkspace.current = self
    return self

  def release(self) -> None:
    """Call with the stream that used it drained (hipFree waits for the device anyway)."""
    if HinvWorkspace.current is self:
      HinvWorkspace.current = None
    self._thread.join()
    if self._ptr.value:
      _ffi.lib().mi355q_device_free(self._ptr)
      self._ptr.value = None

  def __enter__(self):
    return self.install()

  def __exit__(self, *exc):
    self.release()


def gptq_hinv_batched(hessians, damp_factor: float = 0.01):
  """K9 for several Hessians of one order: [(hinv float32 [d, d], info int32[1]), ...], the same
  bits as gptq_hinv on each (for d < 4096 the independent chains interleave on a stream pool)."""
  import ctypes
  rt.require_gpu()
  hs = [h.contiguous() if h.dtype == torch.float64 else h.to(torch.float64).contiguous() for h in hessians]
  if not hs:
    return []
  d = hs[0].shape[0]
  if any(tuple(h.shape) != (d, d) for h in hs):
    raise ValueError("all Hessians of a batch must have one order")
  n = len(hs)
  hinv = rt.empty((n, d, d), torch.float32)
  info = rt.empty((n,), torch.int32)
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_hinv_batched_workspace_bytes(n, d)
  held = HinvWorkspace.current.pointer(nbytes) if HinvWorkspace.current is not None else None   # (calls on one stream follow each other)
  ws = None if held is not None else rt.empty((max(nbytes, 1),), torch.uint8)
  src = (ctypes.c_void_p * n)(*[h.data_ptr() for h in hs])
  dst = (ctypes.c_void_p * n)(*[hinv[i].data_ptr() for i in range(n)])
  _ffi.check(L.mi355q_gptq_hinv_f64_batched(src, n, d, float(damp_factor), dst, rt.ptr(info),
                                            held if held is not None else rt.ptr(ws), nbytes, rt.stream_ptr()))
  return [(hinv[i], info[i:i + 1]) for i in range(n)]




def gptq_hinv_from_product_batched(forms, damp_factor: float = 0.01):
  """K9 for several Hessians of one order handed over as (product float32 [d, d], alpha) pairs (what calibration leaves
  behind: gptq.HessianAccumulator.product_form): [(hinv float32 [d, d], info int32[1]), ...], the same bits as
  gptq_hinv_from_product on each (mi355q_gptq_hinv_from_product_f32_batched; MI355Q_HINV_PAIRS=1: two d >= 4096 matrices in flight)."""
  import ctypes
  rt.require_gpu()
  forms = [(_f32(p), float(a)) for p, a in forms]
  if not forms:
    return []
  d = forms[0][0].shape[0]
  if any(tuple(p.shape) != (d, d) for p, _ in forms):
    raise ValueError("all Hessians of a batch must have one order")
  n = len(forms)
  hinvs = [rt.empty((d, d), torch.float32) for _ in range(n)]      # (separate allocations: each is given back with its last reader)
  info = rt.empty((n,), torch.int32)
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_hinv_from_product_batched_workspace_bytes(n, d)
  held = HinvWorkspace.current.pointer(nbytes) if HinvWorkspace.current is not None else None
  ws = None if held is not None else rt.empty((max(nbytes, 1),), torch.uint8)
  src = (ctypes.c_void_p * n)(*[p.data_ptr() for p, _ in forms])
  alphas = (ctypes.c_double * n)(*[a for _, a in forms])
  dst = (ctypes.c_void_p * n)(*[h.data_ptr() for h in hinvs])
  _ffi.check(L.mi355q_gptq_hinv_from_product_f32_batched(src, alphas, n, d, float(damp_factor), dst, rt.ptr(info),
                                                         held if held is not None else rt.ptr(ws), nbytes, rt.stream_ptr()))
  return [(hinvs[i], info[i:i + 1]) for i in range(n)]


@_contextlib.contextmanager
def hessian_product(mode: str = "exact"):
  """The kernel GPTQ's Hessian products (mi355q_gptq_xtx_f32 / _accum_f32) run on inside this block.

  "exact" (the default everywhere): every float32 product from the exact three-way bfloat16 split, six
  bf16 MFMA products (xtx_bf16x3.hip) -- float32-sgemm-class like the reference's x.T.dot(x), and what the
  recorded parity rates are taken with (the d = 16384 chain reproduces the oracle's integers).
  "fast": the two-way float16 split, three f16 MFMA products, 22-23 of the 24 mantissa bits, 1.8 x faster;
  1.2e-3 of the d = 16384 integers then differ from the oracle's (profiles/r04_parity_rates.txt). The
  library reads MI355Q_XTX_F16X2 per call, which is what this sets."""
  import os
  if mode not in ("exact", "fast"):
    raise ValueError("hessian_product mode must be 'exact' or 'fast'")
  before = os.environ.get("MI355Q_XTX_F16X2")
  if mode == "fast":
    os.environ["MI355Q_XTX_F16X2"] = "1"
  else:
    os.environ.pop("MI355Q_XTX_F16X2", None)
  try:
    yield
  finally:
    if before is None:
      os.environ.pop("MI355Q_XTX_F16X2", None)
    else:
      os.environ["MI355Q_XTX_F16X2"] = before


def gptq_apply(w: torch.Tensor, hinv: torch.Tensor, scale: torch.Tensor,
               zero_point: torch.Tensor | None, scale_mode: int, block_size: int, bits: int,
               narrow: bool, zp_via_f64: bool, diff_bits: int) -> torch.Tensor:
  """K10. int8 [rows, d] (int32 for targets of 9..32 bits). ref: gptq.py:131-216."""
  rt.require_gpu()
  w = _f32(w)
  rows, d = w.shape
  wide = bits > 8         # int32 [rows, d] from mi355q_gptq_apply_wide_f32 (9..16 bits: the caller narrows the container; 17..32: int32 is the container)
  q = rt.empty((rows, d), torch.int32 if wide else torch.int8)
  if zero_point is not None:
    zero_point = zero_point.to(torch.int32).contiguous()
  L = _ffi.lib()
  nbytes = L.mi355q_gptq_apply_workspace_bytes(rows, d)
  ws = rt.empty((max(nbytes, 1),), torch.uint8)
  _ffi.check((L.mi355q_gptq_apply_wide_f32 if wide else L.mi355q_gptq_apply_f32)(
      rt.ptr(w), rows, d, rt.ptr(_f32(hinv)), rt.ptr(scale.contiguous()),
      1 if scale.dtype == torch.float64 else 0, rt.ptr(zero_point), scale_mode, block_size, bits,
      1 if narrow else 0, 1 if zp_via_f64 else 0, diff_bits, rt.ptr(q), rt.ptr(ws), nbytes,
      rt.stream_ptr()))
  return q


# ------------------------------------------------------------------------ OSCAR (f4) ---
def _f64_dev(a) -> torch.Tensor:
  return rt.to_device(np.ascontiguousarray(a, dtype=np.float64))


def oscar_col_sumsq(x: torch.Tensor, mean: bool) -> torch.Tensor:
  """x float32 [rows, d] -> float64 [d]: sum_r x[r, j]^2, rows in order (/ rows when mean)."""
  rt.require_gpu()
  x = _f32(x)
  rows, d = x.shape
  out = rt.empty((d,), torch.float64)
  _ffi.check(_ffi.lib().mi355q_oscar_col_sumsq_f32(rt.ptr(x), rows, d, int(mean), rt.ptr(out),
                                                   rt.stream_ptr()))
  return out


def oscar_group_terms(w: torch.Tensor, s: torch.Tensor, g: int):
  """(sums float64 [d/g], winner int32 [d/g, n], wsq float64 [d/g, n]) for scales s float64 [d]."""
  rt.require_gpu()
  w = _f32(w)
  n, d = w.shape
  groups = d // g
  top2 = rt.empty((groups * n,), torch.float64)
  winner = rt.empty((groups, n), torch.int32)
  wsq = rt.empty((groups, n), torch.float64)
  sums = rt.empty((groups,), torch.float64)
  _ffi.check(_ffi.lib().mi355q_oscar_group_terms_f32(
      rt.ptr(w), rt.ptr(s), n, d, g, rt.ptr(top2), rt.ptr(winner), rt.ptr(wsq), rt.ptr(sums),
      rt.stream_ptr()))
  return sums, winner, wsq


def oscar_winner_energy(winner: torch.Tensor, wsq: torch.Tensor, d: int, g: int) -> torch.Tensor:
  rt.require_gpu()
  n = winner.shape[1]
  eff = rt.empty((d,), torch.float64)
  _ffi.check(_ffi.lib().mi355q_oscar_winner_energy_f64(rt.ptr(winner), rt.ptr(wsq), n, d, g,
                                                       rt.ptr(eff), rt.stream_ptr()))
  return eff


def oscar_clip_bounds(w: torch.Tensor, s: torch.Tensor, masses: torch.Tensor, g: int,
                      u: torch.Tensor, noise: torch.Tensor, qmax: int, blockwise_scale: bool = False,
                      want_bounds: bool = True, want_scale: bool = False, want_rows_left: bool = False):
  """Optimal clip bound (float64) of every g-element segment of the flattened weight and / or
  the symmetric scale derived from it (float64 values; float16-representable when blockwise).
  `want_rows_left`: also the per-segment flags of the rows the prefix kernel handed to the full sort + scan
  (uint8 [n * d / g]; None when the call took the full route for every segment) -- for the tests and the bench."""
  import ctypes
  import os
  rt.require_gpu()
  w = _f32(w)
  n, d = w.shape
  need = ctypes.c_size_t(0)
  _ffi.check(_ffi.lib().mi355q_oscar_clip_workspace_bytes(n, d, g, ctypes.byref(need)))
  ws = rt.empty((need.value,), torch.uint8)
  bounds = rt.empty((n * d // g,), torch.float64) if want_bounds else None
  scale = rt.empty((n * d // g,), torch.float64) if want_scale else None
  _ffi.check(_ffi.lib().mi355q_oscar_clip_bounds_f32(
      rt.ptr(w), rt.ptr(s), rt.ptr(masses), n, d, g, rt.ptr(u), rt.ptr(noise), int(qmax),
      int(blockwise_scale), rt.ptr(bounds), rt.ptr(scale), rt.ptr(ws), need.value, rt.stream_ptr()))
  if want_rows_left:
    took_prefix = (g == d and 384 <= g <= 16384 and qmax >= 7 and os.environ.get("MI355Q_OSCAR_PREFIX", "")[:1] != "0")
    slab = (n * d * 8 + 255) // 256 * 256
    return bounds, scale, (ws[4 * slab:4 * slab + n * d // g].clone() if took_prefix else None)
  return bounds, scale


def oscar_quantize(w: torch.Tensor, s: torch.Tensor, scale: torch.Tensor, g: int, qlo: int,
                   qhi: int) -> torch.Tensor:
  rt.require_gpu()
  w = _f32(w)
  n, d = w.shape
  out = rt.empty((n, d), torch.int8)
  _ffi.check(_ffi.lib().mi355q_oscar_quantize_f32(rt.ptr(w), rt.ptr(s), rt.ptr(scale), n, d, g, qlo,
                                                  qhi, rt.ptr(out), rt.stream_ptr()))
  return out


# ------------------------------------------------------- dequantized weight recovery ---
def dwr_scales(w: torch.Tensor, g: int, rounded: bool) -> torch.Tensor:
  """float64 [n*d/g]: smallest positive step of every g-element segment's sorted magnitudes."""
  import ctypes
  rt.require_gpu()
  w = _f32(w)
  n, d = w.shape
  need = ctypes.c_size_t(0)
  _ffi.check(_ffi.lib().mi355q_oscar_clip_workspace_bytes(n, d, g, ctypes.byref(need)))
  ws = rt.empty((need.value,), torch.uint8)
  out = rt.empty((n * d // g,), torch.float64)
  _ffi.check(_ffi.lib().mi355q_dwr_scales_f32(rt.ptr(w), n, d, g, int(rounded), rt.ptr(out), rt.ptr(ws),
                                              need.value, rt.stream_ptr()))
  return out


def dwr_max_error(w: torch.Tensor, q: torch.Tensor, scale: torch.Tensor, g: int) -> float:
  """max |q * scale - w| over the tensor (float64; NaN if any element is NaN)."""
  rt.require_gpu()
  w = _f32(w)
  out = rt.empty((1,), torch.float64)
  _ffi.check(_ffi.lib().mi355q_dwr_max_error_f32(rt.ptr(w), rt.ptr(q), rt.ptr(scale), w.numel(), g,
                                                 rt.ptr(out), rt.stream_ptr()))
  return float(out.item())
