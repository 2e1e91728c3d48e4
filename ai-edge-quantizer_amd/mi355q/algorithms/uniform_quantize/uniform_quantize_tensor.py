"""Tensor-level uniform quantization, GPU backed.

Mirror of the reference module of the same name
(ref: algorithms/uniform_quantize/uniform_quantize_tensor.py). Per-element work
(divide / round / clip / cast, dequantize) runs in libmi355q kernels; the
O(#scales) parameter math of `tensor_zp_scale_from_min_max` stays on the host in
NumPy exactly as the reference writes it (it is a handful of scalars for
activations; weights get it fused in-kernel, see naive_min_max_quantize).
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ...utils import tfl_flatbuffer_utils


@dataclasses.dataclass(frozen=True)
class IntType:
  num_bits: int
  signed: bool


def is_blockwise(granularity: qtyping.QuantGranularity) -> bool:
  return "BLOCKWISE" in str(granularity)


def get_quantized_range(qtype: IntType) -> tuple[float, float]:
  """ref :37-45."""
  if qtype.signed:
    return float(-(2 ** (qtype.num_bits - 1))), float(2 ** (qtype.num_bits - 1) - 1)
  return 0.0, float(2**qtype.num_bits - 1)


def extract_block_size_from_granularity(granularity: qtyping.QuantGranularity) -> int:
  """ref :48-61."""
  return {qtyping.QuantGranularity.BLOCKWISE_32: 32, qtyping.QuantGranularity.BLOCKWISE_64: 64,
          qtyping.QuantGranularity.BLOCKWISE_128: 128,
          qtyping.QuantGranularity.BLOCKWISE_256: 256}.get(granularity, 0)


def _get_numpy_dtype(qtype: IntType):
  for limit, s, u in ((8, np.int8, np.uint8), (16, np.int16, np.uint16), (32, np.int32, np.uint32)):
    if qtype.num_bits <= limit:
      return s if qtype.signed else u
  return np.int64 if qtype.signed else np.uint64


def assign_quantized_type(tensor: np.ndarray, qtype: IntType) -> np.ndarray:
  return np.asarray(tensor).astype(_get_numpy_dtype(qtype), copy=False)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
  """float32 -> bfloat16 (RNE) -> float32; stands in for `.astype(ml_dtypes.bfloat16)`."""
  x = np.ascontiguousarray(x, dtype=np.float32)
  bits = x.view(np.uint32)
  out = ((bits + np.uint32(0x7FFF) + ((bits >> np.uint32(16)) & np.uint32(1)))
         & np.uint32(0xFFFF0000)).view(np.float32)
  nan = np.isnan(x)
  if nan.any():
    out = np.where(nan, np.float32(np.nan), out)
  return out


def fix_quantization_params_rank(
    tensor_data: np.ndarray, quantization_params: qtyping.UniformQuantParams
) -> qtyping.UniformQuantParams:
  """Expand scale / zero_point to the tensor's rank (ref :112-161)."""
  scales, zps = quantization_params.scale, quantization_params.zero_point
  if tensor_data.ndim == scales.ndim:
    return quantization_params
  if tensor_data.ndim == 0:
    if scales.size != 1 or zps.size != 1:
      raise ValueError("Scale and zero_point must contain single element for scalar tensor."
                       f" Got scale: {scales}, zero_point: {zps}")
    scales, zps = np.array(scales.item()), np.array(zps.item())
  else:
    dims = [d for d in range(tensor_data.ndim) if d != quantization_params.quantized_dimension]
    scales, zps = np.expand_dims(scales, axis=dims), np.expand_dims(zps, axis=dims)
  return dataclasses.replace(quantization_params, scale=scales, zero_point=zps)


def _get_tensor_shape_for_blockwise(tensor_shape: Sequence[int], quantized_dim: int,
                                    block_size: int) -> list[int]:
  """ref :164-194."""
  shape = []
  for i, v in enumerate(tensor_shape):
    if i == quantized_dim:
      if v % block_size != 0:
        raise ValueError(f"Quantized dimension {v} in tensor shape {tensor_shape} is not"
                         f" divisible by block size {block_size}.")
      shape += [int(v / block_size), block_size]
    else:
      shape.append(v)
  return shape


def reshape_data_for_blockwise(tensor_data: np.ndarray, op_name: qtyping.TFLOperationName,
                               granularity: qtyping.QuantGranularity) -> tuple[np.ndarray, int]:
  """ref :197-219."""
  qdim = tfl_flatbuffer_utils.TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM[op_name]
  block = extract_block_size_from_granularity(granularity)
  return tensor_data.reshape(_get_tensor_shape_for_blockwise(tensor_data.shape, qdim, block)), qdim + 1


def _is_valid_quantization_params(tensor_data: np.ndarray,
                                  quantization_params: qtyping.UniformQuantParams) -> None:
  """ref :589-638."""
  s, z = quantization_params.scale, quantization_params.zero_point
  if s.shape != z.shape and z.size != 1:
    raise ValueError("scale and zero_point must have the same shape or zero_point must have"
                     f" only one element. Got {s.shape} and {z.shape}")
  if tensor_data.ndim != s.ndim or tensor_data.ndim != z.ndim:
    raise ValueError(f"Ranks of scales ({s.ndim}) and zps ({z.ndim}) must be the same as the"
                     f" tensor rank ({tensor_data.ndim}).")
  bs = quantization_params.block_size
  if bs != 0 and tensor_data.shape[quantization_params.quantized_dimension] % bs != 0:
    raise ValueError("Tensor dimension must be divisible by block size. Got dimension:"
                     f" {tensor_data.shape[quantization_params.quantized_dimension]} and"
                     f" block size: {bs}")


# --------------------------------------------------------------------------
# host <-> device plumbing for the given-parameter kernels
# --------------------------------------------------------------------------

def _as_f32_exact(x: np.ndarray) -> np.ndarray:
  """The kernels read float32 tensor buffers (LiteRT weights). Other dtypes are
  accepted only when the conversion is lossless, so results equal the
  reference's computation in the wider type. A float32 tensor that already lives in HBM
  (runtime.HbmArray) is passed through: the kernels read it where it is."""
  if isinstance(x, rt.HbmArray) and x.dtype == np.float32:
    return x
  x = np.asarray(x)
  if x.dtype == np.float32:
    return x
  y = x.astype(np.float32)
  if not np.array_equal(y.astype(x.dtype), x, equal_nan=True):
    raise TypeError(f"mi355q quantizes float32 tensor buffers; got {x.dtype} data that is"
                    " not exactly representable in float32.")
  return y


def _channel_view(shape: Sequence[int], pshape: Sequence[int]) -> tuple[int, int, int]:
  """[outer, channels, inner] view for parameters of shape `pshape` broadcast
  over a tensor of `shape` (same rank)."""
  full = [i for i, (d, p) in enumerate(zip(shape, pshape)) if p != 1]
  for i in full:
    if pshape[i] != shape[i]:
      raise ValueError(f"scale shape {tuple(pshape)} does not broadcast to {tuple(shape)}")
  if not full:
    return 1, 1, int(np.prod(shape, dtype=np.int64))
  a, b = full[0], full[-1] + 1
  if any(shape[i] != 1 and pshape[i] == 1 for i in range(a, b)):
    raise ValueError(f"scale {tuple(pshape)} varies over dimensions of {tuple(shape)} that are not adjacent:"
                     " _adjacent_params first")
  return (int(np.prod(shape[:a], dtype=np.int64)), int(np.prod(shape[a:b], dtype=np.int64)),
          int(np.prod(shape[b:], dtype=np.int64)))


def _adjacent_params(shape: Sequence[int], scale: np.ndarray, zp: np.ndarray):
  """Parameters that vary over dimensions with a broadcast one in between (scale [A, 1, C] over a tensor [A, B, C]: the
  reference's arithmetic is NumPy broadcasting, ref :273-409, and takes any such shape; no op of its tables makes one) are
  repeated over the dimensions in between, so that they vary over ONE run of adjacent dimensions -- which is what the
  kernels' [outer, channels, inner] view addresses. The values every element meets are the same."""
  pshape = tuple(scale.shape)
  full = [i for i, p in enumerate(pshape) if p != 1]
  if len(pshape) != len(shape) or len(full) < 2:
    return scale, zp
  a, b = full[0], full[-1] + 1
  if not any(shape[i] != 1 and pshape[i] == 1 for i in range(a, b)):
    return scale, zp
  filled = tuple(shape[i] if a <= i < b else pshape[i] for i in range(len(shape)))
  scale = np.ascontiguousarray(np.broadcast_to(scale, filled))
  if zp is not None and np.size(zp) > 1:
    zp = np.ascontiguousarray(np.broadcast_to(zp, filled))
  return scale, zp


def _flat_params(scale: np.ndarray, zp: np.ndarray, compute64: bool):
  s = np.ascontiguousarray(scale.reshape(-1), dtype=np.float64 if compute64 else np.float32)
  z = np.broadcast_to(zp, scale.shape) if zp.size == 1 else zp
  z = np.ascontiguousarray(z.reshape(-1)).astype(np.int32)
  return s, z


def uniform_quantize(tensor_data: np.ndarray, quantization_params: qtyping.UniformQuantParams,
                     is_blockwise_quant: bool = False) -> np.ndarray:
  """q = cast(clip(rint(x / scale + zp))) on the GPU (ref :273-362)."""
  if not isinstance(tensor_data, rt.HbmArray):
    tensor_data = np.asarray(tensor_data)
  q = uniform_quantize_on_device(tensor_data, quantization_params, is_blockwise_quant)
  want = _get_numpy_dtype(IntType(quantization_params.num_bits, True))
  if q is None:
    return np.zeros(tensor_data.shape, want)
  out = rt.to_numpy(q).reshape(tensor_data.shape)
  return out if out.dtype == want else out.astype(want)


def uniform_quantize_on_device(tensor_data: np.ndarray, quantization_params: qtyping.UniformQuantParams,
                               is_blockwise_quant: bool = False, resident=None):
  """The launch behind `uniform_quantize`: same checks, the result stays in HBM (flat device
  tensor; None for an empty input). `resident` is `tensor_data` already uploaded as float32."""
  p = quantization_params
  block_view = None
  if is_blockwise_quant:
    if p.quantized_dimension is None:
      raise ValueError("Quantized dimension must be specified.")
    if p.block_size is None or p.block_size <= 0:
      raise ValueError("Block size must be specified and positive.")
    qd = p.quantized_dimension
    _get_tensor_shape_for_blockwise(tensor_data.shape, qd, p.block_size)  # divisibility check
    if any(d != 1 for d in tensor_data.shape[qd + 1:]):
      # blocks along an axis that is not the innermost one (ref :164-270 reshapes; no op of the reference's tables asks for it,
      # a direct caller may): the kernel's blocks are contiguous, so the blocked axis is moved last on the device -- elementwise
      # arithmetic, the same values in another place -- and the integers are moved back
      return _blockwise_along_inner_axis(tensor_data, p, qd, resident)
    block_view = (1, int(p.scale.size), p.block_size)
    # validation the reference performs on the broadcast parameters
    if p.scale.ndim != tensor_data.ndim:
      raise ValueError(f"Ranks of scales ({p.scale.ndim}) and zps ({np.ndim(p.zero_point)}) must"
                       f" be the same as the tensor rank ({tensor_data.ndim}).")
    zp = p.zero_point if (p.zero_point is not None and np.size(p.zero_point)) else \
        np.zeros(p.scale.shape, np.int32)
    scale = p.scale
  else:
    p = fix_quantization_params_rank(tensor_data, p)
    _is_valid_quantization_params(tensor_data, p)
    scale, zp = p.scale, p.zero_point
  if not np.issubdtype(zp.dtype, np.signedinteger):
    raise ValueError(f"zero_points need to be {np.signedinteger}. But the actual type is"
                     f" {zp.dtype}.")
  narrow = bool(p.symmetric and p.num_bits >= 8)
  if tensor_data.size == 0:
    return None
  compute64 = np.result_type(tensor_data.dtype, scale.dtype) == np.float64
  if block_view is None:
    scale, zp = _adjacent_params(tensor_data.shape, scale, zp)
  outer, ch, inner = block_view or _channel_view(tensor_data.shape, scale.shape)
  s, z = _flat_params(scale, zp, compute64)
  rt.require_gpu()
  x = resident if resident is not None else rt.to_device(_as_f32_exact(tensor_data))
  return ops.quantize(x, outer, ch, inner, rt.to_device(s), rt.to_device(z),
                      p.num_bits, narrow, zp_via_f64=zp.dtype.itemsize >= 4)


def _moved_last(p: qtyping.UniformQuantParams, qd: int, ndim: int) -> qtyping.UniformQuantParams:
  """`p` for the tensor with axis `qd` moved last (scales and zero points follow their axis)."""
  scale = np.ascontiguousarray(np.moveaxis(p.scale, qd, -1))
  zp = p.zero_point
  if zp is not None and np.size(zp) > 1:
    zp = np.ascontiguousarray(np.moveaxis(zp.reshape(p.scale.shape), qd, -1))
  return dataclasses.replace(p, scale=scale, zero_point=zp, quantized_dimension=ndim - 1)


def _blockwise_along_inner_axis(tensor_data, p: qtyping.UniformQuantParams, qd: int, resident):
  if tensor_data.size == 0:
    return None
  if p.scale.ndim != tensor_data.ndim:
    raise ValueError(f"Ranks of scales ({p.scale.ndim}) and zps ({np.ndim(p.zero_point)}) must"
                     f" be the same as the tensor rank ({tensor_data.ndim}).")
  shape = tuple(tensor_data.shape)
  x = resident if resident is not None else rt.to_device(_as_f32_exact(tensor_data))
  moved = x.reshape(shape).movedim(qd, -1).contiguous()
  q = uniform_quantize_on_device(_ShapeOnly(tuple(moved.shape), tensor_data.dtype),
                                 _moved_last(p, qd, len(shape)), True, resident=moved.reshape(-1))
  return q.reshape(tuple(moved.shape)).movedim(-1, qd).contiguous().reshape(-1)


class _ShapeOnly:
  """What uniform_quantize_on_device reads of `tensor_data` when the values are handed over as `resident`."""

  def __init__(self, shape, dtype):
    self.shape, self.dtype, self.ndim = shape, np.dtype(dtype), len(shape)
    self.size = int(np.prod(shape, dtype=np.int64))


def uniform_dequantize(tensor_data: np.ndarray,
                       quantization_params: qtyping.UniformQuantParams) -> np.ndarray:
  """(q - zp) * scale on the GPU (ref :365-409)."""
  tensor_data = np.asarray(tensor_data)
  p = quantization_params
  view = None
  if p.block_size != 0:
    qd = 1 if p.quantized_dimension == 0 else p.quantized_dimension  # b/443830202 (ref :379-387)
    sshape = list(tensor_data.shape)
    sshape[qd] //= p.block_size
    scale = p.scale.reshape(sshape)
    if any(d != 1 for d in tensor_data.shape[qd + 1:]):      # (see uniform_quantize_on_device: the blocked axis goes last and back)
      moved = np.ascontiguousarray(np.moveaxis(tensor_data, qd, -1))
      back = uniform_dequantize(moved, _moved_last(dataclasses.replace(p, scale=scale), qd, tensor_data.ndim))
      return np.ascontiguousarray(np.moveaxis(back, -1, qd))
    zp = p.zero_point if np.size(p.zero_point) else np.zeros(scale.shape, np.int32)
    if zp.size != 1 and zp.shape != scale.shape:
      zp = zp.reshape(scale.shape)
    view = (1, int(scale.size), p.block_size)
  else:
    p = fix_quantization_params_rank(tensor_data, p)
    _is_valid_quantization_params(tensor_data, p)
    scale, zp = p.scale, p.zero_point
  if tensor_data.dtype not in (np.int8, np.int16, np.int32):
    if np.issubdtype(tensor_data.dtype, np.integer) and np.abs(tensor_data).max(initial=0) < 2**31:
      tensor_data = tensor_data.astype(np.int32)
      zp = zp.astype(np.int32) if zp.dtype.itemsize < 4 else zp
    else:
      raise TypeError(f"uniform_dequantize expects int8/int16/int32 data, got {tensor_data.dtype}")
  diff = np.result_type(tensor_data.dtype, zp.dtype)
  diff_bits = min(32, diff.itemsize * 8)
  if view is None:
    scale, zp = _adjacent_params(tensor_data.shape, scale, zp)
  outer, ch, inner = view or _channel_view(tensor_data.shape, scale.shape)
  s, z = _flat_params(scale, zp, False)
  rt.require_gpu()
  out = ops.dequantize(rt.to_device(tensor_data), outer, ch, inner, rt.to_device(s),
                       rt.to_device(z), diff_bits)
  res = rt.to_numpy(out).reshape(tensor_data.shape)
  want = np.result_type(diff, scale.dtype)
  return res if res.dtype == want else res.astype(want)


def symmetric_quantize_bias_tensor(
    bias_content: np.ndarray, input_tensor_quant_params: qtyping.UniformQuantParams,
    weight_tensor_quant_params: qtyping.UniformQuantParams, check_error: bool = False,
) -> qtyping.UniformQuantParams:
  """int32 (int64 for 16-bit activations) bias with scale = s_in * s_w (ref :412-489)."""
  eff = np.squeeze(input_tensor_quant_params.scale * weight_tensor_quant_params.scale)
  if not eff.shape:
    eff = np.expand_dims(eff, axis=0)
  zp = np.zeros_like(eff, dtype=np.int32)
  qdim = None if len(eff) == 1 else 0
  params = qtyping.UniformQuantParams(scale=eff, zero_point=zp, num_bits=32, symmetric=True,
                                      quantized_dimension=qdim)
  q = uniform_quantize(bias_content, params)
  if check_error:
    deq = uniform_dequantize(q, params)
    err = np.max(np.abs(deq - bias_content))
    tol = np.maximum(1e-6, np.max(eff))
    if err > tol:
      raise ValueError("Quantization error is too large for bias tensor quantization. Max"
                       f" quantization error is {err}, which exceed the threshold {tol}")
  bits = 32
  if input_tensor_quant_params.num_bits == 16:
    q, bits = q.astype(np.int64), 64
  return qtyping.UniformQuantParams(scale=eff, zero_point=zp, num_bits=bits,
                                    quantized_dimension=qdim, symmetric=True, quantized_data=q)


def tensor_zp_scale_from_min_max(min_value, max_value, num_bits: int, symmetric: bool,
                                 granularity: qtyping.QuantGranularity,
                                 clipping_values: Optional[np.ndarray] = None):
  """Zero point and scale from min / max (ref :492-586). Host math on O(#scales) values."""
  qmin, qmax = get_quantized_range(IntType(num_bits, True))
  floor = 1e-9
  pos = clipping_values
  neg = None if clipping_values is None else -clipping_values
  blockwise = is_blockwise(granularity)
  if blockwise:
    cap_hi = np.broadcast_to(np.array(65280) * (2**num_bits - 1), np.shape(max_value))
    cap_lo = np.broadcast_to(np.array(-65280) * (2**num_bits), np.shape(min_value))
    pos = cap_hi if pos is None else np.minimum(pos, cap_hi)
    neg = cap_lo if neg is None else np.maximum(neg, cap_lo)
  if symmetric:
    bound = np.maximum(np.maximum(np.abs(min_value), np.abs(max_value)), floor)
    if clipping_values is not None:
      bound = np.clip(bound, neg, pos)
    scale = bound / qmax
    zp = np.zeros_like(scale, dtype=np.int32)
  else:
    hi = np.maximum(max_value, np.zeros_like(max_value))
    lo = np.minimum(min_value, np.zeros_like(min_value))
    bound = np.maximum(hi - lo, floor)
    if clipping_values is not None:
      bound = np.clip(bound, -clipping_values, clipping_values)
    scale = bound / (qmax - qmin)
    zp = np.rint(qmin - lo / scale)
  if blockwise:
    scale = round_to_bf16(scale).astype(np.float16).astype(np.float32)
  return assign_quantized_type(zp, IntType(num_bits, True)), scale
