"""Recovery of the integer weights behind fake-quantized (QAT, "dequantized") FP32 weights, GPU
backed.

Mirror of ref: algorithms/uniform_quantize/dequantized_weight_recovery.py. A tensor that is
q * scale for integers q has, per quantization group, a smallest positive step between its
sorted magnitudes (0 included) equal to the scale; the tensor is then re-quantized with that
scale and the result is checked to reproduce the input within 1e-4.

The per-group sort + minimum step (mi355q_dwr_scales_f32), the quantization and the FP64
max-error check (mi355q_dwr_max_error_f32) run on the GPU; groups are contiguous for the ops
the algorithm is registered for (FULLY_CONNECTED, CONV_2D, EMBEDDING_LOOKUP: rows, or blocks
along the last axis).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ..utils import common_utils
from . import naive_min_max_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "dequantized_weight_recovery"


def get_blockwise_shape(shape, quantized_dimension: int, block_size: int) -> tuple[int, ...]:
  """Scale shape of a blockwise quantized tensor (ref: algorithms/utils/common_utils.py:1221-1247)."""
  out = list(shape)
  if out[quantized_dimension] % block_size != 0:
    raise ValueError(f"Dimension {out[quantized_dimension]} along axis {quantized_dimension} is not"
                     f" divisible by block size {block_size}")
  out[quantized_dimension] //= block_size
  return tuple(out)


def get_zp_scale_from_dequantized_symmetric_weights(
    dequant_vals: np.ndarray, quantized_dimension: Optional[int] = None, block_size: int = 0,
    min_scale: float = 1e-9, resident=None) -> tuple[np.ndarray, np.ndarray]:
  """(zero points, scales) of symmetric fake-quantized weights (ref :118-186). `resident` is
  `dequant_vals` already in HBM (the caller uploads a weight once for all three kernels)."""
  if quantized_dimension not in (0, 1, None):
    raise ValueError(f"quantized_dimension must be 0, 1, or None. Got {quantized_dimension}")
  if min_scale != 1e-9:
    raise NotImplementedError("the GPU kernel is built for the reference's min_scale of 1e-9")
  rt.require_gpu()
  vals = dequant_vals if isinstance(dequant_vals, rt.HbmArray) else np.asarray(dequant_vals)
  if vals.dtype != np.float32:
    raise TypeError(f"dequantized weight recovery expects float32 weights, got {vals.dtype}")
  last = vals.shape[-1] if vals.ndim else 1
  dev = resident if resident is not None else rt.to_device(vals)
  flat = dev.reshape(-1, last)
  if quantized_dimension is None:       # one group, float64 arithmetic (np.append(arr, 0) promotes)
    scale = rt.to_numpy(ops.dwr_scales(flat, vals.size, rounded=False)).reshape(1, 1)
  elif block_size > 0:
    if quantized_dimension != vals.ndim - 1:
      raise NotImplementedError("GPU path: blocks along the last axis only")
    shape = get_blockwise_shape(vals.shape, quantized_dimension, block_size)
    scale = rt.to_numpy(ops.dwr_scales(flat, block_size, rounded=True)).astype(np.float32).reshape(shape)
  else:
    if quantized_dimension != 0:
      raise NotImplementedError("GPU path: channelwise groups along axis 0 only")
    rows = vals.shape[0]
    shape = (rows,) + (1,) * (vals.ndim - 1)
    scale = rt.to_numpy(ops.dwr_scales(dev.reshape(rows, -1), vals.size // rows,
                                       rounded=True)).astype(np.float32).reshape(shape)
  return np.zeros_like(scale, dtype=np.int32), scale


def _check_unique_values(tensor_content: np.ndarray, quantized_dimension, *, block_size: int,
                         num_bits: int) -> tuple[int, int]:
  """Largest number of distinct values in any group, and the limit (ref :64-115). Only runs to
  word the error message: host NumPy."""
  limit = 1 << num_bits
  if block_size > 0:
    groups = tensor_content.reshape(-1, block_size)
  elif quantized_dimension is not None:
    groups = np.moveaxis(tensor_content, quantized_dimension, 0).reshape(tensor_content.shape[quantized_dimension], -1)
  else:
    return np.unique(tensor_content).size, limit
  most = 0
  for row in groups:
    most = max(most, np.unique(row).size)
    if most > limit:
      break
  return most, limit


def get_tensor_quant_params(op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
                            tensor_content: Optional[np.ndarray] = None,
                            tensor_qsv: Optional[dict[str, Any]] = None) -> qtyping.UniformQuantParams:
  """ref :189-283."""
  if tensor_content is None:
    return naive_min_max_quantize.get_tensor_quant_params(op_info, tensor_quant_config, tensor_content,
                                                          tensor_qsv)
  cfg = tensor_quant_config
  blockwise = uniform_quantize_tensor.is_blockwise(cfg.granularity)
  block_size = uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity) if blockwise else 0
  if not cfg.symmetric:
    raise ValueError("Only symmetric weights are supported for dequantized weight recovery.")
  quantized_dim = common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity)
  rt.require_gpu()
  if tensor_content.dtype != np.float32:
    raise TypeError(f"dequantized weight recovery expects float32 weights, got {tensor_content.dtype}")
  resident = rt.to_device(tensor_content)
  zp, scale = get_zp_scale_from_dequantized_symmetric_weights(
      dequant_vals=tensor_content, quantized_dimension=quantized_dim, block_size=block_size,
      resident=resident)
  params = qtyping.UniformQuantParams(scale=scale, zero_point=zp, num_bits=cfg.num_bits,
                                      symmetric=cfg.symmetric, quantized_dimension=quantized_dim,
                                      block_size=block_size)
  q = uniform_quantize_tensor.uniform_quantize_on_device(tensor_content, params, blockwise,
                                                         resident=resident)
  if q is None:
    return dataclasses.replace(params, quantized_data=uniform_quantize_tensor.uniform_quantize(
        tensor_content, params, is_blockwise_quant=blockwise))
  if not op_info.op_quant_config.skip_checks:
    group = tensor_content.size // scale.size
    worst = ops.dwr_max_error(resident.reshape(-1), q.reshape(-1),
                              ops._f64_dev(scale.reshape(-1)), group)  # pylint: disable=protected-access
    if worst > 1e-4:
      base = ("Failed to recover the original quantized values from dequantized values. Max diff"
              f" between recovered and original values: {worst} (tolerance: 0.0001)")
      most, limit = _check_unique_values(tensor_content, quantized_dim, block_size=block_size,
                                         num_bits=cfg.num_bits)
      if most > limit:
        extra = (f"Detected a quantization group with {most} unique values, which exceeds the limit of"
                 f" {limit} for {cfg.num_bits}-bit quantization. This suggests the input tensor is NOT"
                 " dequantized (fake-quantized) weights. Please verify if you are using a QAT"
                 " checkpoint.")
      else:
        extra = (f"Max unique values in any group is {most} (limit: {limit}). The recovery failed"
                 " despite reasonable unique value count. Check if the weights are symmetric or if"
                 " tolerance is too tight.")
      raise RuntimeError(f"Failed to recover weights. Original error: {base}. {extra}")
  return dataclasses.replace(params, quantized_data=rt.quantized_result(
      q, cfg.num_bits, tensor_content.nbytes, tensor_content.shape))


def calibrate(tfl_op: Any, graph_info: qtyping.GraphInfo, tensor_content_map: dict[str, np.ndarray],
              inputs_to_ignore: Optional[list[int]] = None,
              outputs_to_ignore: Optional[list[int]] = None) -> dict[str, qtyping.QSV]:
  """ref :286-312."""
  return naive_min_max_quantize.min_max_calibrate(tfl_op, graph_info, tensor_content_map,
                                                  inputs_to_ignore, outputs_to_ignore)


def init_qsvs(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
              inputs_to_ignore: Optional[list[int]] = None,
              outputs_to_ignore: Optional[list[int]] = None) -> qtyping.QSV:
  """ref :315-362."""
  return naive_min_max_quantize.init_qsvs(op_info, graph_info, inputs_to_ignore, outputs_to_ignore)
