"""OCTAV (optimally clipped tensors and vectors) scale search, GPU backed.

Mirror of ref: algorithms/uniform_quantize/octav.py. The Newton iteration for the
clipping constants runs in mi355q_octav_clip_f32 with NumPy's float32 summation
order reproduced exactly, then the fused requant kernel applies
bound = clip(max|x|, -c, c) -> scale -> quantize in one more pass.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ..utils import common_utils
from . import common_quantize
from . import naive_min_max_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "OCTAV"


def _unit_view(x: np.ndarray, axis):
  """(outer, channels, inner) such that reducing over `axis` reduces x viewed as
  [outer, channels, inner] over its first and last axis; None when no such view exists
  (two kept axes separated by a reduced one)."""
  if axis is None:
    return 1, 1, int(x.size)
  axis = (axis,) if isinstance(axis, int) else tuple(axis)
  keep = [d for d in range(x.ndim) if d not in axis and x.shape[d] != 1]
  reduced = [d for d in axis if x.shape[d] != 1]
  if not keep or not reduced or max(keep) < min(reduced):   # units are contiguous runs
    units = int(np.prod([x.shape[d] for d in keep], dtype=np.int64)) if keep else 1
    return 1, units, int(x.size // max(units, 1))
  if any(d in reduced for d in range(keep[0], keep[-1])):
    return None
  channels = int(np.prod([x.shape[d] for d in keep], dtype=np.int64))
  outer = int(np.prod(x.shape[:keep[0]], dtype=np.int64))
  return outer, channels, int(x.size // (outer * channels))


def _guess_clipping_with_octav(x: np.ndarray, bits: int, axis, max_iterations: int,
                               exponent_divisor: float, early_stop: bool = True) -> np.ndarray:
  """Clipping constants with shape of x reduced over `axis` (keepdims). ref :30-112."""
  x = np.asarray(x)
  if axis is not None:
    ax = (axis,) if isinstance(axis, int) else tuple(axis)
    reduced = tuple(1 if k in ax else d for k, d in enumerate(x.shape))
  else:
    # the reference starts from shape (1,) but np.sum(..., axis=None, keepdims=True)
    # turns the guess into (1,)*ndim on the first iteration
    reduced = (1,) * x.ndim if max_iterations > 0 and x.ndim > 0 else (1,)
  view = _unit_view(x, axis)
  if x.size == 0:
    return np.ones(reduced, dtype=np.float32)
  xf = uniform_quantize_tensor._as_f32_exact(x)  # pylint: disable=protected-access
  rt.require_gpu()
  xd = rt.to_device(xf.reshape(-1))
  if view is None:
    # Two kept axes with a reduced one in between (x [A, R, C] reduced over axis 1; the reference's np.sum takes any axis
    # tuple, ref :55-61, no op of its tables makes one -- weights are reduced over whole trailing or leading axes). The
    # kernels address [outer, channels, inner]; here the reduced axes are moved last on the device and each unit summed as
    # one contiguous run. That is NumPy's pairwise order for a contiguous run, not the order NumPy walks the original
    # layout in: the clipping constants are within the tolerance class SURVEY 7 gives OCTAV (T2, 1e-6 relative), not
    # bit-exact like every layout above (tests/test_gpu_algorithms.py::test_octav_kept_axes_separated_by_a_reduced_one).
    kept = [d for d in range(x.ndim) if d not in ax]
    moved = xd.reshape(tuple(x.shape)).permute(kept + [d for d in range(x.ndim) if d in ax]).contiguous()
    units = int(np.prod([x.shape[d] for d in kept], dtype=np.int64))
    clip, _ = ops.octav_clip(moved.view(-1), units, int(x.size // units), bits, max_iterations, exponent_divisor,
                             early_stop, axis_given=True)
    return rt.to_numpy(clip).reshape(reduced)
  if view[0] == 1:
    clip, _ = ops.octav_clip(xd, view[1], view[2], bits, max_iterations, exponent_divisor,
                             early_stop, axis_given=axis is not None)
  else:
    clip, _ = ops.octav_clip_nd(xd, view[0], view[1], view[2], bits, max_iterations,
                                exponent_divisor, early_stop)
  return rt.to_numpy(clip).reshape(reduced)


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: Optional[np.ndarray] = None, tensor_qsv: Optional[dict[str, Any]] = None,
) -> qtyping.UniformQuantParams:
  """ref :115-227."""
  cfg = tensor_quant_config
  if tensor_content is None:  # activations: plain min/max parameters
    return naive_min_max_quantize.get_tensor_quant_params(op_info, cfg, tensor_content, tensor_qsv)
  if not cfg.symmetric:
    raise ValueError(f"Unsupported symmetry: {cfg.symmetric}. OCTAV supports symmetric"
                     " quantization only for now.")
  have_qsv = bool(tensor_qsv) and "min" in tensor_qsv
  quantized_dim = common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity)
  blockwise = uniform_quantize_tensor.is_blockwise(cfg.granularity)
  block_size = uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity)
  if blockwise:
    data, axes = uniform_quantize_tensor.reshape_data_for_blockwise(
        tensor_content, op_info.op_name, cfg.granularity)
  else:
    data, axes = tensor_content, common_utils.get_reduce_dims(quantized_dim, tensor_content.shape)

  # ---- fused path: clip search kernel + one requant pass, weight stays in HBM ----
  weight_cfg = op_info.op_quant_config.weight_tensor_config
  layout = None
  if (not have_qsv and cfg.num_bits in (2, 4, 8) and weight_cfg is not None
      and weight_cfg.granularity == cfg.granularity):
    layout = naive_min_max_quantize.fused_weight_layout(tensor_content, cfg.granularity,
                                                        quantized_dim)
  if layout is not None:
    rows, cols, block = layout
    rt.require_gpu()
    # one upload (none for a weight that is in HBM already); clip search, quantize and pack all
    # read the resident copy, and large results stay in HBM for the model writer
    resident = (tensor_content if isinstance(tensor_content, rt.HbmArray)
                else rt.HbmArray(rt.to_device(tensor_content)))
    x = rt.to_device(resident.reshape(rows, cols))
    units, unit_len = (rows * (cols // block), block) if block else (rows, cols)
    clip, _ = ops.octav_clip(x.view(-1), units, unit_len, cfg.num_bits, 10, 3.0, True, True)
    scale, q = naive_min_max_quantize.fused_symmetric_requant(resident, layout, cfg.num_bits, clip=clip)
    scale = scale.reshape(
        naive_min_max_quantize.scale_shape_for(tensor_content, cfg.granularity, quantized_dim))
    return qtyping.UniformQuantParams(
        scale=scale, zero_point=np.zeros(scale.shape, np.int8), num_bits=cfg.num_bits,
        symmetric=True, quantized_dimension=quantized_dim, block_size=block_size,
        quantized_data=q)

  # ---- general path (TENSORWISE, supplied min/max, other layouts) ----
  if have_qsv:
    tensor_min_max = tensor_qsv
  else:
    tensor_min_max = common_quantize.init_tensor_min_max(tensor_content, op_info)
  if "min" not in tensor_min_max or "max" not in tensor_min_max:
    raise ValueError(
        "min and max must be provided to produce tensor quantization parameters. Check if the"
        " correct calibration results are passed into the ParamsGenerator.")
  clip = _guess_clipping_with_octav(data, cfg.num_bits, axes, max_iterations=10,
                                    exponent_divisor=3.0 if cfg.symmetric else 12.0)
  if blockwise:
    clip = clip.reshape(tensor_min_max["min"].shape)
  zp, scale = uniform_quantize_tensor.tensor_zp_scale_from_min_max(
      tensor_min_max["min"], tensor_min_max["max"], cfg.num_bits, cfg.symmetric,
      cfg.granularity, clip)
  params = qtyping.UniformQuantParams(
      scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=cfg.symmetric,
      quantized_dimension=quantized_dim, block_size=block_size)
  q = uniform_quantize_tensor.uniform_quantize(tensor_content, params, is_blockwise_quant=blockwise)
  return dataclasses.replace(params, quantized_data=q)
