"""MSE-optimal scale (scale = k * sqrt(mean(x^2))), GPU backed.

Mirror of ref: algorithms/uniform_quantize/mse.py. The row reduction runs in
mi355q_mse_scale_f32 in NumPy's pairwise order (bit-identical scales).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ..utils import common_utils
from . import naive_min_max_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "MSE"

# ref :30-33 ("Coefficients from offline numeric analysis")
_MSE_QUANT_MULS = {8: 0.05408, 4: 0.37755}


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: Optional[np.ndarray] = None, tensor_qsv: Optional[dict[str, Any]] = None,
) -> qtyping.UniformQuantParams:
  """ref :36-128."""
  cfg = tensor_quant_config
  if uniform_quantize_tensor.is_blockwise(cfg.granularity):
    raise ValueError("Blockwise quantization is not supported for MSE quantization.")
  if tensor_content is None:
    return naive_min_max_quantize.get_tensor_quant_params(op_info, cfg, tensor_content, tensor_qsv)
  if not cfg.symmetric:
    raise ValueError(f"Unsupported symmetry: {cfg.symmetric}. MSE supports symmetric"
                     " quantization only for now.")
  if not tensor_qsv or "min" not in tensor_qsv:
    # the reference collects (and then ignores) min/max here; an unset weight
    # config makes that {} and raises below, which we preserve.
    if op_info.op_quant_config.weight_tensor_config is None:
      raise ValueError(
          "min and max must be provided to produce tensor quantization parameters. Check if"
          " the correct calibration results are passed into the ParamsGenerator.")
  elif "max" not in tensor_qsv:
    raise ValueError(
        "min and max must be provided to produce tensor quantization parameters. Check if the"
        " correct calibration results are passed into the ParamsGenerator.")
  quantized_dim = common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity)
  if cfg.num_bits not in _MSE_QUANT_MULS:
    raise KeyError(cfg.num_bits)
  shape = tensor_content.shape
  if quantized_dim is None:
    outer, units, out_shape = 1, 1, (1,) * tensor_content.ndim
  else:
    outer = int(np.prod(shape[:quantized_dim], dtype=np.int64))
    units = shape[quantized_dim]
    out_shape = tuple(d if i == quantized_dim else 1 for i, d in enumerate(shape))
  x = uniform_quantize_tensor._as_f32_exact(tensor_content)  # pylint: disable=protected-access
  if x.size == 0:
    raise ValueError("MSE quantization of an empty tensor")
  inner = x.size // (outer * units)
  rt.require_gpu()
  xd = rt.to_device(x.reshape(-1))
  if outer == 1:     # rows of a weight: scale and integers in one launch (the unit is still in the L2 when it is quantized)
    scale_d, q = ops.mse_requant(xd, units, inner, _MSE_QUANT_MULS[cfg.num_bits], cfg.num_bits, cfg.num_bits >= 8)
  else:
    scale_d = ops.mse_scale_nd(xd, outer, units, inner, _MSE_QUANT_MULS[cfg.num_bits])
    q = ops.quantize(xd, outer, units, inner, scale_d, None, cfg.num_bits, cfg.num_bits >= 8,
                     zp_via_f64=True)
  # a large weight's results stay in HBM (the integers for the model writer, the scales so that
  # the call does not wait for its own kernels); NumPy consumers get host copies on demand
  scale = (rt.HbmArray(scale_d.reshape(out_shape)) if tensor_content.nbytes >= rt.KEEP_IN_HBM_BYTES
           else rt.to_numpy(scale_d).reshape(out_shape))
  return qtyping.UniformQuantParams(
      scale=scale, zero_point=np.zeros(scale.shape, np.int32), num_bits=cfg.num_bits,
      symmetric=cfg.symmetric, quantized_dimension=quantized_dim, block_size=0,
      quantized_data=rt.quantized_result(q, cfg.num_bits, tensor_content.nbytes, tensor_content.shape))
