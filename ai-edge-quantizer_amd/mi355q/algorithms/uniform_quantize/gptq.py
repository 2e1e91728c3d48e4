"""GPTQ (Hessian-aware weight quantization), GPU backed.

Mirror of ref: algorithms/uniform_quantize/gptq.py. calibrate() builds
H = (2/num_samples) X^T X with the FP32 MFMA GEMM, _prepare_hessian_inverse runs
the blocked FP64 Cholesky / inverse on the GPU and _apply_gptq the column-serial
OBS update (mi355q_gptq_*). QSVs stay NumPy dictionaries at this interface, as in
the reference; the min/max up-front scales are computed exactly as there.
"""
from __future__ import annotations

import dataclasses
from collections.abc import Mapping, MutableMapping, Sequence
from typing import Any

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ..utils import common_utils
from . import common_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "GPTQ"


def calibrate(tfl_op, graph_info: qtyping.GraphInfo,
              tensor_content_map: MutableMapping[str, np.ndarray],
              inputs_to_ignore: Sequence[int] | None = None,
              outputs_to_ignore: Sequence[int] | None = None,
              valid_range: tuple[float, float] = (-3e38, 3e38)) -> dict[str, qtyping.QSV]:
  """min/max + Hessian of every runtime tensor of the op (ref :55-108)."""
  lo, hi = valid_range
  out = {}
  for tid in common_quantize.get_tensor_indices_requiring_calibration(
      tfl_op, graph_info, inputs_to_ignore, outputs_to_ignore):
    res = common_quantize.collect_activation_tensor_statistics(
        tid, graph_info, tensor_content_map, valid_float_range_min=lo, valid_float_range_max=hi)
    if res is None:
      continue
    name, content, qsv = res
    qsv["hessian"] = hessian_of(content, qsv["num_samples"])
    out[name] = qsv
  return out


def hessian_of(tensor_content: np.ndarray, num_samples):
  """(2.0 / num_samples) * x.T.dot(x), x = content.reshape(-1, last) (ref :100-107).

  float32 content: X^T X on the FP32 MFMA units, scaled into float64 (NumPy's
  promotion of `2.0 / np.array(n)`). float64 content (the reference's own test
  feeds 1e39): the FP64 MFMA GEMM.
  """
  alpha = 2.0 / np.asarray(num_samples)
  rt.require_gpu()
  rec = rt.staged(tensor_content)        # the calibrator put this sample's activations in HBM
  if rec is not None and tensor_content.dtype == np.float32:
    xd = rec["dev"].reshape(-1, tensor_content.shape[-1])
    return rt.HbmArray(ops.gptq_xtx(xd, float(alpha)))
  x = np.ascontiguousarray(tensor_content.reshape([-1, tensor_content.shape[-1]]))
  if x.dtype == np.float32:   # stays in HBM: merged per sample and consumed by the GPU again
    return rt.HbmArray(ops.gptq_xtx(rt.to_device(x), float(alpha)))
  if x.dtype == np.float64:
    xd = rt.to_device(x)
    return alpha * rt.to_numpy(ops.gemm(xd, xd, trans_a=True))
  raise TypeError(f"GPTQ calibration expects float32 / float64 activations, got {x.dtype}")


def _prepare_hessian_inverse(hessian: np.ndarray, damp_factor: float = 0.01) -> np.ndarray:
  """Damped inverse through Cholesky; float32 result (ref :111-128)."""
  hinv, info = _device_hessian_inverse(hessian, damp_factor)
  if int(info.item()) != 0:
    raise np.linalg.LinAlgError("Matrix is not positive definite")
  return rt.to_numpy(hinv)


def _device_hessian_inverse(hessian, damp_factor: float = 0.01):
  """(hinv float32 [d,d], info) on the device. A Hessian that lives in HBM remembers its
  inverse: q / k / v (and gate / up) share one input activation, hence one Hessian."""
  rt.require_gpu()
  if isinstance(hessian, rt.HbmArray):
    key = ("hinv", float(damp_factor))
    if key not in hessian.cache:
      hessian.cache[key] = ops.gptq_hinv(rt.on_device(hessian, torch_f64()), damp_factor)
    return hessian.cache[key]
  return ops.gptq_hinv(rt.to_device(np.ascontiguousarray(hessian, dtype=np.float64)), damp_factor)


def torch_f64():
  import torch
  return torch.float64


def _apply_gptq(tensor_content: np.ndarray, quant_params: qtyping.UniformQuantParams,
                activation_tensor_qsv: Mapping[str, Any],
                tensor_quant_config: qtyping.TensorQuantizationConfig,
                blocksize: int = 64) -> qtyping.UniformQuantParams:
  """Blocked OBS update + column-serial quantization (ref :131-216)."""
  if blocksize != 64:
    raise NotImplementedError("the GPU kernel is built for the reference's blocksize of 64")
  if tensor_quant_config.num_bits > 8:
    raise NotImplementedError("GPTQ kernel supports <= 8 bit targets")
  if tensor_content.ndim != 2 or tensor_content.dtype != np.float32:
    raise TypeError("GPTQ expects a 2-D float32 weight")
  rt.require_gpu()
  rows, d = tensor_content.shape
  hinv, info = _device_hessian_inverse(activation_tensor_qsv["hessian"], 0.01)
  scale, zp = quant_params.scale, quant_params.zero_point
  blockwise = uniform_quantize_tensor.is_blockwise(tensor_quant_config.granularity)
  if blockwise:
    mode, bs = 2, quant_params.block_size
  elif scale.size == 1:
    mode, bs = 0, 0
  elif scale.size == rows:
    mode, bs = 1, 0
  else:
    raise NotImplementedError(f"scale shape {scale.shape} for a [{rows}, {d}] weight")
  if not np.issubdtype(zp.dtype, np.signedinteger):
    raise ValueError(f"zero_points need to be {np.signedinteger}. But the actual type is"
                     f" {zp.dtype}.")
  sdt = np.float64 if scale.dtype == np.float64 else np.float32
  s_dev = rt.to_device(np.ascontiguousarray(scale.reshape(-1), dtype=sdt))
  # all-zero zero points (symmetric recipes) are passed as "none": same results, and the block
  # kernel then runs its specialised form
  z_dev = (rt.to_device(np.ascontiguousarray(np.broadcast_to(zp, scale.shape).reshape(-1)).astype(np.int32))
           if np.any(zp) else None)
  narrow = bool(quant_params.symmetric and quant_params.num_bits >= 8)
  diff_bits = min(32, np.result_type(np.int8, zp.dtype).itemsize * 8)
  q = ops.gptq_apply(rt.to_device(tensor_content), hinv, s_dev, z_dev, mode, bs,
                     quant_params.num_bits, narrow, zp.dtype.itemsize >= 4, diff_bits)
  if int(info.item()) != 0:
    raise np.linalg.LinAlgError("Matrix is not positive definite")
  return dataclasses.replace(quant_params, quantized_data=rt.quantized_result(
      q, quant_params.num_bits, tensor_content.nbytes, tensor_content.shape))


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: np.ndarray | None = None, tensor_qsv: Mapping[str, Any] | None = None,
) -> qtyping.UniformQuantParams:
  """ref :219-300."""
  cfg = tensor_quant_config
  act_qsv = tensor_qsv.get("activation_tensor_qsv") if tensor_qsv else None
  if (act_qsv is not None and "hessian" in act_qsv and isinstance(tensor_content, np.ndarray)
      and tensor_content.dtype == np.float32 and tensor_content.ndim == 2):
    # one upload serves both the min / max below and the update (a Gemma-2B layer is 440 MB)
    rt.require_gpu()
    tensor_content = rt.HbmArray(rt.to_device(tensor_content))
  if tensor_qsv is None or "min" not in tensor_qsv:
    if tensor_content is None:
      raise ValueError(
          f"{op_info.op_name}(index: {op_info.subgraph_op_index}) not found in"
          " tensor_name_to_qsv. Check if the correct calibration results are passed into the"
          " ParamsGenerator.")
    tensor_min_max = common_quantize.init_tensor_min_max(tensor_content, op_info)
  else:
    tensor_min_max = tensor_qsv
  if "min" not in tensor_min_max or "max" not in tensor_min_max:
    raise ValueError(
        "min and max must be provided to produce tensor quantization parameters. Check if the"
        " correct calibration results are passed into the ParamsGenerator.")
  zp, scale = uniform_quantize_tensor.tensor_zp_scale_from_min_max(
      tensor_min_max["min"], tensor_min_max["max"], cfg.num_bits, cfg.symmetric, cfg.granularity,
      None)
  params = qtyping.UniformQuantParams(
      scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=cfg.symmetric,
      quantized_dimension=common_utils.get_weight_quantized_dim(op_info, tensor_content,
                                                               cfg.granularity),
      block_size=uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity))
  if tensor_content is None or act_qsv is None or "hessian" not in act_qsv:
    return params
  return _apply_gptq(tensor_content, params, act_qsv, cfg)
