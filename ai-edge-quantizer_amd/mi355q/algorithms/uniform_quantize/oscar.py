"""OSCAR: activation-aware channel scaling + optimal clipping, GPU backed.

Mirror of ref: algorithms/uniform_quantize/oscar.py (FULLY_CONNECTED only). calibrate() adds the
per-input-channel second moment `mu2` to the min/max QSVs; materialize_fully_connected() scales
the weight columns by s, quantizes W*s with per-group optimal clip bounds and asks for an
elementwise MUL by 1/s on the activation (INSERT_MULTIPLY).

Everything that touches the [out_ch, in_ch] matrix runs on the GPU in the reference's FP64
arithmetic and NumPy's summation orders (csrc/oscar.hip): column energies, per-group maxima /
winners, the sort + running-sum breakpoint scan and the final quantization. The O(in_ch) vector
algebra in between (geometric-mean normalisation through log/exp, clamps, group masses) is host
NumPy, the very calls the reference makes, so both sides round identically.
"""
from __future__ import annotations

import logging
from collections.abc import MutableMapping, Sequence
from typing import Any

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ...utils import tfl_flatbuffer_utils
from ..utils import common_utils
from . import common_quantize
from . import naive_min_max_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "OSCAR"
_Op = qtyping.TFLOperationName
_T = qtyping.QuantTransformation

_EPS = 1e-12
_SCALE_CLAMP = (1e-4, 1e4)


def _floor_positive(mu2: np.ndarray) -> np.ndarray:
  """Dead channels get a tiny positive mass (ref :56-59)."""
  mu2 = np.asarray(mu2, np.float64)
  return np.maximum(mu2, float(np.max(mu2)) * 1e-8 + _EPS)


def _check_fc(op_name, w) -> None:
  if op_name != _Op.FULLY_CONNECTED:
    raise ValueError(f"OSCAR supports FULLY_CONNECTED only, got: {op_name}")
  if np.ndim(w) != 2:
    raise ValueError(f"OSCAR expects 2-D weights for {op_name}, got {np.shape(w)}")


def _columns_of(op_name, w, mu2) -> np.ndarray:
  """One mass per weight column: mu2, or ones for the weight-only fallback (ref :111-153)."""
  in_ch = np.shape(w)[1]
  if mu2 is None:
    return np.ones(in_ch)
  col = np.asarray(mu2, np.float64).ravel()
  if col.size != in_ch:
    raise ValueError(
        f"OSCAR: activation mu2 has {col.size} channels but {op_name} weights of shape"
        f" {np.shape(w)} expect {in_ch}. The calibration statistics do not match this tensor.")
  return col


def _weight_on_device(w):
  """float32 [n, d] device copy. The reference widens to FP64 first; float32 -> FP64 is exact, so
  the kernels widen element by element instead."""
  if isinstance(w, rt.HbmArray) and w.dtype == np.float32:
    return w.device_tensor.contiguous()
  w = np.asarray(w)
  if w.dtype == np.float64:       # accepted when it is a widened float32 tensor (the model's dtype)
    narrow = w.astype(np.float32)
    if not np.array_equal(narrow.astype(np.float64), w):
      raise TypeError("OSCAR GPU path expects float32 weights (float64 values that are not"
                      " float32-representable were passed)")
    w = narrow
  if w.dtype != np.float32:
    raise TypeError(f"OSCAR GPU path expects float32 weights, got {w.dtype}")
  return rt.to_device(w)


def _group_width(d: int, block_size: int) -> int:
  return block_size if (block_size and d % block_size == 0) else d


class _Objective:
  """The proxy objective of ref :175-194 for one weight; remembers the winners of its last
  evaluation (the fixed-point step of ref :236-246 needs them for the same scales)."""

  def __init__(self, wd, mu2: np.ndarray, block_size: int):
    self.wd, self.mu2 = wd, mu2
    self.d = wd.shape[1]
    self.g = _group_width(self.d, block_size)
    if self.g != self.d and self.g not in (32, 64, 128, 256):
      raise NotImplementedError(f"OSCAR GPU path: block size {self.g}")
    self.winner = self.wsq = None

  def __call__(self, s: np.ndarray) -> float:
    sums, self.winner, self.wsq = ops.oscar_group_terms(self.wd, ops._f64_dev(s), self.g)  # pylint: disable=protected-access
    sums = rt.to_numpy(sums)
    m = self.mu2 / (s * s)
    total = 0.0
    for k in range(self.d // self.g):
      total += float(sums[k]) * float(m[k * self.g:(k + 1) * self.g].sum())
    return total

  def winner_energy(self) -> np.ndarray:
    return rt.to_numpy(ops.oscar_winner_energy(self.winner, self.wsq, self.d, self.g))


def _channel_scale_objective(w, s: np.ndarray, mu2: np.ndarray, block_size: int) -> float:
  """Total quantization proxy objective for scales s on weight w (ref :175-194)."""
  rt.require_gpu()
  wd = w if not isinstance(w, np.ndarray) else _weight_on_device(w)
  return _Objective(wd, np.asarray(mu2, np.float64), block_size)(np.asarray(s, np.float64))


def _compute_channel_scales(w, mu2: np.ndarray, block_size: int = 0, num_iters: int = 3):
  """(s, gain): per-input-channel scales, or (None, 1.0) when identity is at least as good
  (ref :197-263). `w` float32 ndarray or device tensor."""
  rt.require_gpu()
  wd = w if not isinstance(w, np.ndarray) else _weight_on_device(w)
  mu2 = np.asarray(mu2, np.float64)
  in_ch = mu2.size
  mu2 = _floor_positive(mu2)
  mu = np.sqrt(mu2)

  def normalized(v):
    v = v / np.exp(np.mean(np.log(v)))
    return np.clip(v, *_SCALE_CLAMP)

  a_base = rt.to_numpy(ops.oscar_col_sumsq(wd, mean=False)) + _EPS
  objective = _Objective(wd, mu2, block_size)
  identity_loss = objective(np.ones(in_ch))
  s = normalized(np.sqrt(mu / np.sqrt(a_base)))
  best = (objective(s), s)
  for _ in range(num_iters):
    a_eff = np.maximum(objective.winner_energy(), 0.25 * a_base)   # winners of the current s
    s_cand = normalized(np.sqrt(mu / np.sqrt(a_eff)))
    s = normalized(np.sqrt(s * s_cand))
    loss = objective(s)
    if loss < best[0]:
      best = (loss, s)
  if best[0] >= identity_loss:
    return None, 1.0
  return best[1], identity_loss / max(best[0], _EPS)


def _second_moment(tensor_content: np.ndarray) -> np.ndarray:
  """np.mean(x*x, axis=0) over x = content.reshape(-1, channels) in FP64 (ref :318-324)."""
  rt.require_gpu()
  if tensor_content.dtype != np.float32:
    raise TypeError(f"OSCAR calibration expects float32 activations, got {tensor_content.dtype}")
  rec = rt.staged(tensor_content)
  xd = rec["dev"] if rec is not None else rt.to_device(tensor_content)
  return rt.to_numpy(ops.oscar_col_sumsq(xd.reshape(-1, tensor_content.shape[-1]), mean=True))


def calibrate(tfl_op, graph_info: qtyping.GraphInfo,
              tensor_content_map: MutableMapping[str, np.ndarray],
              inputs_to_ignore: Sequence[int] | None = None,
              outputs_to_ignore: Sequence[int] | None = None,
              valid_range: tuple[float, float] = (-3e38, 3e38)) -> dict[str, qtyping.QSV]:
  """min/max/num_samples + mu2 of every runtime tensor of the op (ref :266-326)."""
  lo, hi = valid_range
  out = {}
  for tid in common_quantize.get_tensor_indices_requiring_calibration(
      tfl_op, graph_info, inputs_to_ignore, outputs_to_ignore):
    res = common_quantize.collect_activation_tensor_statistics(
        tid, graph_info, tensor_content_map, valid_float_range_min=lo, valid_float_range_max=hi)
    if res is None:
      continue
    name, content, qsv = res
    qsv["mu2"] = _second_moment(content)
    out[name] = qsv
  return out


def _clip_bounds_device(wd, s: np.ndarray, col_mu2: np.ndarray, num_bits: int,
                        granularity: qtyping.QuantGranularity, ndim: int = 2, want: str = "bounds"):
  """Bounds of W*s in the shape min/max QSVs have (ref :327-383); col_mu2 unfloored masses.
  want="scale": the device tensor of symmetric scales derived from the bounds instead (float64
  values, flat) plus the parameter shape -- the bounds never visit the host."""
  n, d = wd.shape
  qmax = 2 ** (num_bits - 1) - 1
  masses = _floor_positive(col_mu2)
  blockwise = uniform_quantize_tensor.is_blockwise(granularity)
  if granularity == qtyping.QuantGranularity.TENSORWISE:
    g, shape = n * d, (1,) * ndim
    totals = np.array([float(np.tile(masses, n).sum()) + _EPS])
  elif granularity == qtyping.QuantGranularity.CHANNELWISE:
    g, shape = d, (n, 1)
    totals = np.array([float(masses.sum()) + _EPS])
  elif blockwise:
    g = uniform_quantize_tensor.extract_block_size_from_granularity(granularity)
    if _Op.FULLY_CONNECTED not in tfl_flatbuffer_utils.TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM:
      raise ValueError(f"Blockwise granularity is not supported for op: {_Op.FULLY_CONNECTED}")
    if d % g != 0:
      raise ValueError(f"Block size {g} must divide the reduction dimension {d} of"
                       f" {_Op.FULLY_CONNECTED} weights with shape {(n, d)}.")
    shape = (n, d // g)
    totals = np.array([float(masses[k * g:(k + 1) * g].sum()) + _EPS for k in range(d // g)])
  else:
    raise ValueError(f"Unsupported granularity: {granularity}")
  u = totals / (6.0 * qmax * qmax)
  noise = totals / (12.0 * qmax * qmax)
  f64 = ops._f64_dev  # pylint: disable=protected-access
  bounds, scale = ops.oscar_clip_bounds(wd, f64(s), f64(masses), g, f64(u), f64(noise), qmax,
                                        blockwise_scale=blockwise, want_bounds=want == "bounds",
                                        want_scale=want == "scale")
  if want == "scale":
    return scale, shape
  return rt.to_numpy(bounds).reshape(shape)


def get_clip_bounds(op_name, tensor_content: np.ndarray, mu2, num_bits: int,
                    granularity: qtyping.QuantGranularity) -> np.ndarray:
  """Activation-weighted optimal symmetric clip bounds of a float32 FC weight (ref :327-383)."""
  _check_fc(op_name, tensor_content)
  col = _columns_of(op_name, tensor_content, mu2)
  rt.require_gpu()
  return _clip_bounds_device(_weight_on_device(tensor_content), np.ones(col.size), col, num_bits,
                             granularity, np.ndim(tensor_content))


def _extract_mu2(tensor_qsv):
  if not tensor_qsv:
    return None
  if "mu2" in tensor_qsv:
    return tensor_qsv["mu2"]
  act = tensor_qsv.get("activation_tensor_qsv")
  return act.get("mu2") if act else None


def _compute_oscar_weight_quant_params(op_info: qtyping.OpInfo,
                                       cfg: qtyping.TensorQuantizationConfig, w: np.ndarray,
                                       mu2) -> qtyping.UniformQuantParams:
  """Scales, bounds of W*s, quantized W*s and the activation multiplier (ref :400-478)."""
  rt.require_gpu()
  _check_fc(op_info.op_name, w)
  n, in_ch = w.shape
  granularity = cfg.granularity
  blockwise = uniform_quantize_tensor.is_blockwise(granularity)
  block_size = (uniform_quantize_tensor.extract_block_size_from_granularity(granularity)
                if blockwise else 0)
  wd = _weight_on_device(w)
  s = None
  if mu2 is not None:
    mu2_arr = np.asarray(mu2, np.float64).ravel()
    if mu2_arr.size != in_ch:
      raise ValueError(f"OSCAR: activation mu2 has {mu2_arr.size} channels but"
                       f" {op_info.op_name} weights of shape {w.shape} expect {in_ch}.")
    s, _ = _compute_channel_scales(wd, mu2_arr, block_size)
  else:
    logging.warning("OSCAR: no activation second moments (mu2) found for op %s (index %d);"
                    " falling back to unscaled optimal clipping.", op_info.op_name,
                    op_info.subgraph_op_index)
  if s is None:
    s = np.ones(in_ch, dtype=np.float64)
  col = (np.ones(in_ch) if mu2 is None
         else _columns_of(op_info.op_name, w, np.asarray(mu2, np.float64).ravel() / (s * s)))
  # tensor_zp_scale_from_min_max(-bounds, bounds, ...) of a symmetric signed target happens in
  # the scan kernel's epilogue: scale = max(bound, 1e-9) / qmax (blockwise: bf16 / f16 rounded),
  # zero point 0 (ref :437-445)
  scale_d, shape = _clip_bounds_device(wd, s, col, cfg.num_bits, granularity, want="scale")
  quantized_dim = common_utils.get_weight_quantized_dim(op_info, w, granularity)
  narrow = bool(cfg.symmetric and cfg.num_bits >= 8)
  qtype = uniform_quantize_tensor.IntType(cfg.num_bits, True)
  qmin, qmax = uniform_quantize_tensor.get_quantized_range(qtype)
  q = ops.oscar_quantize(wd, ops._f64_dev(s), scale_d, n * in_ch // scale_d.numel(),  # pylint: disable=protected-access
                         int(qmin) + (1 if narrow else 0), int(qmax))
  scale = rt.to_numpy(scale_d).reshape(shape)
  if blockwise:
    scale = scale.astype(np.float32)         # exact: the values are float16-representable
  zp = uniform_quantize_tensor.assign_quantized_type(np.zeros_like(scale, dtype=np.int32), qtype)
  return qtyping.UniformQuantParams(
      scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=cfg.symmetric,
      quantized_dimension=quantized_dim, block_size=block_size,
      custom_algorithm_param={"multiplier": (1.0 / s).astype(np.float32)},
      quantized_data=rt.quantized_result(q, cfg.num_bits, w.nbytes, w.shape))


def get_tensor_quant_params(op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
                            tensor_content: np.ndarray | None = None,
                            tensor_qsv: dict[str, Any] | None = None) -> qtyping.UniformQuantParams:
  """ref :481-545. Non-weight tensors take the min/max path."""
  if tensor_content is None:
    return naive_min_max_quantize.get_tensor_quant_params(op_info, tensor_quant_config,
                                                          tensor_content, tensor_qsv)
  if not tensor_quant_config.symmetric:
    raise ValueError("OSCAR supports symmetric weight quantization only, got asymmetric config"
                     f" for op {op_info.op_name}.")
  if op_info.op_name != _Op.FULLY_CONNECTED:
    raise ValueError(f"OSCAR supports FULLY_CONNECTED only, got: {op_info.op_name}")
  return _compute_oscar_weight_quant_params(op_info, tensor_quant_config, tensor_content,
                                            _extract_mu2(tensor_qsv))


def _link(op_info, transformation, params=None) -> qtyping.OpToTensorParams:
  return qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index, parameters=params,
                                  transformations=[transformation])


def materialize_fully_connected(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
                                tensor_quant_params_cache: common_utils.TensorQuantParamsCache,
                                tensor_name_to_qsv: dict[str, Any] | None = None
                                ) -> list[qtyping.TensorTransformationParams]:
  """input: INSERT_MULTIPLY (x * 1/s), weight: QUANTIZE_TENSOR, bias / output untouched
  (ref :633-682)."""
  weight_config = op_info.op_quant_config.weight_tensor_config
  if weight_config is None:
    raise ValueError("Weight tensor quantization config is not provided for OSCAR quantization.")
  if op_info.op_name != _Op.FULLY_CONNECTED:
    raise ValueError(f"OSCAR supports FULLY_CONNECTED only, got: {op_info.op_name}")
  tensors = graph_info.subgraph_tensors
  name_of = tfl_flatbuffer_utils.get_tensor_name
  inputs = op_info.op.inputs
  input_name = name_of(tensors[inputs[0]])
  mu2 = None
  if tensor_name_to_qsv and input_name in tensor_name_to_qsv:
    mu2 = tensor_name_to_qsv[input_name].get("mu2")

  weight = tensors[inputs[1]]
  params = tensor_quant_params_cache.lookup(weight.buffer, weight_config)
  if not params:
    params = get_tensor_quant_params(
        op_info, weight_config, tfl_flatbuffer_utils.get_tensor_data(weight, graph_info.buffers),
        tensor_qsv={"mu2": mu2})
    tensor_quant_params_cache.insert(weight.buffer, weight_config, params)

  out = [
      qtyping.TensorTransformationParams(
          tensor_name=input_name, consumers=[_link(op_info, _T.INSERT_MULTIPLY, params)]),
      qtyping.TensorTransformationParams(
          tensor_name=name_of(weight), consumers=[_link(op_info, _T.QUANTIZE_TENSOR, params)]),
  ]
  if len(inputs) > 2 and inputs[2] >= 0:
    out.append(qtyping.TensorTransformationParams(
        tensor_name=name_of(tensors[inputs[2]]), consumers=[_link(op_info, _T.NO_QUANTIZE)]))
  out.append(qtyping.TensorTransformationParams(
      tensor_name=name_of(tensors[op_info.op.outputs[0]]), producer=_link(op_info, _T.NO_QUANTIZE)))
  return out

