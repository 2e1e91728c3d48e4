"""Statistics collection and the op materializers, GPU backed.

Mirror of the reference's common_quantize.py (ref: algorithms/uniform_quantize/common_quantize.py):
the hot-path tail (:1311-1495: min/max of weights and activations, K1 / K7 in libmi355q), the
materializers of the weight-bearing ops (:251-266, :306-396, :519-636) and, from one rule table,
those of the activation-only ops (:127-304, :416-516, :639-1201).
"""
from __future__ import annotations

from collections.abc import MutableMapping, Sequence
from typing import Any, Optional

import numpy as np

from ... import default_policy
from ... import ops
from ... import qtyping
from ... import runtime as rt
from ...utils import tfl_flatbuffer_utils
from ..utils import common_utils
from . import uniform_quantize_tensor

_Op = qtyping.TFLOperationName
_ComputePrecision = qtyping.ComputePrecision


def check_if_quantized(tensor: Any) -> bool:
  return tensor.quantization is not None and tensor.quantization.scale is not None


def check_op_quantization_config(op_name, op_quant_config: qtyping.OpQuantizationConfig,
                                 config_check_policy=None) -> None:
  """ref :49-90; the policy table is default_policy.py's rule form."""
  w = op_quant_config.weight_tensor_config
  if w is None:
    raise ValueError("Weight tensor quantization is required for min/max uniform quantization.")
  if w.dtype != qtyping.TensorDataType.INT:
    raise ValueError(
        "Weights need to have integer type for min/max uniform quantization. If you wish to"
        " perform float casting quantization (e.g., fp16 weight only), please set algorithm"
        " key as 'float_casting'.")
  if op_quant_config.min_weight_elements < 0:
    raise ValueError(f"min_weight_elements must be non-negative for op: {op_name} with"
                     f" config: {op_quant_config}.")
  if op_quant_config.compute_precision in (qtyping.ComputePrecision.INTEGER,
                                           qtyping.ComputePrecision.FLOAT):
    default_policy.check_if_valid_op_config(op_name, op_quant_config, config_check_policy)
  common_utils.check_subchannel_config(op_name, op_quant_config)


# ------------------------------------------------------------------ K1 ----
def _minmax_view(tensor_data: np.ndarray, quantized_dim: Optional[int]):
  shape = tensor_data.shape
  if quantized_dim is None:
    return 1, 1, int(tensor_data.size)
  return (int(np.prod(shape[:quantized_dim], dtype=np.int64)), int(shape[quantized_dim]),
          int(np.prod(shape[quantized_dim + 1:], dtype=np.int64)))


def init_tensor_min_max(tensor_data: Optional[np.ndarray], op_info: qtyping.OpInfo) -> qtyping.QSV:
  """Per-tensor / per-channel / per-block min & max of a weight (ref :1311-1359)."""
  cfg = op_info.op_quant_config.weight_tensor_config
  if tensor_data is None or cfg is None:
    return {}
  g = cfg.granularity
  blocked_axis_to_last = None
  if g == qtyping.QuantGranularity.TENSORWISE:
    view, out_shape = (1, 1, int(tensor_data.size)), (1,) * tensor_data.ndim
  elif g == qtyping.QuantGranularity.CHANNELWISE:
    qd = common_utils.get_weight_quantized_dim(op_info, tensor_data, g)
    view = _minmax_view(tensor_data, qd)
    out_shape = tuple(d if i == qd else 1 for i, d in enumerate(tensor_data.shape)) \
        if qd is not None else (1,) * tensor_data.ndim
  elif uniform_quantize_tensor.is_blockwise(g):
    reshaped, red = uniform_quantize_tensor.reshape_data_for_blockwise(
        tensor_data, op_info.op_name, g)
    block = reshaped.shape[red]
    view = (1, int(tensor_data.size // block), int(block))
    out_shape = tuple(d for i, d in enumerate(reshaped.shape) if i != red)
    blocked_axis_to_last = red if any(d != 1 for d in reshaped.shape[red + 1:]) else None
  else:
    raise ValueError(f"Unsupported granularity: {g}")
  if tensor_data.size == 0:
    raise ValueError("zero-size array to reduction operation minimum which has no identity")
  x = uniform_quantize_tensor._as_f32_exact(tensor_data)  # pylint: disable=protected-access
  rt.require_gpu()
  xd = rt.to_device(x)
  if blocked_axis_to_last is not None:
    # blocks along an axis that is not the innermost one (ref :1336-1352 reduces the reshaped array over that axis; no op of
    # the reference's tables asks for it, a direct caller may): the kernel's blocks are contiguous runs, so the blocked axis
    # is moved last on the device -- a minimum and a maximum do not depend on where their elements sit
    xd = xd.reshape(tuple(reshaped.shape)).movedim(blocked_axis_to_last, -1).contiguous()
  mn, mx = ops.minmax(xd, *view)
  dt = tensor_data.dtype if np.issubdtype(tensor_data.dtype, np.floating) else np.float32
  return {"min": rt.to_numpy(mn).reshape(out_shape).astype(dt, copy=False),
          "max": rt.to_numpy(mx).reshape(out_shape).astype(dt, copy=False)}


# ------------------------------------------------------------------ K7 ----
def get_activation_min_max(tensor_content: np.ndarray,
                           valid_float_range_min: float | None = None,
                           valid_float_range_max: float | None = None) -> dict[str, np.ndarray]:
  """Scalar min over x > lo / max over x < hi with the all-masked fallback (ref :1362-1413)."""
  shape = (1,) * tensor_content.ndim
  if tensor_content.size == 0:
    raise ValueError("zero-size array to reduction operation minimum which has no identity")
  rec = rt.staged(tensor_content)        # reduced already with the rest of this sample?
  if rec is not None and rec["lo"] == valid_float_range_min and rec["hi"] == valid_float_range_max:
    mn, mx = rec["minmax"]
    return {"min": mn.reshape(shape), "max": mx.reshape(shape)}
  is_int = np.issubdtype(tensor_content.dtype, np.integer)
  # The kernel reads float32; other dtypes are accepted when the conversion is exact
  # (integers below 2^24, float64 holding float32 values) so min / max are unchanged.
  x = uniform_quantize_tensor._as_f32_exact(tensor_content)  # pylint: disable=protected-access
  rt.require_gpu()
  flat = rt.to_device(np.ascontiguousarray(x).reshape(-1))
  if is_int or (valid_float_range_min is None and valid_float_range_max is None):
    mm = ops.act_minmax([flat], None, None)  # integers: plain min / max (ref :1382-1384)
  else:  # a missing side gets an always-true mask
    lo = -np.inf if valid_float_range_min is None else valid_float_range_min
    hi = np.inf if valid_float_range_max is None else valid_float_range_max
    mm = ops.act_minmax([flat], lo, hi)
  host = rt.to_numpy(mm).astype(tensor_content.dtype if is_int else np.float32)
  return {"min": np.reshape(host[0, 0], shape), "max": np.reshape(host[0, 1], shape)}


def collect_activation_tensor_statistics(tensor_idx: int, graph_info: qtyping.GraphInfo,
                                         tensor_content_map: MutableMapping[str, np.ndarray],
                                         valid_float_range_min: float | None = None,
                                         valid_float_range_max: float | None = None):
  """(name, content, qsv{min,max,num_samples}) or None for constants (ref :1416-1456)."""
  tensor = graph_info.subgraph_tensors[tensor_idx]
  if graph_info.buffers[tensor.buffer].data is not None:   # a constant (get_tensor_data's test)
    return None
  name = tfl_flatbuffer_utils.get_tensor_name(tensor)
  content = tensor_content_map[name]
  qsv = get_activation_min_max(content, valid_float_range_min, valid_float_range_max)
  qsv["num_samples"] = np.array(content.shape[0] if content.ndim > 0 else 1)
  return name, content, qsv


def get_tensor_indices_requiring_calibration(tfl_op, graph_info: qtyping.GraphInfo,
                                             inputs_to_ignore: Sequence[int] | None = None,
                                             outputs_to_ignore: Sequence[int] | None = None):
  """ref :1459-1495."""
  skip_in = set(inputs_to_ignore or [])
  skip_in.update(k for k, tid in enumerate(tfl_op.inputs)
                 if tid != -1 and check_if_quantized(graph_info.subgraph_tensors[tid]))
  skip_out = set(outputs_to_ignore or [])
  return ([tid for k, tid in enumerate(tfl_op.inputs) if k not in skip_in and tid != -1]
          + [tid for k, tid in enumerate(tfl_op.outputs) if k not in skip_out and tid != -1])


# -------------------------------------------------------- materializers ----
def _are_weights_too_small(op_info, graph_info, weight_index: int) -> bool:
  tensor = graph_info.subgraph_tensors[op_info.op.inputs[weight_index]]
  data = tfl_flatbuffer_utils.get_tensor_data(tensor, graph_info.buffers)
  return data is not None and np.size(data) < op_info.op_quant_config.min_weight_elements


def _is_srq(op_info: qtyping.OpInfo) -> bool:
  c = op_info.op_quant_config
  return c.compute_precision == _ComputePrecision.INTEGER and c.activation_tensor_config is not None


def _materialize_bias_for_fc_conv_ops(op_info, graph_info, op_tensor_params, op_input_index=0,
                                      op_weight_index=1, op_bias_index=2) -> None:
  """Fused bias: int32 with scale = s_in * s_w under SRQ, untouched otherwise (ref :306-396)."""
  _, weight_tensor, bias_tensor, _ = tfl_flatbuffer_utils.parse_fc_bmm_conv_tensors(
      op_info.op, graph_info.subgraph_tensors, op_input_index, op_weight_index, op_bias_index)
  if bias_tensor is None or check_if_quantized(bias_tensor):
    return
  # position of the bias entry in op_tensor_params (-1 inputs are skipped there)
  pos = sum(1 for t in op_info.op.inputs[:op_bias_index] if t != -1)
  in_pos = sum(1 for t in op_info.op.inputs[:op_input_index] if t != -1)
  w_pos = sum(1 for t in op_info.op.inputs[:op_weight_index] if t != -1)
  bias_params = None
  if _is_srq(op_info):
    bias = tfl_flatbuffer_utils.get_tensor_data(bias_tensor, graph_info.buffers)
    in_params = op_tensor_params[in_pos].consumers[0].parameters
    w_params = op_tensor_params[w_pos].consumers[0].parameters
    if w_params is None and check_if_quantized(weight_tensor):
      wq = weight_tensor.quantization
      if op_info.op_quant_config.weight_tensor_config is None:
        raise ValueError("weight_tensor_config cannot be None when weight tensor is quantized.")
      w_params = qtyping.UniformQuantParams(
          num_bits=op_info.op_quant_config.weight_tensor_config.num_bits, scale=wq.scale,
          zero_point=wq.zeroPoint, quantized_dimension=wq.quantizedDimension)
    try:
      bias_params = uniform_quantize_tensor.symmetric_quantize_bias_tensor(
          bias, in_params, w_params)
    except ValueError as e:
      raise ValueError(f"Failed to quantize bias tensor for op {op_info.op_name} with op id"
                       f" {op_info.subgraph_op_index}.") from e
  op_tensor_params[pos] = common_utils.get_tensor_transformation_params(
      tfl_flatbuffer_utils.get_tensor_name(bias_tensor), op_info, is_inbounding_tensor=True,
      quant_params=bias_params, is_constant=_is_srq(op_info))


def materialize_fc_conv(get_tensor_quant_params_fn, op_info: qtyping.OpInfo,
                        graph_info: qtyping.GraphInfo, tensor_name_to_qsv: dict[str, Any],
                        tensor_quant_params_cache: common_utils.TensorQuantParamsCache,
                        input_index: int = 0, weight_index: int = 1, bias_index: int = 2):
  """FULLY_CONNECTED / CONV_2D / DEPTHWISE_CONV_2D (ref :519-576)."""
  ignored = [bias_index]
  w_tensor = graph_info.subgraph_tensors[op_info.op.inputs[weight_index]]
  if check_if_quantized(w_tensor) or _are_weights_too_small(op_info, graph_info, weight_index):
    ignored.append(weight_index)
  params = common_utils.materialize_standard_op(
      op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
      tensor_quant_params_cache=tensor_quant_params_cache, inputs_to_ignore=ignored)
  _materialize_bias_for_fc_conv_ops(op_info, graph_info, params, input_index, weight_index,
                                    bias_index)
  return params


def materialize_batch_matmul(get_tensor_quant_params_fn, op_info, graph_info, tensor_name_to_qsv,
                             tensor_quant_params_cache):
  """BATCH_MATMUL: a standard op; a constant rhs is the weight (ref :234-248)."""
  return common_utils.materialize_standard_op(
      op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
      tensor_quant_params_cache=tensor_quant_params_cache)


def materialize_conv2d_transpose(get_tensor_quant_params_fn, op_info, graph_info,
                                 tensor_name_to_qsv, tensor_quant_params_cache):
  """TRANSPOSE_CONV: inputs are (output_shape, weight, input, bias) (ref :579-636)."""
  shape_index, weight_index, input_index, bias_index = 0, 1, 2, 3
  ignored = [shape_index, bias_index]
  if _are_weights_too_small(op_info, graph_info, weight_index):
    ignored.append(weight_index)
  params = common_utils.materialize_standard_op(
      op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
      tensor_quant_params_cache=tensor_quant_params_cache, inputs_to_ignore=ignored)
  if len(params) < 2:
    raise ValueError("Materialize standard op should return at least two tensors for"
                     " conv2d_transpose.")
  _materialize_bias_for_fc_conv_ops(op_info, graph_info, params, op_input_index=input_index,
                                    op_weight_index=weight_index, op_bias_index=bias_index)
  return params


def materialize_embedding_lookup(get_tensor_quant_params_fn, op_info, graph_info,
                                 tensor_name_to_qsv, tensor_quant_params_cache):
  """EMBEDDING_LOOKUP: the index input is never quantized (ref :251-266)."""
  return common_utils.materialize_standard_op(
      op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
      tensor_quant_params_cache=tensor_quant_params_cache, inputs_to_ignore=[0])


def materialize_input(get_tensor_quant_params_fn, op_info, graph_info, tensor_name_to_qsv,
                      tensor_quant_params_cache):
  return common_utils.materialize_standard_op(
      op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
      tensor_quant_params_cache=tensor_quant_params_cache)


materialize_output = materialize_input


# ---- the ops that only carry activations (ref :127-304, 416-516, 639-1201) ---------------------
# One row per op: (scale constraint, operand positions that are never quantized -- shapes, axes,
# indices, conditions). The fixed-output ops list the parameters their TFLite kernels hard-code.
_C = common_utils.OpQuantConstraint
_ACTIVATION_OP_RULES = {
    "composite": (_C.NO_CONSTRAIN, ()), "add": (_C.NO_CONSTRAIN, ()), "sub": (_C.NO_CONSTRAIN, ()),
    "mul": (_C.NO_CONSTRAIN, ()), "div": (_C.NO_CONSTRAIN, ()), "gelu": (_C.NO_CONSTRAIN, ()),
    "rsqrt": (_C.NO_CONSTRAIN, ()), "sqrt": (_C.NO_CONSTRAIN, ()), "hard_swish": (_C.NO_CONSTRAIN, ()),
    "relu": (_C.NO_CONSTRAIN, ()), "equal": (_C.NO_CONSTRAIN, ()), "not_equal": (_C.NO_CONSTRAIN, ()),
    "squared_difference": (_C.NO_CONSTRAIN, ()), "sum": (_C.NO_CONSTRAIN, (1,)),
    "mean": (_C.NO_CONSTRAIN, (1,)),
    "reshape": (_C.SAME_AS_INPUT_SCALE, (1,)), "transpose": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "average_pool_2d": (_C.SAME_AS_INPUT_SCALE, ()), "max_pool_2d": (_C.SAME_AS_INPUT_SCALE, ()),
    "space_to_depth": (_C.SAME_AS_INPUT_SCALE, ()), "unpack": (_C.SAME_AS_INPUT_SCALE, ()),
    "slice": (_C.SAME_AS_INPUT_SCALE, (1, 2)), "strided_slice": (_C.SAME_AS_INPUT_SCALE, (1, 2, 3)),
    "split": (_C.SAME_AS_INPUT_SCALE, (0,)), "pad": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "mirror_pad": (_C.SAME_AS_INPUT_SCALE, (1,)), "resize_bilinear": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "resize_nearest_neighbor": (_C.SAME_AS_INPUT_SCALE, (1,)), "gather_nd": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "gather": (_C.SAME_AS_INPUT_SCALE, (1,)), "broadcast_to": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "reduce_min": (_C.SAME_AS_INPUT_SCALE, (1,)),
    "concatenation": (_C.SAME_AS_OUTPUT_SCALE, ()), "maximum": (_C.SAME_AS_OUTPUT_SCALE, ()),
    "pack": (_C.SAME_AS_OUTPUT_SCALE, ()), "select": (_C.SAME_AS_OUTPUT_SCALE, (0,)),
    "select_v2": (_C.SAME_AS_OUTPUT_SCALE, (0,)), "padv2": (_C.SAME_AS_OUTPUT_SCALE, (1,)),
    "dynamic_update_slice": (_C.SAME_AS_OUTPUT_SCALE, (2,)),
}


def _activation_op_materializer(name: str, constraint, ignored: tuple):
  def materialize(get_tensor_quant_params_fn, op_info, graph_info, tensor_name_to_qsv,
                  tensor_quant_params_cache):
    return common_utils.materialize_standard_op(
        op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
        tensor_quant_params_cache, constraint=constraint, inputs_to_ignore=list(ignored) or None)
  materialize.__name__ = materialize.__qualname__ = f"materialize_{name}"
  materialize.__doc__ = (f"tfl.{name}: {constraint.name}"
                         + (f", operands {list(ignored)} left alone" if ignored else "") + ".")
  return materialize


for _name, (_constraint, _ignored) in _ACTIVATION_OP_RULES.items():
  globals()[f"materialize_{_name}"] = _activation_op_materializer(_name, _constraint, _ignored)


def _fixed(num_bits: int, scale: float, zero_point: int, symmetric: bool) -> qtyping.UniformQuantParams:
  return qtyping.UniformQuantParams(num_bits=num_bits, quantized_dimension=None, scale=np.array(scale),
                                    zero_point=np.array(zero_point), symmetric=symmetric)


def materialize_softmax_and_logistic(get_tensor_quant_params_fn, op_info, graph_info,
                                     tensor_name_to_qsv, tensor_quant_params_cache):
  """Output in [0, 1): 1/256 with zero point -128 at 8 bits, 1/32768 at 16 (ref :195-231)."""
  fixed = {8: _fixed(8, 1.0 / 256, -128, False), 16: _fixed(16, 1.0 / 32768, 0, True)}
  return common_utils.materialize_op_with_output_activation_constraint(
      op_info, graph_info, tensor_name_to_qsv, fixed, get_tensor_quant_params_fn,
      tensor_quant_params_cache)


def materialize_tanh(get_tensor_quant_params_fn, op_info, graph_info, tensor_name_to_qsv,
                     tensor_quant_params_cache):
  """Output in [-1, 1): 2^-(bits-1), zero point 0 (ref :639-666)."""
  fixed = {bits: _fixed(bits, 1.0 / (1 << (bits - 1)), 0, bits == 16) for bits in (8, 16)}
  return common_utils.materialize_op_with_output_activation_constraint(
      op_info, graph_info, tensor_name_to_qsv, fixed, get_tensor_quant_params_fn,
      tensor_quant_params_cache)
