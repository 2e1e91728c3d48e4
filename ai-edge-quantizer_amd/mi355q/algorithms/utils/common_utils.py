"""Host glue between the registry and get_tensor_quant_params.

Compact restatement of the parts of the reference's
algorithms/utils/common_utils.py that sit on the calibration / requantization
path: the quant-params cache (ref :48-77), activation-QSV injection for GPTQ
(ref :182-216), the per-tensor wrapper that calls get_tensor_quant_params
(ref :219-291), the WEIGHT_ONLY / DRQ / SRQ transformation table
(ref :1068-1121), the quantized-dimension helpers (ref :1162-1207) and the
scale constraints that tie an op's tensors together (same-as-input, same-as-output,
fixed output; ref :167-179, 381-590, 878-1065).
"""
from __future__ import annotations

import dataclasses
import enum
from typing import Any, Optional, Sequence

import numpy as np

from ... import qtyping
from ...utils import tfl_flatbuffer_utils

_Op = qtyping.TFLOperationName
_T = qtyping.QuantTransformation

_DRQ_OR_WEIGHT_ONLY_OPS = frozenset([
    _Op.FULLY_CONNECTED, _Op.CONV_2D, _Op.BATCH_MATMUL, _Op.EMBEDDING_LOOKUP,
    _Op.DEPTHWISE_CONV_2D, _Op.CONV_2D_TRANSPOSE])
_SUPPORTED_SUBCHANNEL_OPS = frozenset([_Op.FULLY_CONNECTED, _Op.EMBEDDING_LOOKUP])


class TensorQuantParamsCache:
  """Computed quant params keyed by (buffer id, TensorQuantizationConfig)."""

  def __init__(self):
    self._cache: dict[tuple[Any, qtyping.TensorQuantizationConfig], Any] = {}

  def lookup(self, buffer_id, quant_config):
    return self._cache.get((buffer_id, quant_config))

  def insert(self, buffer_id, quant_config, quant_params):
    self._cache[(buffer_id, quant_config)] = quant_params
    return quant_params


def _is_blockwise(granularity) -> bool:
  return "BLOCKWISE" in str(granularity)


def check_subchannel_config(op_name, op_quant_config: qtyping.OpQuantizationConfig) -> None:
  """ref :80-101."""
  w = op_quant_config.weight_tensor_config
  if w is None or not _is_blockwise(w.granularity):
    return
  if op_name not in _SUPPORTED_SUBCHANNEL_OPS:
    raise ValueError(f"Unsupported op for blockwise quantization: {op_name}.")
  if op_quant_config.activation_tensor_config is not None:
    raise ValueError("Blockwise quantization does not support activation tensor quantization.")
  if not w.symmetric:
    raise ValueError("Blockwise quantization does not support for asymmetric weight"
                     " quantization.")


def get_bmm_weight_quantized_dim(weight_tensor_data: np.ndarray, adj_y: bool) -> int:
  rank = len(weight_tensor_data.shape)
  return rank - 2 if adj_y else rank - 1


def get_weight_quantized_dim(op_info: qtyping.OpInfo, tensor_data: np.ndarray,
                             granularity: qtyping.QuantGranularity):
  """ref :1162-1193."""
  if granularity == qtyping.QuantGranularity.CHANNELWISE:
    if op_info.op_name == _Op.BATCH_MATMUL:
      return get_bmm_weight_quantized_dim(tensor_data, adj_y=op_info.op.builtinOptions.adjY)
    return tfl_flatbuffer_utils.TFL_OP_TO_WEIGHT_QUANTIZED_DIM.get(op_info.op_name, None)
  if _is_blockwise(granularity):
    return tfl_flatbuffer_utils.TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM[op_info.op_name]
  return None


def get_reduce_dims(quantized_dim: Optional[int], tensor_shape: Sequence[int]):
  """ref :1196-1207."""
  if quantized_dim is None:
    return None
  return tuple(d for d in range(len(tensor_shape)) if d != quantized_dim)


def get_tensor_transformations(op_quant_config: qtyping.OpQuantizationConfig,
                               is_inbounding_tensor: bool, is_constant: bool):
  """SRQ / DRQ / WEIGHT_ONLY transformation table (ref :1068-1121)."""
  integer = op_quant_config.compute_precision == qtyping.ComputePrecision.INTEGER
  if integer and op_quant_config.activation_tensor_config is not None:   # SRQ
    if not is_inbounding_tensor:
      return [_T.ADD_DEQUANTIZE]
    return [_T.QUANTIZE_TENSOR] if is_constant else [_T.ADD_QUANTIZE]
  if integer:                                                            # DRQ
    return [_T.QUANTIZE_TENSOR] if (is_inbounding_tensor and is_constant) else [_T.NO_QUANTIZE]
  if (op_quant_config.compute_precision == qtyping.ComputePrecision.FLOAT
      and op_quant_config.explicit_dequantize):                          # WEIGHT_ONLY
    return [_T.ADD_DEQUANTIZE] if (is_inbounding_tensor and is_constant) else [_T.NO_QUANTIZE]
  raise ValueError("Unsupported compute precision: %s" % op_quant_config.compute_precision)


def get_tensor_transformation_params(tensor_name: str, op_info: qtyping.OpInfo,
                                     is_inbounding_tensor: bool, quant_params=None,
                                     is_constant: bool = False):
  """ref :1124-1159."""
  link = qtyping.OpToTensorParams(
      subgraph_op_id=op_info.subgraph_op_index, parameters=quant_params,
      transformations=get_tensor_transformations(op_info.op_quant_config,
                                                 is_inbounding_tensor, is_constant))
  if is_inbounding_tensor:
    return qtyping.TensorTransformationParams(tensor_name=tensor_name, consumers=[link])
  return qtyping.TensorTransformationParams(tensor_name=tensor_name, producer=link)


def _no_quantize_params(tensor_name: str, op_info: qtyping.OpInfo, is_inbounding_tensor: bool):
  link = qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index,
                                  transformations=[_T.NO_QUANTIZE])
  if is_inbounding_tensor:
    return qtyping.TensorTransformationParams(tensor_name=tensor_name, consumers=[link])
  return qtyping.TensorTransformationParams(tensor_name=tensor_name, producer=link)


def _get_tensor_qsv_val(tensor_name, op_info, graph_info, tensor_name_to_qsv):
  """QSV of the tensor + the op's first input's QSV nested under
  "activation_tensor_qsv" (what GPTQ reads its Hessian from; ref :182-216)."""
  val = tensor_name_to_qsv.get(tensor_name)
  if op_info.op and op_info.op.inputs:
    act_name = tfl_flatbuffer_utils.get_tensor_name(
        graph_info.subgraph_tensors[op_info.op.inputs[0]])
    act = tensor_name_to_qsv.get(act_name)
    if act is not None:
      val = dict(val) if val is not None else {}
      val["activation_tensor_qsv"] = act
  return val


def _tensor_params(tensor, is_inbounding_tensor, op_info, graph_info, tensor_name_to_qsv,
                   get_tensor_quant_params_fn, cache: TensorQuantParamsCache, quant_params=None):
  """ref :219-291 -- THE call site of get_tensor_quant_params (the drop-in point)."""
  name = tfl_flatbuffer_utils.get_tensor_name(tensor)
  data = tfl_flatbuffer_utils.get_tensor_data(tensor, graph_info.buffers)
  config = op_info.op_quant_config.activation_tensor_config
  is_constant = data is not None
  if is_constant and op_info.op_name in _DRQ_OR_WEIGHT_ONLY_OPS:
    config = op_info.op_quant_config.weight_tensor_config
  if quant_params is None and config is not None:
    cached = cache.lookup(tensor.buffer, config) if is_constant else None
    if cached:
      quant_params = cached
    else:
      try:
        quant_params = get_tensor_quant_params_fn(
            op_info, config, data,
            _get_tensor_qsv_val(name, op_info, graph_info, tensor_name_to_qsv))
      except Exception as e:  # noqa: BLE001 - same wrapping as the reference
        raise ValueError(
            f"Failed to get quantization parameters for tensor: {name}. Error: {e}") from e
      if is_constant:
        cache.insert(tensor.buffer, config, quant_params)
  return get_tensor_transformation_params(name, op_info, is_inbounding_tensor, quant_params,
                                          is_constant)


class OpQuantConstraint(enum.Enum):
  """How an op ties the scales of its tensors (ref :167-179)."""
  NO_CONSTRAIN = 0
  SAME_AS_INPUT_SCALE = 1    # transpose / reshape / split ...: every tensor uses the input's scale
  SAME_AS_OUTPUT_SCALE = 2   # concatenation ...: every tensor uses the output's scale
  FIXED_OUTPUT_SCALE = 3     # softmax / logistic / tanh: the kernel dictates the output scale


def _get_min_max_from_quant_params(quant_params: qtyping.UniformQuantParams):
  """The float range a set of (single-scale) quantization parameters can represent: dequantized
  qmin / qmax, mirrored when symmetric (ref :858-875). O(1) host math with the dtype flow of
  `uniform_dequantize(np.array(q), params)`: float64 scalars."""
  from ..uniform_quantize import uniform_quantize_tensor  # (circular at import time)
  qmin, qmax = uniform_quantize_tensor.get_quantized_range(
      uniform_quantize_tensor.IntType(quant_params.num_bits, True))
  scale, zp = quant_params.scale, quant_params.zero_point
  if np.ndim(scale) != 0:
    if np.size(scale) != 1 or np.size(zp) != 1:
      raise ValueError("Scale and zero_point must contain single element for scalar tensor."
                       f" Got scale: {scale}, zero_point: {zp}")
    scale, zp = np.array(np.asarray(scale).item()), np.array(np.asarray(zp).item())
  lo = np.multiply(np.array(qmin) - zp, scale)
  hi = np.multiply(np.array(qmax) - zp, scale)
  if quant_params.symmetric:
    lo = -hi
  return lo, hi


def _with_given_params(tensors, quant_params, is_inbounding_tensor, op_info, graph_info,
                       tensor_name_to_qsv, get_tensor_quant_params_fn, cache):
  """Every tensor takes `quant_params`; constants are quantized with them (ref :381-437)."""
  from ..uniform_quantize import uniform_quantize_tensor
  if quant_params is not None and quant_params.quantized_data is not None:
    quant_params = dataclasses.replace(quant_params, quantized_data=None)
  out = []
  for tensor in tensors:
    data = tfl_flatbuffer_utils.get_tensor_data(tensor, graph_info.buffers)
    params = quant_params
    if quant_params is not None and data is not None:
      params = dataclasses.replace(
          quant_params, quantized_data=uniform_quantize_tensor.uniform_quantize(data, quant_params))
    out.append(_tensor_params(tensor, is_inbounding_tensor, op_info, graph_info, tensor_name_to_qsv,
                              get_tensor_quant_params_fn, cache, quant_params=params))
  return out


def _same_as_input_scale(inputs, outputs, op_info, graph_info, tensor_name_to_qsv, fn, cache):
  """ref :440-524. The outputs inherit the (single) input's parameters AND its QSV, so that
  ops further down see the range that will actually be used."""
  if len(inputs) != 1:
    raise ValueError("Trying to get a single tensor params with a list of multiple tensor with"
                     f" size {len(inputs)}.")
  first = _tensor_params(inputs[0], True, op_info, graph_info, tensor_name_to_qsv, fn, cache)
  params = first.consumers[0].parameters
  if not isinstance(params, qtyping.UniformQuantParams):
    raise ValueError("_materialize_standard_op_with_same_as_input_scale only supports"
                     f" UniformQuantParams. For tensor {first.tensor_name}, got {type(params)}")
  out = [first] + _with_given_params(outputs, params, False, op_info, graph_info,
                                     tensor_name_to_qsv, fn, cache)
  qsv = tensor_name_to_qsv.get(first.tensor_name)
  if qsv is None:
    if tfl_flatbuffer_utils.get_tensor_data(inputs[0], graph_info.buffers) is None:
      raise ValueError(f"Input tensor qsv is None for tensor {first.tensor_name}.")
    lo, hi = _get_min_max_from_quant_params(params)      # a constant input: range of its params
    qsv = {"min": lo, "max": hi}
  for tensor in outputs:
    tensor_name_to_qsv[tfl_flatbuffer_utils.get_tensor_name(tensor)] = qsv
  return out


def _same_as_output_scale(inputs, outputs, op_info, graph_info, tensor_name_to_qsv, fn, cache):
  """ref :527-590."""
  if len(outputs) != 1:
    raise ValueError("Trying to get a single tensor params with a list of multiple tensor with"
                     f" size {len(outputs)}.")
  last = _tensor_params(outputs[0], False, op_info, graph_info, tensor_name_to_qsv, fn, cache)
  params = None
  if last.producer is not None:
    params = last.producer.parameters
    if not isinstance(params, qtyping.UniformQuantParams):
      raise ValueError("_materialize_standard_op_with_same_as_output_scale only supports"
                       f" UniformQuantParams. For tensor {last.tensor_name}, got {type(params)}")
  return _with_given_params(inputs, params, True, op_info, graph_info, tensor_name_to_qsv, fn,
                            cache) + [last]


def materialize_standard_op(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
                            tensor_name_to_qsv: dict[str, Any], get_tensor_quant_params_fn,
                            tensor_quant_params_cache: TensorQuantParamsCache,
                            constraint: OpQuantConstraint = OpQuantConstraint.NO_CONSTRAIN,
                            inputs_to_ignore: Optional[Sequence[int]] = None,
                            outputs_to_ignore: Optional[Sequence[int]] = None):
  """Per-tensor params of an op (ref :878-984). Result order: inputs then outputs, missing (-1)
  tensors skipped; ignored positions and non-float32 tensors get NO_QUANTIZE; the others follow
  the op's scale constraint."""
  tensors = graph_info.subgraph_tensors
  is_f32 = lambda tid: int(tensors[tid].type) == int(qtyping.TensorType.FLOAT32)  # noqa: E731
  chosen, skipped = {}, {}
  for inbound, ids, ignore in ((True, op_info.op.inputs, set(inputs_to_ignore or [])),
                               (False, op_info.op.outputs, set(outputs_to_ignore or []))):
    # (the reference looks the dtype of index -1 up as well: Python's last tensor)
    keep = [k for k, tid in enumerate(ids) if k not in ignore and is_f32(tid)]
    chosen[inbound] = [tensors[tid] for k, tid in enumerate(ids) if tid != -1 and k in keep]
    skipped[inbound] = [k for k, tid in enumerate(ids) if tid != -1 and k not in keep]
  args = (op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn, tensor_quant_params_cache)
  if not chosen[True] and not chosen[False]:
    params = []
  elif constraint == OpQuantConstraint.SAME_AS_INPUT_SCALE:
    params = _same_as_input_scale(chosen[True], chosen[False], *args)
  elif constraint == OpQuantConstraint.SAME_AS_OUTPUT_SCALE:
    params = _same_as_output_scale(chosen[True], chosen[False], *args)
  else:
    params = ([_tensor_params(t, True, *args) for t in chosen[True]]
              + [_tensor_params(t, False, *args) for t in chosen[False]])
  # weave the NO_QUANTIZE records of the skipped tensors back in, in operand order
  params = iter(params)
  out = []
  for inbound, ids in ((True, op_info.op.inputs), (False, op_info.op.outputs)):
    for k, tid in enumerate(ids):
      if tid == -1:
        continue
      if k in skipped[inbound]:
        out.append(_no_quantize_params(tfl_flatbuffer_utils.get_tensor_name(tensors[tid]), op_info, inbound))
      else:
        out.append(next(params))
  return out


def materialize_op_with_output_activation_constraint(
    op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo, tensor_name_to_qsv: dict[str, Any],
    output_activation_constraints: dict[int, qtyping.UniformQuantParams], get_tensor_quant_params_fn,
    tensor_quant_params_cache: TensorQuantParamsCache):
  """Ops whose kernels hard-code the output scale (ref :987-1065): under SRQ the output takes
  the fixed parameters for the activation bit width, and its QSV becomes their range."""
  if len(op_info.op.outputs) != 1:
    raise ValueError("Materialize op with output activation constraint only supports ops with a"
                     " single output tensor.")
  params = materialize_standard_op(op_info, graph_info, tensor_name_to_qsv, get_tensor_quant_params_fn,
                                   tensor_quant_params_cache, constraint=OpQuantConstraint.FIXED_OUTPUT_SCALE)
  last = params[-1]
  act = op_info.op_quant_config.activation_tensor_config
  if act is not None and last.producer is not None:
    if act.num_bits not in output_activation_constraints:
      raise ValueError("Output activation constraints dictionary does not contain entity for"
                       f" activation num bits {act.num_bits}.")
    fixed = output_activation_constraints[act.num_bits]
    last.producer = qtyping.OpToTensorParams(subgraph_op_id=last.producer.subgraph_op_id,
                                             transformations=last.producer.transformations,
                                             parameters=fixed)
    lo, hi = _get_min_max_from_quant_params(fixed)
    tensor_name_to_qsv[last.tensor_name]["min"] = lo
    tensor_name_to_qsv[last.tensor_name]["max"] = hi
  return params
