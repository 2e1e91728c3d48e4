"""Host glue between the registry and get_tensor_quant_params.

Compact restatement of the parts of the reference's
algorithms/utils/common_utils.py that sit on the calibration / requantization
path: the quant-params cache (ref :48-77), activation-QSV injection for GPTQ
(ref :182-216), the per-tensor wrapper that calls get_tensor_quant_params
(ref :219-291), the WEIGHT_ONLY / DRQ / SRQ transformation table
(ref :1068-1121) and the quantized-dimension helpers (ref :1162-1207).
Scale-constraint propagation between ops (SAME_AS_INPUT_SCALE, ...) is graph
bookkeeping outside the hot path and is not restated.
"""
from __future__ import annotations

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


def materialize_standard_op(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
                            tensor_name_to_qsv: dict[str, Any], get_tensor_quant_params_fn,
                            tensor_quant_params_cache: TensorQuantParamsCache,
                            inputs_to_ignore: Optional[Sequence[int]] = None,
                            outputs_to_ignore: Optional[Sequence[int]] = None):
  """Per-tensor params for an op without scale constraints (ref :878-984,
  NO_CONSTRAIN branch). Order: inputs then outputs; missing (-1) tensors are
  skipped; non-float32 and ignored tensors get NO_QUANTIZE."""
  ignore_in, ignore_out = set(inputs_to_ignore or []), set(outputs_to_ignore or [])
  out = []
  for inbound, ids, ignored in ((True, op_info.op.inputs, ignore_in),
                                (False, op_info.op.outputs, ignore_out)):
    for pos, tid in enumerate(ids):
      if tid == -1:
        continue
      tensor = graph_info.subgraph_tensors[tid]
      name = tfl_flatbuffer_utils.get_tensor_name(tensor)
      if pos in ignored or int(tensor.type) != int(qtyping.TensorType.FLOAT32):
        out.append(_no_quantize_params(name, op_info, inbound))
      else:
        out.append(_tensor_params(tensor, inbound, op_info, graph_info, tensor_name_to_qsv,
                                  get_tensor_quant_params_fn, tensor_quant_params_cache))
  return out
