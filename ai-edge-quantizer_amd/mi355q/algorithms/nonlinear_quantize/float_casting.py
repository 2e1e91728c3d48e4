"""float_casting: FP32 weights stored as FP16 behind a DEQUANTIZE op, GPU backed.

Mirror of ref: algorithms/nonlinear_quantize/float_casting.py. Weight-only: the constant of a
FULLY_CONNECTED / CONV_2D / DEPTHWISE_CONV_2D / CONV_2D_TRANSPOSE / EMBEDDING_LOOKUP op becomes
`NonLinearQuantParams(num_bits=16, quantized_data=float16 weights)` with ADD_DEQUANTIZE; every
other tensor of the op is left alone. The cast (round to nearest even, as `astype(np.float16)`)
is one HBM-bound pass on the GPU (mi355q_cast_f32_to_f16).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from ...utils import tfl_flatbuffer_utils
from ..utils import common_utils

ALGORITHM_KEY = "float_casting"
_Op = qtyping.TFLOperationName
_T = qtyping.QuantTransformation

_FP16_QUANT_CONFIG = qtyping.TensorQuantizationConfig(num_bits=16, dtype=qtyping.TensorDataType.FLOAT)
SUPPORTED_WEIGHT_QUANT_OPS = frozenset([_Op.FULLY_CONNECTED, _Op.CONV_2D, _Op.DEPTHWISE_CONV_2D,
                                        _Op.CONV_2D_TRANSPOSE, _Op.EMBEDDING_LOOKUP])


def check_op_quantization_config(op_name, op_quant_config: qtyping.OpQuantizationConfig,
                                 config_check_policy: Optional[qtyping.ConfigCheckPolicyDict] = None) -> None:
  """ref :37-95."""
  if config_check_policy is not None and config_check_policy:
    raise ValueError(f"Config check isn't implemented yet for op: {op_name}.")
  if op_quant_config.compute_precision != qtyping.ComputePrecision.FLOAT:
    raise ValueError("Currently, only Weight-Only is supported for float casting quantization. Got"
                     f" unsupported execution mode: {op_quant_config.compute_precision} for op:"
                     f" {op_name}")
  if op_quant_config.activation_tensor_config is not None:
    raise ValueError("Activation tensor quantization is not supported for float casting"
                     " quantization.")
  if op_name not in SUPPORTED_WEIGHT_QUANT_OPS:
    raise ValueError(f"Unsupported op: {op_name} for float casting quantization.")
  w = op_quant_config.weight_tensor_config
  if w is None:
    raise ValueError("Weight tensor quantization config is required for float casting quantization.")
  if w.num_bits != 16 or w.dtype != qtyping.TensorDataType.FLOAT:
    raise ValueError("Currently, float casting quantization config requires number of bits to be"
                     f" set as 16, dtype as float, got {w.num_bits} and {w.dtype} .")


def _to_float16(weight: np.ndarray) -> np.ndarray:
  rt.require_gpu()
  if weight.dtype != np.float32:
    raise TypeError(f"float casting expects float32 weights, got {weight.dtype}")
  return rt.to_numpy(ops.cast_f16(rt.to_device(weight)))


def _left_alone(op_info, tensor, is_inbounding_tensor: bool) -> qtyping.TensorTransformationParams:
  link = qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index, transformations=[_T.NO_QUANTIZE])
  name = tfl_flatbuffer_utils.get_tensor_name(tensor)
  if is_inbounding_tensor:
    return qtyping.TensorTransformationParams(tensor_name=name, consumers=[link])
  return qtyping.TensorTransformationParams(tensor_name=name, producer=link)


def _materialize(op_info, graph_info, cache, share_cached: bool, **indices):
  """Result order as in the reference: input, weight, output, bias."""
  inp, weight, bias, out = tfl_flatbuffer_utils.parse_fc_bmm_conv_tensors(
      op_info.op, graph_info.subgraph_tensors, **indices)
  content = tfl_flatbuffer_utils.get_tensor_data(weight, graph_info.buffers)
  if content is None:
    link = qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index, transformations=[_T.NO_QUANTIZE])
  else:
    params = cache.lookup(weight.buffer, _FP16_QUANT_CONFIG)
    if not params:
      params = qtyping.NonLinearQuantParams(num_bits=16, quantized_data=_to_float16(content))
      cache.insert(weight.buffer, _FP16_QUANT_CONFIG, params)
    elif not share_cached:
      # ref :167-175: the FC / conv form hands every op its own parameter object (equal data)
      params = qtyping.NonLinearQuantParams(num_bits=16, quantized_data=params.quantized_data)
    link = qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index, parameters=params,
                                    transformations=[_T.ADD_DEQUANTIZE])
  res = [_left_alone(op_info, inp, True),
         qtyping.TensorTransformationParams(tensor_name=tfl_flatbuffer_utils.get_tensor_name(weight),
                                            consumers=[link]),
         _left_alone(op_info, out, False)]
  if bias is not None:
    res.append(_left_alone(op_info, bias, True))
  return res


def materialize_fc_conv(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
                        tensor_name_to_qsv: dict[str, Any],
                        tensor_quant_params_cache: common_utils.TensorQuantParamsCache):
  """FULLY_CONNECTED / CONV_2D / DEPTHWISE_CONV_2D (ref :98-196)."""
  del tensor_name_to_qsv
  return _materialize(op_info, graph_info, tensor_quant_params_cache, share_cached=False)


def materialize_embedding_lookup(op_info, graph_info, tensor_name_to_qsv, tensor_quant_params_cache):
  """ref :199-223."""
  return materialize_fc_conv(op_info, graph_info, tensor_name_to_qsv, tensor_quant_params_cache)


def materialize_conv2d_transpose(op_info, graph_info, tensor_name_to_qsv, tensor_quant_params_cache):
  """Operands: output shape, weight, input, bias (ref :226-310)."""
  del tensor_name_to_qsv
  return _materialize(op_info, graph_info, tensor_quant_params_cache, share_cached=True,
                      input_index=2, weight_index=1, bias_index=3, output_index=0)


def init_qsvs(*_) -> qtyping.QSV:
  return {}


def calibrate(*_) -> dict[str, qtyping.QSV]:
  return {}
