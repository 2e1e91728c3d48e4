"""Tensor-buffer access helpers and the op -> quantized-dimension tables.

ref: utils/tfl_flatbuffer_utils.py:34-112 (tables), :236-263 (get_tensor_name /
get_tensor_data). Model file I/O lives elsewhere (out of the hot path).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from .. import qtyping
from .. import schema

_Op = qtyping.TFLOperationName

# Per-channel quantized dimension (TFLite quantization spec); ref :95-101.
TFL_OP_TO_WEIGHT_QUANTIZED_DIM = qtyping.FrozenMapping({
    _Op.FULLY_CONNECTED: 0,
    _Op.DEPTHWISE_CONV_2D: 3,
    _Op.CONV_2D: 0,
    _Op.EMBEDDING_LOOKUP: 0,
    _Op.CONV_2D_TRANSPOSE: 0,
})

# Blockwise quantized dimension (AEQ-internal convention); ref :103-106.
TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM = qtyping.FrozenMapping({
    _Op.FULLY_CONNECTED: 1,
    _Op.EMBEDDING_LOOKUP: 1,
})

TFL_OP_NAME_TO_CODE = qtyping.FrozenMapping({
    _Op.FULLY_CONNECTED: schema.BuiltinOperator.FULLY_CONNECTED,
    _Op.BATCH_MATMUL: schema.BuiltinOperator.BATCH_MATMUL,
    _Op.CONV_2D: schema.BuiltinOperator.CONV_2D,
    _Op.DEPTHWISE_CONV_2D: schema.BuiltinOperator.DEPTHWISE_CONV_2D,
    _Op.CONV_2D_TRANSPOSE: schema.BuiltinOperator.TRANSPOSE_CONV,
    _Op.EMBEDDING_LOOKUP: schema.BuiltinOperator.EMBEDDING_LOOKUP,
})
TFL_OP_CODE_TO_NAME = qtyping.FrozenMapping({v: k for k, v in TFL_OP_NAME_TO_CODE.items()})


def get_tensor_name(tensor: Any) -> str:
  return schema.tensor_name(tensor)


def get_tensor_data(tensor: Any, buffers: list[Any]) -> Optional[np.ndarray]:
  """Zero-copy NumPy view of a constant tensor's buffer, or None (ref :242-263)."""
  raw = buffers[tensor.buffer].data
  if raw is None:
    return None
  dtype = schema.NUMPY_DTYPE[schema.TensorType(tensor.type)]
  data = raw if isinstance(raw, np.ndarray) and raw.dtype == np.dtype(dtype) \
      else np.frombuffer(raw, dtype=dtype)
  if tensor.shape is not None:
    data = np.reshape(data, tensor.shape)
  return data


def parse_fc_bmm_conv_tensors(op: Any, tensors: list[Any], input_index: int = 0,
                              weight_index: int = 1, bias_index: int = 2,
                              output_index: int = 0):
  """(input, weight, bias|None, output) tensors of an FC / conv style op."""
  inp = tensors[op.inputs[input_index]]
  w = tensors[op.inputs[weight_index]]
  bias = None
  if len(op.inputs) > bias_index and op.inputs[bias_index] != -1:
    bias = tensors[op.inputs[bias_index]]
  return inp, w, bias, tensors[op.outputs[output_index]]


def get_subgraph_input_output_operators(subgraph: Any) -> list[qtyping.IOOperator]:
  """Virtual INPUT / OUTPUT ops of a subgraph (ref :318-340)."""
  return [qtyping.IOOperator(inputs=[], outputs=list(subgraph.inputs), op_key=_Op.INPUT),
          qtyping.IOOperator(inputs=list(subgraph.outputs), outputs=[], op_key=_Op.OUTPUT)]


def get_op_scope(op: Any, subgraph_tensors: list[Any], max_length: int = 10000) -> str:
  """Scope string recipes match their regex against: the op's output tensor names
  (inputs when it has no outputs), each followed by ';' (ref :371-417)."""
  def names(ids):
    out = [get_tensor_name(subgraph_tensors[i]) for i in ids if i != -1]
    return [n for n in out if n]
  picked = names(op.outputs) or names(op.inputs)
  scope = ";".join(picked) + (";" if picked else "")
  return scope[:max_length]
