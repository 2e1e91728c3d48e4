"""Tensor-buffer access helpers and the op -> quantized-dimension tables.

ref: utils/tfl_flatbuffer_utils.py:34-112 (tables), :236-263 (get_tensor_name /
get_tensor_data), :115-163 (model file I/O, on this build's own flatbuffer reader/writer).
"""
from __future__ import annotations

import collections
import mmap
import os
import pathlib
from typing import Any, Optional, Union

import numpy as np

from .. import qtyping
from .. import schema
from . import tflite_flatbuffer

Path = Union[str, pathlib.Path]

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

# Every TFLOperationName that names a builtin op maps to the BuiltinOperator of the same name;
# CONV_2D_TRANSPOSE is the schema's TRANSPOSE_CONV (ref :34-89).
TFL_OP_NAME_TO_CODE = qtyping.FrozenMapping({
    **{name: schema.BuiltinOperator[name.name] for name in _Op
       if name.name in schema.BuiltinOperator.__members__},
    _Op.CONV_2D_TRANSPOSE: schema.BuiltinOperator.TRANSPOSE_CONV,
})
TFL_OP_CODE_TO_NAME = qtyping.FrozenMapping({v: k for k, v in TFL_OP_NAME_TO_CODE.items()})


def get_tensor_name(tensor: Any) -> str:
  return schema.tensor_name(tensor)


def get_tensor_data(tensor: Any, buffers: list[Any]) -> Optional[np.ndarray]:
  """Zero-copy NumPy view of a constant tensor's buffer, or None (ref :242-263)."""
  raw = buffers[tensor.buffer].data
  if raw is None:
    return None
  if hasattr(raw, "copy_into"):          # quantized data still in HBM (runtime.HbmArray)
    raw = np.ravel(np.asarray(raw)).view(np.uint8)
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


# ---- model file I/O (ref :115-163) --------------------------------------------------------

def get_model_content(tflite_path: Path) -> memoryview:
  """Read-only, memory-mapped bytes of the model file (weights are paged in on demand).
  (Populating the mapping's page tables ahead of the copies on helper threads --
  madvise(MADV_POPULATE_READ), eight threads, 32 MiB chunks -- was measured and made the file -> file
  run slower, 0.24 against 0.13 s for 1.4 GB: the helpers and the copying thread's own faults
  queue on the mapping's lock.)"""
  with open(tflite_path, "rb") as f:
    if os.fstat(f.fileno()).st_size == 0:
      raise ValueError(f"{tflite_path} is empty")
    mapping = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    try:
      from .. import runtime          # (weights that are views of this mapping can be read from the file)
      runtime.register_file_mapping(mapping, f.fileno())
    except Exception:  # noqa: BLE001 - host-only tools import this module without a GPU runtime
      pass
  return memoryview(mapping)


def get_model_buffer(tflite_path: Path) -> bytearray:
  """A mutable copy of the model file."""
  with open(tflite_path, "rb") as f:
    return bytearray(f.read())


def read_model(tflite_model) -> Any:
  """`.tflite` path or bytes -> ModelT tree whose constant data are zero-copy views."""
  if isinstance(tflite_model, (str, pathlib.Path)):
    return tflite_flatbuffer.read_model(get_model_content(tflite_model))
  if isinstance(tflite_model, (bytes, bytearray, memoryview, mmap.mmap)):
    return tflite_flatbuffer.read_model(tflite_model)
  raise ValueError("Unsupported tflite_model type: %s" % type(tflite_model).__name__)


def write_model(model: Any, output_tflite_file: Path) -> None:
  """Serialize (all buffers inline) and write to `output_tflite_file`."""
  set_file_contents(output_tflite_file, tflite_flatbuffer.write_model(model))


def set_file_contents(path: Path, data) -> None:
  with open(path, "wb") as f:
    f.write(data)


def get_op_side_effect_subgraphs(op: Any) -> list[int]:
  """Subgraphs an op invokes: a composite op's decomposition (ref :342-359)."""
  opts = getattr(op, "builtinOptions2", None)
  if isinstance(opts, schema.StableHLOCompositeOptionsT):
    return [opts.decompositionSubgraphIndex]
  return []


def buffer_to_tensors(model: Any) -> dict[int, list[Any]]:
  """buffer id -> tensors using it, in first-use order (ref :215-223)."""
  out: dict[int, list[Any]] = collections.defaultdict(list)
  for sg in model.subgraphs:
    for t in sg.tensors:
      out[t.buffer].append(t)
  return out
