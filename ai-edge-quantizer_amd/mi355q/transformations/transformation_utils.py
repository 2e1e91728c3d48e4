"""Transformation inputs and GPU bit packing.

ref: transformations/transformation_utils.py:31-69 (TransformationInput),
:226-283 (add_new_constant_buffer / add_new_constant_tensor), :293-353 (pack_data).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional, Union

import numpy as np

from .. import ops
from .. import qtyping
from .. import runtime as rt


@dataclasses.dataclass
class TransformationInput:
  """What a transformation needs to rewrite one tensor (ref :31-69)."""
  tensor_id: int
  model: Any
  subgraph: Any
  producer: int
  consumers: list[int]
  quant_params: Optional[Union[qtyping.UniformQuantParams, qtyping.NonLinearQuantParams]] = None
  buffer_origin: dict[int, Any] = dataclasses.field(default_factory=dict)


def _content_key(view: np.ndarray):
  return (view.size, view[:16].tobytes())


def get_constant_buffer(data: np.ndarray, model: Any, force_duplicate_buffer: bool = False) -> int:
  """Id of a buffer holding exactly `data`'s bytes; appended when the model has none.

  Same sharing rule as ref :119-164: the lookup table is built once per model from the buffers
  present at that time (a later buffer with equal bytes shadows an earlier one) and only grows
  by the buffers added through this function.
  """
  view = np.ravel(np.ascontiguousarray(data)).view(np.uint8)
  table = getattr(model, "_buffers_by_content", None)
  if table is None:
    table = {}
    for i, b in enumerate(model.buffers):
      if b.data is not None:
        if getattr(b.data, "rank", None) is not None and hasattr(b.data, "key"):
          # runtime.RemoteBuffer: a quantized weight whose bytes stayed in another rank's HBM (sharded run that writes a
          # file). It cannot be offered for sharing -- its bytes are not here -- and nothing this function is asked for
          # (scale tensors, zero points, small new constants) is a quantized weight's payload.
          continue
        d = np.ravel(np.asarray(b.data)).view(np.uint8)
        _remember(table, d, i)
    model._buffers_by_content = table
  if not force_duplicate_buffer:
    for held, idx in table.get(_content_key(view), ()):
      if held.size == view.size and np.array_equal(held, view):
        return idx
  buf = qtyping.BufferT()
  buf.data = view
  buf.offset = 0
  buf.size = 0
  model.buffers.append(buf)
  _remember(table, view, len(model.buffers) - 1)
  return len(model.buffers) - 1


def _remember(table: dict, view: np.ndarray, idx: int) -> None:
  bucket = table.setdefault(_content_key(view), [])
  for n, (held, _) in enumerate(bucket):
    if held.size == view.size and np.array_equal(held, view):
      bucket[n] = (held, idx)
      return
  bucket.append((view, idx))


def add_new_constant_buffer(data: np.ndarray, model: Any) -> int:
  return get_constant_buffer(data, model)


def add_new_constant_tensor(tensor_name: bytes, data: np.ndarray, tensor_type, subgraph: Any,
                            model: Any, tensor_shape=None, force_duplicate_buffer: bool = False,
                            quantization=None, allow_tensor_sharing: bool = False) -> int:
  """Appends a constant tensor to the subgraph (its buffer is shared with an existing one of
  equal content unless `force_duplicate_buffer`); returns the tensor id. With
  `allow_tensor_sharing` an existing tensor with the same buffer, shape, type and
  quantized-or-not state is returned instead (ref :167-247)."""
  buffer_id = get_constant_buffer(data, model, force_duplicate_buffer)
  shape = list(data.shape) if tensor_shape is None else list(tensor_shape)
  lookup = getattr(subgraph, "_tensor_lookup", None)
  if allow_tensor_sharing and not force_duplicate_buffer:
    if lookup is None:
      lookup = {(t.buffer, tuple(t.shape or ()), int(t.type), t.quantization is not None): i
                for i, t in enumerate(subgraph.tensors)}
      subgraph._tensor_lookup = lookup
    hit = lookup.get((buffer_id, tuple(shape), int(tensor_type), quantization is not None))
    if hit is not None:
      return hit
  t = qtyping.TensorT()
  t.shape = shape
  t.buffer = buffer_id
  t.type = tensor_type
  t.name = tensor_name
  t.quantization = quantization
  subgraph.tensors.append(t)
  if lookup is not None:
    lookup[(buffer_id, tuple(shape), int(tensor_type), quantization is not None)] = len(subgraph.tensors) - 1
  return len(subgraph.tensors) - 1


def add_op_code(op_code: int, model_op_codes: list[Any], custom_op_name: Optional[str] = None) -> int:
  """Index of the operator code in the model, appended when absent (ref :80-116)."""
  if op_code == qtyping.BuiltinOperator.CUSTOM and custom_op_name is None:
    raise ValueError("Custom string is required for custom op code.")
  for i, existing in enumerate(model_op_codes):
    if existing.builtinCode == op_code and (custom_op_name is None
                                            or existing.customCode == custom_op_name):
      return i
  code = qtyping.OperatorCodeT()
  code.builtinCode = int(op_code)
  if custom_op_name is not None:
    code.customCode = custom_op_name
  model_op_codes.append(code)
  return len(model_op_codes) - 1


def add_new_activation_tensor(tensor_name: bytes, shape, tensor_type, subgraph: Any,
                              quantization=None) -> int:
  """Appends a non-constant tensor (buffer 0); a dynamic dimension (-1) goes to
  shapeSignature with 1 in shape (ref :250-284)."""
  t = qtyping.TensorT()
  shape = None if shape is None else list(shape)
  if shape is not None and -1 in shape:
    t.shapeSignature = shape
    t.shape = [1 if d == -1 else d for d in shape]
  else:
    t.shape = shape
  t.type = tensor_type
  t.name = tensor_name
  t.quantization = quantization
  t.buffer = 0
  subgraph.tensors.append(t)
  return len(subgraph.tensors) - 1


def pack_data(bitwidth: int, data: np.ndarray) -> np.ndarray:
  """int4 / int2 packing, element 0 in the lowest bits (ref :293-353), on the GPU.

  `data` is the flattened quantized tensor viewed as uint8/int8 containers.
  Other bit widths are returned unchanged, as in the reference.
  """
  data = np.asarray(data).reshape(-1)
  if bitwidth not in (2, 4):
    return data
  if data.size == 0:
    return np.zeros(0, np.uint8)
  rt.require_gpu()
  if data.dtype not in (np.uint8, np.int8):
    data = data.astype(np.uint8)
  packed = ops.pack_bits(rt.to_device(data), bitwidth)
  return rt.to_numpy(packed)
