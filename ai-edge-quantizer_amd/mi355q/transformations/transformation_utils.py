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


def add_new_constant_buffer(data: np.ndarray, model: Any) -> int:
  """Appends a buffer holding `data`'s bytes; returns its id."""
  buf = qtyping.BufferT()
  buf.data = np.frombuffer(np.ascontiguousarray(data).tobytes(), dtype=np.uint8)
  buf.offset = 0
  buf.size = 0
  model.buffers.append(buf)
  return len(model.buffers) - 1


def add_new_constant_tensor(tensor_name: bytes, data: np.ndarray, tensor_type, subgraph: Any,
                            model: Any, tensor_shape=None, force_duplicate_tensor_name=False) -> int:
  """Appends a constant tensor (and its buffer) to the subgraph; returns its id."""
  del force_duplicate_tensor_name
  t = qtyping.TensorT()
  t.shape = list(data.shape) if tensor_shape is None else list(tensor_shape)
  t.buffer = add_new_constant_buffer(data, model)
  t.type = tensor_type
  t.name = tensor_name
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
