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


class SharingNotDecided(Exception):
  """A constant's bytes were still being computed when its buffer was added, it was assumed to equal no other such
  constant of its size, and the values say otherwise: the model has to be built again with the values read first."""


def _still_on_device(data) -> bool:
  return isinstance(data, rt.HbmArray) and getattr(data, "_host", None) is None and rt.late_constants_allowed()


def get_constant_buffer(data: np.ndarray, model: Any, force_duplicate_buffer: bool = False) -> int:
  """Id of a buffer holding exactly `data`'s bytes; appended when the model has none.

  Same sharing rule as ref :119-164: the lookup table is built once per model from the buffers
  present at that time (a later buffer with equal bytes shadows an earlier one) and only grows
  by the buffers added through this function.

  A constant whose values are still in HBM (blockwise scales written by the launch that quantized: 1.4 MB per
  FULLY_CONNECTED of a C3 layer) would have to be read -- a wait for the GPU per tensor -- only to learn that it
  equals no other buffer. While a file is being written it is added unread when no HOST buffer has its size (equal
  bytes need equal sizes), remembered by size, and verify_late_constants() compares the device-resident constants
  of equal size with each other once their values exist; a match raises SharingNotDecided and the caller builds
  the model again the slow way (values first). The result is the reference's sharing either way.
  """
  table = getattr(model, "_buffers_by_content", None)
  if table is None:
    table = {}
    late: dict[int, list] = {}
    for i, b in enumerate(model.buffers):
      if b.data is not None:
        if isinstance(b.data, rt.RemoteBuffer):    # (isinstance, not getattr: an HbmArray answers unknown attributes from its host copy)
          # runtime.RemoteBuffer: a quantized weight whose bytes stayed in another rank's HBM (sharded run that writes a
          # file). It cannot be offered for sharing -- its bytes are not here -- and nothing this function is asked for
          # (scale tensors, zero points, small new constants) is a quantized weight's payload.
          continue
        if _still_on_device(b.data):
          late.setdefault(b.data.nbytes, []).append((b.data, i))
          continue
        d = np.ravel(np.asarray(b.data)).view(np.uint8)
        _remember(table, d, i)
    model._buffers_by_content = table
    model._late_constants = late
  late = model._late_constants
  if _still_on_device(data) and not force_duplicate_buffer and data.nbytes not in table.get(_SIZES, ()):
    buf = qtyping.BufferT()
    buf.data = data
    buf.offset = 0
    buf.size = 0
    model.buffers.append(buf)
    late.setdefault(data.nbytes, []).append((data, len(model.buffers) - 1))
    return len(model.buffers) - 1
  view = np.ravel(np.ascontiguousarray(data)).view(np.uint8)
  for held, idx in late.pop(view.size, ()):          # device-resident buffers of this size: now they have to be read
    _remember(table, np.ravel(np.asarray(held)).view(np.uint8), idx)
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


def verify_late_constants(model: Any) -> None:
  """The device-resident constants get_constant_buffer() added unread, compared by size class now that their values
  exist (two checksums per constant in one pass over each class, exact comparison where both tie)."""
  import torch
  late = getattr(model, "_late_constants", None) or {}
  for nbytes, entries in late.items():
    added = list(entries)      # (also the device-resident buffers the model already had when the table was built)
    if len(added) < 2:
      continue
    # two checksums per constant, a piece at a time: the int64 widening of a whole size class (16 x its bytes: GiBs for the
    # blockwise scales of a 32-layer model) never exists, and the class's marks come back in one copy
    first = added[0][0].device_tensor
    marks = torch.zeros((len(added), 2), dtype=torch.int64, device=first.device)
    piece = 4 << 20
    weights = (torch.arange(min(piece, nbytes), device=first.device, dtype=torch.int64) % 65521) + 1
    for r, (d, _) in enumerate(added):
      flat = d.device_tensor.contiguous().reshape(-1).view(torch.uint8)
      for o in range(0, nbytes, piece):
        part = flat[o:o + piece]
        marks[r, 0] += part.sum(dtype=torch.int64)
        # (the weight of a byte depends on its place within its piece and on the piece's index: same bytes, same marks)
        marks[r, 1] += (part.to(torch.int64) * weights[:part.numel()]).sum() * (o // piece % 65521 + 1)
    marks = marks.cpu().numpy()
    seen: dict = {}
    for row, (d, i) in zip(map(tuple, marks), added):
      for other, j in seen.get(row, ()):
        if torch.equal(other.device_tensor.reshape(-1).view(torch.uint8), d.device_tensor.reshape(-1).view(torch.uint8)):
          raise SharingNotDecided(f"buffers {j} and {i} hold the same {nbytes} bytes")
      seen.setdefault(row, []).append((d, i))


_SIZES = "byte sizes of the host buffers in the table"


def _remember(table: dict, view: np.ndarray, idx: int) -> None:
  table.setdefault(_SIZES, set()).add(view.size)
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
