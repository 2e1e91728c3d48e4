"""Produces the serialized quantized model (ref: model_modifier.py:90-391).

`ModelModifier.modify_model` works on a structural copy of the float model whose arrays are
views (the float model and its mmap'd weights are never written), applies the QUANTIZE_TENSOR
transformation in the reference's tensor processing order, and serializes with this build's own
flatbuffer writer: inline when the large buffers total < 256 KiB, otherwise buffers >= 1 KiB go
behind the flatbuffer at 16-byte aligned offsets (`Buffer.offset/size`), written straight into
an mmap of the output file when a path is given.

Transformations that rewrite the graph (ADD_QUANTIZE / ADD_DEQUANTIZE / Hadamard op insertion /
buffer or tensor duplication) are outside this build's scope and raise NotImplementedError.
"""
from __future__ import annotations

import mmap
from typing import Any, Optional

import numpy as np

from . import qtyping
from .transformations import quantize_tensor
from .transformations import transformation_utils
from .utils import tfl_flatbuffer_utils
from .utils import tflite_flatbuffer

_T = qtyping.QuantTransformation
_MIN_EXTERNAL_BUFFER_BYTES = 1024
_INLINE_LIMIT_BYTES = 256 * 1024


def copy_with_views(value: Any) -> Any:
  """Deep copy of the table tree that shares array storage (ref :79-101)."""
  if isinstance(value, np.ndarray):
    return value.view()
  if isinstance(value, list):
    return [copy_with_views(v) for v in value]
  if isinstance(value, tflite_flatbuffer.TableT):
    out = type(value).__new__(type(value))
    for k, v in value.__dict__.items():
      out.__dict__[k] = set(v) if k == "_explicit" else copy_with_views(v)
    out.__dict__.pop("_buffers_by_content", None)
    return out
  return value


def tensor_processing_order(names: set[str], model: Any) -> list[str]:
  """Tensor names grouped by buffer, in first-use order, so that the last user of a shared
  buffer is processed last (ref :120-147)."""
  order = []
  for tensors in tfl_flatbuffer_utils.buffer_to_tensors(model).values():
    for t in tensors:
      name = tfl_flatbuffer_utils.get_tensor_name(t)
      if name in names:
        order.append(name)
  return order


def _instruction(p: qtyping.TensorTransformationParams):
  """The single QUANTIZE_TENSOR instruction of a tensor, None for NO_QUANTIZE, or raises."""
  links = list(p.consumers or []) + ([p.producer] if p.producer is not None else [])
  wanted = {t for link in links for t in link.transformations}
  if wanted <= {_T.NO_QUANTIZE}:
    return None
  if wanted != {_T.QUANTIZE_TENSOR}:
    raise NotImplementedError(
        f"tensor {p.tensor_name}: transformations {sorted(t.name for t in wanted)} need graph"
        " rewriting, which is outside this build's scope")
  first = links[0].parameters
  if any(link.parameters != first for link in links[1:]):
    raise NotImplementedError(f"tensor {p.tensor_name}: consumers disagree on parameters")
  return first


def apply_quantize_tensor_transformations(model: Any, params: dict[str, qtyping.TensorTransformationParams]) -> None:
  """QUANTIZE_TENSOR for every constant whose consumers all ask for it with equal parameters
  (the case the reference's instruction generator leaves as a single QUANTIZE_TENSOR
  instruction). As in the reference, a tensor name resolves to its last occurrence in the
  model (transformation_instruction_generator.py:237-245)."""
  where: dict[str, tuple[Any, int]] = {}
  for sg in model.subgraphs:
    for tid, tensor in enumerate(sg.tensors):
      where[tfl_flatbuffer_utils.get_tensor_name(tensor)] = (sg, tid)
  todo = {name: inst for name, p in params.items() if (inst := _instruction(p)) is not None}
  buffer_origin: dict[int, Any] = {}
  for name in tensor_processing_order(set(todo), model):
    sg, tid = where[name]
    quantize_tensor.quantize_tensor(transformation_utils.TransformationInput(
        tensor_id=tid, model=model, subgraph=sg, producer=-1, consumers=[], quant_params=todo[name],
        buffer_origin=buffer_origin))


def _large_buffer_bytes(model: Any) -> int:
  total = 0
  for b in model.buffers or []:
    if b.data is not None:
      n = np.asarray(b.data).nbytes if isinstance(b.data, np.ndarray) else len(b.data)
      if n >= _MIN_EXTERNAL_BUFFER_BYTES:
        total = (total + n + 15) & ~15
  return total


def serialize_model(model: Any, serialize_to_path: Optional[tfl_flatbuffer_utils.Path] = None):
  """ref :184-199 (layout choice), :290-391 (the two serializers)."""
  if _large_buffer_bytes(model) < _INLINE_LIMIT_BYTES:
    out = tflite_flatbuffer.write_model(model)
    if serialize_to_path:
      tfl_flatbuffer_utils.set_file_contents(serialize_to_path, out)
    return out

  def sink(total: int):
    if not serialize_to_path:
      return None
    with open(serialize_to_path, "w+b") as f:
      f.truncate(total)
      return mmap.mmap(f.fileno(), total)

  out = tflite_flatbuffer.serialize_with_external_buffers(model, _MIN_EXTERNAL_BUFFER_BYTES, sink)
  if isinstance(out, mmap.mmap):
    out.flush()
  return out


class ModelModifier:
  def __init__(self, float_model: Any):
    self._model = float_model

  def modify_model(self, params: dict[str, qtyping.TensorTransformationParams],
                   serialize_to_path: Optional[tfl_flatbuffer_utils.Path] = None,
                   enable_progress_bar: Optional[bool] = None):
    del enable_progress_bar
    quantized = copy_with_views(self._model)
    apply_quantize_tensor_transformations(quantized, params)
    self.quantized_model_object = quantized
    return serialize_model(quantized, serialize_to_path)
