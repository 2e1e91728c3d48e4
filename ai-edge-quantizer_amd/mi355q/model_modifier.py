"""Produces the serialized quantized model (ref: model_modifier.py:90-391).

`ModelModifier.modify_model` works on a structural copy of the float model whose arrays are
views (the float model and its mmap'd weights are never written), turns the parameters into
transformation instructions, applies them in the reference's tensor processing order, and
serializes with this build's own flatbuffer writer: inline when the large buffers total < 256 KiB, otherwise buffers >= 1 KiB go
behind the flatbuffer at 16-byte aligned offsets (`Buffer.offset/size`), written straight into
an mmap of the output file when a path is given.

Every transformation of the reference is applied: QUANTIZE_TENSOR, ADD_QUANTIZE, ADD_DEQUANTIZE, constant
duplication, and the op insertions -- Hadamard rotation (custom op and decomposed RESHAPE / FULLY_CONNECTED / RESHAPE),
OSCAR's multiply (transformations/graph_edits.py, ref transformations/insert_*.py).
"""
from __future__ import annotations

import mmap
import os
from typing import Any, Optional

import numpy as np

from . import qtyping
from . import requant_queue
from . import runtime
from . import transformation_instruction_generator
from . import transformation_performer
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


def apply_transformations(model: Any, params: dict[str, qtyping.TensorTransformationParams]):
  """Parameters -> instructions -> graph edits, in the reference's tensor processing order.
  Returns the instructions (for signature fix-ups)."""
  insts = transformation_instruction_generator.TransformationInstructionsGenerator(
  ).quant_params_to_transformation_insts(params, model)
  transformation_performer.TransformationPerformer().transform_graph(
      insts, model, tensor_processing_order(set(insts), model))
  return insts


# name kept for callers of the first, QUANTIZE_TENSOR-only version of this module
apply_quantize_tensor_transformations = apply_transformations


def _inserted_before_output(insts, kind) -> bool:
  return any(inst.transformation == kind and inst.consumers == [-1]
             for tti in insts.values() for inst in (tti.instructions or []))


def _repoint_signature_outputs(model: Any, suffix: str) -> None:
  """A Q / DQ op inserted in front of a graph output created `<name><suffix>`; signature
  outputs that named the old tensor follow it (ref :201-255)."""
  for sig in model.signatureDefs or []:
    sg = model.subgraphs[sig.subgraphIndex]
    for out_id in sg.outputs:
      new_name = tfl_flatbuffer_utils.get_tensor_name(sg.tensors[out_id])
      for item in sig.outputs or []:
        old = tfl_flatbuffer_utils.get_tensor_name(sg.tensors[item.tensorIndex])
        if old + suffix == new_name:
          item.tensorIndex = out_id
          break


def _large_buffer_bytes(model: Any) -> int:
  total = 0
  for b in model.buffers or []:
    if b.data is not None:
      n = b.data.nbytes if hasattr(b.data, "nbytes") else len(b.data)
      if n >= _MIN_EXTERNAL_BUFFER_BYTES:
        total = (total + n + 15) & ~15
  return total


def serialize_model(model: Any, serialize_to_path: Optional[tfl_flatbuffer_utils.Path] = None, sink=None):
  """ref :184-199 (layout choice), :290-391 (the two serializers). `sink(total_bytes)` -> a writable
  buffer to build a large model into (e.g. its place inside a container file's mapping)."""
  if _large_buffer_bytes(model) < _INLINE_LIMIT_BYTES:
    for b in model.buffers or []:
      if hasattr(b.data, "copy_into"):        # device resident: the inline writer wants bytes
        b.data = np.ravel(np.asarray(b.data)).view(np.uint8)
    out = tflite_flatbuffer.write_model(model)
    if serialize_to_path:
      tfl_flatbuffer_utils.set_file_contents(serialize_to_path, out)
    return out

  caller_sink = sink
  opened: list = []

  def sink(total: int):
    if caller_sink is not None:
      return caller_sink(total)
    if not serialize_to_path:
      return None
    # (no O_TRUNC: dropping an old file's pages from the page cache costs as much as writing them --
    # 13 ms for 186 MB -- and the new bytes overwrite them anyway; the length is set below)
    fd = os.open(serialize_to_path, os.O_RDWR | os.O_CREAT, 0o644)
    if os.fstat(fd).st_size != total:
      os.ftruncate(fd, total)
    mapping = mmap.mmap(fd, total)
    runtime.register_output_mapping(mapping, fd)      # device-resident buffers: pinned staging + pwrite()
    opened.append((mapping, fd))
    return mapping

  # (no msync: a shared mapping is coherent with the page cache, so readers see the bytes at once,
  # and they reach the disk when the kernel writes them back -- what a plain write() gives too;
  # the synchronous flush cost 19 ms of a 147 ms file -> file run)
  try:
    def before_values():      # every payload has its place and is on its way: the one wait for the accelerator
      runtime.mark("writer: every payload handed over (host)")
      requant_queue.complete_active()
      runtime.mark("writer: values in (host)")
      transformation_utils.verify_late_constants(model)     # (may raise SharingNotDecided: modify_model builds again)
    return tflite_flatbuffer.serialize_with_external_buffers(model, _MIN_EXTERNAL_BUFFER_BYTES, sink, before_values=before_values,
                                                             on_layout=lambda: runtime.mark("writer: flatbuffer laid out (host)"))
  finally:
    runtime.mark("writer: flatbuffer complete (host)")
    if opened:
      runtime.finish_downloads()
    runtime.mark("writer: file written (host)")
    for mapping, fd in opened:
      runtime.forget_output_mapping(mapping)
      os.close(fd)


class ModelModifier:
  def __init__(self, float_model: Any):
    self._model = float_model

  def modify_model(self, params: dict[str, qtyping.TensorTransformationParams],
                   serialize_to_path: Optional[tfl_flatbuffer_utils.Path] = None,
                   enable_progress_bar: Optional[bool] = None, sink=None):
    del enable_progress_bar
    # Inside a caller's requant_queue.batching() block that covers this call (Quantizer.quantize with a path), constants
    # still in HBM are laid out unread (transformation_utils.get_constant_buffer); should two of them turn out equal --
    # the reference would have shared one buffer -- everything is built once more with the values read first.
    late = (serialize_to_path is not None or sink is not None) and requant_queue.active() is not None
    if late:
      runtime._LATE_CONSTANTS[0] += 1   # pylint: disable=protected-access
      try:
        return self._modify(params, serialize_to_path, sink)
      except transformation_utils.SharingNotDecided:
        pass
      finally:
        runtime._LATE_CONSTANTS[0] -= 1   # pylint: disable=protected-access
    return self._modify(params, serialize_to_path, sink)

  def _modify(self, params, serialize_to_path, sink):
    quantized = copy_with_views(self._model)
    insts = apply_transformations(quantized, params)
    runtime.mark("writer: transformations applied (host)")
    if _inserted_before_output(insts, _T.ADD_DEQUANTIZE):
      _repoint_signature_outputs(quantized, "_dequant")
    if _inserted_before_output(insts, _T.ADD_QUANTIZE):
      _repoint_signature_outputs(quantized, "_quantized")
    self.quantized_model_object = quantized
    return serialize_model(quantized, serialize_to_path, sink)
