"""Executes transformation instructions on a ModelT tree (ref: transformation_performer.py).

Instructions name ops by their index in the *float* graph. Inserting an op shifts every later
index, so two maps are kept per subgraph: float index -> current index, and the current index
of every op inserted so far (later instructions of the same tensor that follow an inserted op
are re-anchored to it).
"""
from __future__ import annotations

from typing import Any, Optional, Sequence

from . import qtyping
from .transformations import graph_edits
from .transformations import quantize_tensor
from .transformations import transformation_utils

_T = qtyping.QuantTransformation


def _unsupported(name: str):
  def fn(_):
    raise NotImplementedError(
        f"{name} rewrites the graph with ops this build does not emit (outside its scope)")
  return fn


def _deprecated(_):
  raise NotImplementedError("This transformation is deprecated. Please contact AI Edge Quantizer team"
                            " if you see this error.")


class TransformationPerformer:
  def __init__(self):
    self._apply = {
        _T.QUANTIZE_TENSOR: quantize_tensor.quantize_tensor,
        _T.ADD_QUANTIZE: graph_edits.insert_quant,
        _T.ADD_DEQUANTIZE: graph_edits.insert_dequant,
        _T.DUPLICATE_BUFFER: graph_edits.duplicate_buffer,
        _T.DUPLICATE_TENSOR: graph_edits.duplicate_tensor,
        _T.EMULATED_SUBCHANNEL: _deprecated,      # ref transformation_utils.py:286-290
        # (the custom op's options are a FlexBuffer: utils/flexbuffer.py)
        _T.INSERT_HADAMARD_ROTATION: graph_edits.insert_hadamard_rotation,
        _T.INSERT_DECOMPOSED_HADAMARD_ROTATION: graph_edits.insert_decomposed_hadamard_rotation,
        _T.INSERT_MULTIPLY: graph_edits.insert_multiply,
    }
    self._current_index: list[list[int]] = []    # [subgraph][float op index] -> index now
    self._inserted: list[list[int]] = []         # [subgraph] -> index now of each inserted op
    self._buffer_origin: dict[int, Any] = {}

  def _producer_now(self, producer: Optional[int], sg: int) -> int:
    if producer is None or producer < 0:
      return -1
    n = len(self._current_index[sg])
    return self._current_index[sg][producer] if producer < n else self._inserted[sg][producer - n]

  def _run(self, tti: qtyping.TensorTransformationInsts, index: int, model: Any) -> None:
    sg = tti.subgraph_id
    inst = tti.instructions[index]
    producer = self._producer_now(inst.producer, sg)
    consumers = [-1 if c == -1 else self._current_index[sg][c] for c in inst.consumers]
    info = self._apply[inst.transformation](transformation_utils.TransformationInput(
        inst.tensor_id, model, model.subgraphs[sg], producer, consumers, inst.parameters,
        self._buffer_origin))
    # later instructions of this tensor that serve the same consumers now start from the new
    # tensor (and, when an op was inserted, from that op)
    added = info.num_ops_added > 0
    if added or inst.transformation == _T.DUPLICATE_TENSOR:
      if added:
        self._inserted[sg].append(info.op_id + info.num_ops_added - 1)
      for later in tti.instructions[index + 1:]:
        for c in later.consumers:
          if c in inst.consumers:
            if added:
              later.producer = len(self._current_index[sg]) + len(self._inserted[sg]) - 1
            later.tensor_id = info.output_tensor_id
    # the op went in right before the nearest consumer (after the producer for graph outputs)
    first = min(inst.consumers)
    start = first if first >= 0 else inst.producer + 1
    for k in range(start, len(self._current_index[sg])):
      self._current_index[sg][k] += info.num_ops_added

  def transform_graph(self, transformation_instructions: dict[str, qtyping.TensorTransformationInsts],
                      tflite_model: Any, tensor_processing_order: Optional[Sequence[str]] = None,
                      enable_progress_bar: Optional[bool] = None) -> None:
    del enable_progress_bar
    self._current_index = [list(range(len(sg.operators))) for sg in tflite_model.subgraphs]
    self._inserted = [[] for _ in tflite_model.subgraphs]
    self._buffer_origin = {}
    order = transformation_instructions.keys() if tensor_processing_order is None else tensor_processing_order
    for name in order:
      tti = transformation_instructions[name]
      for i, inst in enumerate(tti.instructions or []):
        if inst.transformation != _T.NO_QUANTIZE:
          self._run(tti, i, tflite_model)
    self._buffer_origin = {}
