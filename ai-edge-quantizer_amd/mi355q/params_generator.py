"""Model-level loop that calls the registered materializers.

Restatement of the op loop of ref: params_generator.py:69-185 for the ops this
build registers: for every subgraph op (+ the virtual INPUT / OUTPUT ops) resolve
the recipe, look the materializer up in the registry and merge the per-tensor
results. Ops the recipe / policy leaves alone get NO_QUANTIZE; constants shared between
differently quantized users are marked for duplication (ref :291-463).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from . import algorithm_manager
from . import default_policy
from . import qtyping
from . import requant_queue
from .algorithms.utils import common_utils
from .utils import tfl_flatbuffer_utils


_T = qtyping.QuantTransformation
_FLOAT_SOURCE = (_T.ADD_QUANTIZE, _T.NO_QUANTIZE, _T.INSERT_HADAMARD_ROTATION,
                 _T.INSERT_DECOMPOSED_HADAMARD_ROTATION, _T.INSERT_MULTIPLY)
_QUANTIZED_SOURCE = (_T.QUANTIZE_TENSOR, _T.ADD_DEQUANTIZE)


def _links_compatible(a: qtyping.OpToTensorParams, b: qtyping.OpToTensorParams) -> bool:
  """Can two ops read the same stored tensor? Same request, or both read it as float, or both
  read it quantized with equal parameters (ref :527-559)."""
  if a.transformations == b.transformations and (
      a.parameters == b.parameters or (a.parameters is None and b.parameters is None)):
    return True
  if a.transformations[0] in _FLOAT_SOURCE and b.transformations[0] in _FLOAT_SOURCE:
    return True
  return (a.transformations[0] in _QUANTIZED_SOURCE and b.transformations[0] in _QUANTIZED_SOURCE
          and a.parameters == b.parameters)


class ParamsGenerator:
  def __init__(self, float_tflite: Any):
    self.float_model = float_tflite
    self.model_quant_results: dict[str, qtyping.TensorTransformationParams] = {}
    self._tensor_quant_params_cache = common_utils.TensorQuantParamsCache()
    self.buffer_to_tensors = tfl_flatbuffer_utils.buffer_to_tensors(float_tflite)
    seen: set[str] = set()
    for sg in float_tflite.subgraphs:
      for t in sg.tensors:
        name = tfl_flatbuffer_utils.get_tensor_name(t)
        if name in seen:
          raise ValueError(
              "Tensor name %s is not unique in the model. Please check your model and rename the"
              " tensor as ParamsGenerator assumes tensor names are unique." % name)
        seen.add(name)

  # ---- shared constants (ref :291-463) -----------------------------------------------------
  def _is_constant(self, tensor: Any) -> bool:
    return self.float_model.buffers[tensor.buffer].data is not None

  def _consumers_agree(self, tensor: Any) -> bool:
    p = self.model_quant_results.get(tfl_flatbuffer_utils.get_tensor_name(tensor))
    if p is None or p.consumers is None or len(p.consumers) < 2:
      return True
    if all(_links_compatible(c, p.consumers[0]) for c in p.consumers[1:]):
      return True
    if self._is_constant(tensor):
      return False
    raise RuntimeError(
        f"The tensor {tensor.name} consumers do not have the same quantization parameters. Please"
        " modify your quantization recipe to make sure the two tensors have the same quantization"
        " settings.")

  def _tensors_agree(self, t1: Any, t2: Any) -> bool:
    p1 = self.model_quant_results.get(tfl_flatbuffer_utils.get_tensor_name(t1))
    p2 = self.model_quant_results.get(tfl_flatbuffer_utils.get_tensor_name(t2))
    if p1 is None or p2 is None:
      return True
    ok = True
    if p1.producer is None or p2.producer is None:
      ok = p1.producer == p2.producer
    else:
      ok = _links_compatible(p1.producer, p2.producer)
    if ok:
      if p1.consumers is None or p2.consumers is None:
        ok = p1.consumers == p2.consumers
      else:
        ok = _links_compatible(p1.consumers[0], p2.consumers[0])
    if ok:
      return True
    if self._is_constant(t1):
      return False
    raise RuntimeError(
        f"The tensors {t1.name} and {t2.name} do not have the same quantization parameters even"
        " though they share the same buffer. Please modify your quantization recipe to make sure"
        " the two tensors have the same quantization settings.")

  def _check_and_fix_buffer_sharing(self) -> None:
    """A constant whose consumers disagree is marked for tensor duplication; constants sharing
    a buffer but not a quantization are marked for buffer duplication (all but the last user
    of the buffer). The marks are DUPLICATE_* transformations put first for every consumer."""
    dup_buffers, dup_tensors = [], []
    for buffer_idx, tensors in self.buffer_to_tensors.items():
      if not tensors or buffer_idx == 0:
        continue
      for t in tensors:
        if not self._consumers_agree(t):
          dup_tensors.append(tfl_flatbuffer_utils.get_tensor_name(t))
      if tfl_flatbuffer_utils.get_tensor_name(tensors[0]) in dup_tensors:
        dup_buffers.append(buffer_idx)
        continue
      for t2 in tensors[1:]:
        if (tfl_flatbuffer_utils.get_tensor_name(t2) in dup_tensors
            or not self._tensors_agree(tensors[0], t2)):
          dup_buffers.append(buffer_idx)
          break
    for buffer_idx in dup_buffers:
      for t in self.buffer_to_tensors[buffer_idx][:-1]:
        for link in self.model_quant_results[tfl_flatbuffer_utils.get_tensor_name(t)].consumers:
          link.transformations.insert(0, _T.DUPLICATE_BUFFER)
    for name in dup_tensors:
      for link in self.model_quant_results[name].consumers:
        link.transformations.insert(0, _T.DUPLICATE_TENSOR)

  def _no_quant_results(self, op_id: int, op: Any, tensors: list[Any]):
    def link():   # a fresh record per tensor: duplication marks are inserted in place later
      return qtyping.OpToTensorParams(subgraph_op_id=op_id,
                                      transformations=[qtyping.QuantTransformation.NO_QUANTIZE])
    out = []
    for ids, inbound in ((op.inputs, True), (op.outputs, False)):
      for tid in ids:
        if tid == -1:
          continue
        name = tfl_flatbuffer_utils.get_tensor_name(tensors[tid])
        out.append(qtyping.TensorTransformationParams(
            tensor_name=name, consumers=[link()] if inbound else None,
            producer=None if inbound else link()))
    return out

  def _merge(self, results) -> None:
    for r in results:
      cur = self.model_quant_results.get(r.tensor_name)
      if cur is None:
        self.model_quant_results[r.tensor_name] = r
        continue
      if r.producer is not None:
        if cur.producer is not None:      # a tensor has exactly one source op (ref :229-238)
          raise RuntimeError(
              "Tensor %s received multiple quantization parameters from the source op, which"
              " should not happen as every tensor should have only one source op." % r.tensor_name)
        cur.producer = r.producer
      if r.consumers:
        cur.consumers = (cur.consumers or []) + list(r.consumers)

  def plan_ops(self, model_recipe_manager) -> list[tuple]:
    """The op walk without any arithmetic: one (graph_info, op, op_id, op_key, algorithm, config)
    item per real or virtual op, in the reference's order; algorithm NO_QUANTIZE (config None) for
    ops the recipe, the policy or a skipped composite leaves alone (ref :100-160)."""
    codes = self.float_model.operatorCodes
    no_q = algorithm_manager.AlgorithmName.NO_QUANTIZE
    skip_subgraphs: set[int] = set()     # decompositions of composite ops left unquantized
    items = []
    for sg_ind, subgraph in enumerate(self.float_model.subgraphs):
      graph_info = qtyping.GraphInfo(subgraph.tensors, self.float_model.buffers)
      ops = list(subgraph.operators) + tfl_flatbuffer_utils.get_subgraph_input_output_operators(subgraph)
      for op_id, op in enumerate(ops):
        if isinstance(op, qtyping.IOOperator):
          op_key, op_id = op.op_key, -1
        else:
          code = codes[op.opcodeIndex].builtinCode
          op_key = tfl_flatbuffer_utils.TFL_OP_CODE_TO_NAME.get(code)
          if op_key is None:
            items.append((graph_info, op, op_id, None, no_q, None))
            continue
        scope = tfl_flatbuffer_utils.get_op_scope(op, subgraph.tensors)
        alg, cfg = model_recipe_manager.get_quantization_configs(op_key, scope)
        if sg_ind in skip_subgraphs or default_policy.is_non_quantizable_composite_op(op):
          alg = no_q
        if alg == no_q:
          skip_subgraphs.update(tfl_flatbuffer_utils.get_op_side_effect_subgraphs(op))
          cfg = None
        items.append((graph_info, op, op_id, op_key, alg, cfg))
    return items

  def materialize_op(self, item: tuple, model_qsvs: dict[str, qtyping.QSV]):
    """Per-tensor results of one planned op (the GPU work happens here)."""
    graph_info, op, op_id, op_key, alg, cfg = item
    if alg == algorithm_manager.AlgorithmName.NO_QUANTIZE:
      return self._no_quant_results(op_id, op, graph_info.subgraph_tensors)
    fn = algorithm_manager.get_quantization_func(alg, op_key, qtyping.QuantizeMode.MATERIALIZE)
    return fn(op_info=qtyping.OpInfo(op, op_key, op_id, cfg), graph_info=graph_info,
              tensor_name_to_qsv=model_qsvs, tensor_quant_params_cache=self._tensor_quant_params_cache)

  def finish(self, per_op_results) -> dict[str, qtyping.TensorTransformationParams]:
    """Merges per-op results (in plan order) and applies the shared-constant fix-ups."""
    for results in per_op_results:
      self._merge(results)
    self._check_and_fix_buffer_sharing()
    return self.model_quant_results

  def generate_quantization_parameters(self, model_recipe_manager,
                                       model_qsvs: Optional[dict[str, qtyping.QSV]] = None,
                                       enable_progress_bar: bool | None = None):
    del enable_progress_bar
    if model_recipe_manager.need_calibration() and not model_qsvs:
      raise RuntimeError(
          "Model quantization statistics values (QSVs) are required for the input recipe. This"
          " can be obtained by running calibration on sample dataset.")
    model_qsvs = model_qsvs if model_qsvs is not None else {}
    # The op loop only enqueues the fused min/max weight requantizations; equally shaped weights
    # leave in one batched launch per group when the queue flushes (requant_queue), at the latest
    # before the shared-constant checks of finish() compare parameters by value.
    plan = self.plan_ops(model_recipe_manager)
    self.prefetch(plan, model_qsvs)
    try:
      with requant_queue.batching() as queue:
        per_op = [self.materialize_op(item, model_qsvs) for item in plan]
    finally:
      self.release_derived(model_qsvs)
    self.batch_stats = dict(queue.stats)
    return self.finish(per_op)

  @staticmethod
  def prefetch(plan_items, model_qsvs) -> None:
    """Work that is cheaper done for all of `plan_items` at once than op by op: the uploads of the weights the
    quantized ops read from a mapped model file (started now, in plan order, on the library's upload thread:
    runtime.prefetch_uploads) and the damped inverses of the small GPTQ Hessians (batched on the device)."""
    ParamsGenerator.prefetch_weights(plan_items)
    if model_qsvs and any(isinstance(q, dict) and "hessian" in q for q in model_qsvs.values()):
      from .algorithms.uniform_quantize import gptq
      gptq.prefetch_hessian_inverses(plan_items, model_qsvs)

  @staticmethod
  def prefetch_weights(plan_items, submit: bool = True) -> int:
    """The constant operands (1 MiB and more, views of a mapped model file) of the ops the plan quantizes."""
    from . import runtime as rt
    no_q = algorithm_manager.AlgorithmName.NO_QUANTIZE
    arrays, seen = [], set()
    for graph_info, op, _, _, alg, _ in plan_items:
      if alg == no_q:
        continue
      for tid in getattr(op, "inputs", ()):
        if tid == -1:
          continue
        buffer_id = graph_info.subgraph_tensors[tid].buffer
        data = graph_info.buffers[buffer_id].data if buffer_id else None
        if isinstance(data, np.ndarray) and data.nbytes >= (1 << 20) and buffer_id not in seen:
          seen.add(buffer_id)
          arrays.append(data)
    return rt.prefetch_uploads(arrays, submit) if arrays else 0

  @staticmethod
  def release_derived(model_qsvs) -> None:
    """Derived device results cached on the statistics (the damped Hessian inverse, one d x d
    float32 per GPTQ Hessian, shared by the ops that read the same activation) have served their
    last consumer: give the HBM back instead of keeping it for as long as the QSVs live."""
    for qsv in (model_qsvs or {}).values():
      h = qsv.get("hessian") if isinstance(qsv, dict) else None
      if hasattr(h, "cache"):
        h.cache.clear()
    from . import runtime as rt
    if requant_queue.active() is None:      # (an outer block -- the writer's -- still meets announced uploads: it releases them)
      rt.release_upload_files()
