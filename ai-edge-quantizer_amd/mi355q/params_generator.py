"""Model-level loop that calls the registered materializers.

Restatement of the op loop of ref: params_generator.py:69-185 for the ops this
build registers: for every subgraph op (+ the virtual INPUT / OUTPUT ops) resolve
the recipe, look the materializer up in the registry and merge the per-tensor
results. Ops the recipe / policy leaves alone get NO_QUANTIZE. (Buffer-sharing fix-ups of the
reference, ref :291-463, concern graph surgery and are not restated.)
"""
from __future__ import annotations

from typing import Any, Optional

from . import algorithm_manager
from . import default_policy
from . import qtyping
from .algorithms.utils import common_utils
from .utils import tfl_flatbuffer_utils


class ParamsGenerator:
  def __init__(self, float_tflite: Any):
    self.float_model = float_tflite
    self.model_quant_results: dict[str, qtyping.TensorTransformationParams] = {}
    self._tensor_quant_params_cache = common_utils.TensorQuantParamsCache()

  def _no_quant_results(self, op_id: int, op: Any, tensors: list[Any]):
    link = qtyping.OpToTensorParams(subgraph_op_id=op_id,
                                    transformations=[qtyping.QuantTransformation.NO_QUANTIZE])
    out = []
    for ids, inbound in ((op.inputs, True), (op.outputs, False)):
      for tid in ids:
        if tid == -1:
          continue
        name = tfl_flatbuffer_utils.get_tensor_name(tensors[tid])
        out.append(qtyping.TensorTransformationParams(
            tensor_name=name, consumers=[link] if inbound else None,
            producer=None if inbound else link))
    return out

  def _merge(self, results) -> None:
    for r in results:
      cur = self.model_quant_results.get(r.tensor_name)
      if cur is None:
        self.model_quant_results[r.tensor_name] = r
        continue
      if r.producer is not None:
        cur.producer = r.producer
      if r.consumers:
        cur.consumers = (cur.consumers or []) + list(r.consumers)

  def generate_quantization_parameters(self, model_recipe_manager,
                                       model_qsvs: Optional[dict[str, qtyping.QSV]] = None,
                                       enable_progress_bar: bool | None = None):
    del enable_progress_bar
    if model_recipe_manager.need_calibration() and not model_qsvs:
      raise RuntimeError(
          "Model quantization statistics values (QSVs) are required for the input recipe. This"
          " can be obtained by running calibration on sample dataset.")
    model_qsvs = model_qsvs if model_qsvs is not None else {}
    codes = self.float_model.operatorCodes
    skip_subgraphs: set[int] = set()     # decompositions of composite ops left unquantized
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
            self._merge(self._no_quant_results(op_id, op, subgraph.tensors))
            continue
        scope = tfl_flatbuffer_utils.get_op_scope(op, subgraph.tensors)
        alg, cfg = model_recipe_manager.get_quantization_configs(op_key, scope)
        if sg_ind in skip_subgraphs or default_policy.is_non_quantizable_composite_op(op):
          alg = algorithm_manager.AlgorithmName.NO_QUANTIZE
        if alg == algorithm_manager.AlgorithmName.NO_QUANTIZE:
          skip_subgraphs.update(tfl_flatbuffer_utils.get_op_side_effect_subgraphs(op))
          self._merge(self._no_quant_results(op_id, op, subgraph.tensors))
          continue
        fn = algorithm_manager.get_quantization_func(alg, op_key, qtyping.QuantizeMode.MATERIALIZE)
        self._merge(fn(op_info=qtyping.OpInfo(op, op_key, op_id, cfg), graph_info=graph_info,
                       tensor_name_to_qsv=model_qsvs,
                       tensor_quant_params_cache=self._tensor_quant_params_cache))
    return self.model_quant_results
