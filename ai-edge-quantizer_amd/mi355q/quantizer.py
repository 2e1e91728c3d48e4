"""Quantizer facade for in-memory models (ref: quantizer.py:131-620).

Covers the weight-requantization flow of the hot path: load a recipe, generate
parameters through the registry, and apply the QUANTIZE_TENSOR transformation
(pack + store + metadata). Transformations that need graph surgery (Q/DQ insertion,
Hadamard op insertion) and .tflite (de)serialization are outside this build's scope
 and raise NotImplementedError.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

from . import params_generator
from . import qtyping
from . import recipe_manager
from .transformations import quantize_tensor
from .transformations import transformation_utils
from .utils import tfl_flatbuffer_utils

_T = qtyping.QuantTransformation


@dataclasses.dataclass(frozen=True)
class QuantizationResult:
  recipe: qtyping.ModelQuantizationRecipe
  quantized_model: Optional[Any]


class Quantizer:
  def __init__(self, float_model: Any, quantization_recipe: Optional[qtyping.ModelQuantizationRecipe] = None):
    self.float_model = float_model
    self._recipe_manager = recipe_manager.RecipeManager()
    self._result = QuantizationResult([{}], None)
    if quantization_recipe is not None:
      self.load_quantization_recipe(quantization_recipe)

  def load_quantization_recipe(self, recipe: qtyping.ModelQuantizationRecipe) -> None:
    self._recipe_manager.load_quantization_recipe(recipe)

  def get_quantization_recipe(self) -> qtyping.ModelQuantizationRecipe:
    return self._recipe_manager.get_quantization_recipe()

  def need_calibration(self) -> bool:
    return self._recipe_manager.need_calibration()

  def quantize(self, calibration_result: Optional[dict[str, qtyping.QSV]] = None) -> QuantizationResult:
    """Quantizes the model IN PLACE (buffers, tensor types, quantization records)."""
    if not self.get_quantization_recipe():
      raise RuntimeError("Can not quantize without a quantization recipe.")
    params = params_generator.ParamsGenerator(self.float_model).generate_quantization_parameters(
        self._recipe_manager, calibration_result)
    apply_quantize_tensor_transformations(self.float_model, params)
    self._result = QuantizationResult(self.get_quantization_recipe(), self.float_model)
    return self._result


def apply_quantize_tensor_transformations(model: Any, params: dict[str, qtyping.TensorTransformationParams]) -> None:
  """QUANTIZE_TENSOR for every constant whose consumers all ask for it with equal
  parameters (the case the reference's instruction generator leaves as a single
  QUANTIZE_TENSOR instruction)."""
  buffer_origin: dict[int, Any] = {}
  for sg in model.subgraphs:
    n_before = len(sg.tensors)
    for tid in range(n_before):
      tensor = sg.tensors[tid]
      p = params.get(tfl_flatbuffer_utils.get_tensor_name(tensor))
      if p is None:
        continue
      links = list(p.consumers or []) + ([p.producer] if p.producer is not None else [])
      wanted = {t for link in links for t in link.transformations}
      if wanted <= {_T.NO_QUANTIZE}:
        continue
      if wanted != {_T.QUANTIZE_TENSOR}:
        raise NotImplementedError(
            f"tensor {p.tensor_name}: transformations {sorted(t.name for t in wanted)} need graph"
            " rewriting, which is outside this build's scope")
      first = links[0].parameters
      if any(link.parameters != first for link in links[1:]):
        raise NotImplementedError(f"tensor {p.tensor_name}: consumers disagree on parameters")
      quantize_tensor.quantize_tensor(transformation_utils.TransformationInput(
          tensor_id=tid, model=model, subgraph=sg, producer=-1, consumers=[], quant_params=first,
          buffer_origin=buffer_origin))
