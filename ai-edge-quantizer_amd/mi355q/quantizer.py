"""Quantizer facade: `.tflite` in, quantized `.tflite` out (ref: quantizer.py:59-620).

Covers the weight-requantization flow of the hot path: read the model (zero-copy views of the
mmap'd file), load a recipe, generate parameters through the registry (GPU kernels), apply the
QUANTIZE_TENSOR transformation (pack + store + metadata) and serialize. Calibration by running
the float model (LiteRT interpreter) and model validation are outside this build's scope;
calibration results (QSVs) are passed in.
"""
from __future__ import annotations

import contextlib
import dataclasses
import json
import os
import pathlib
from typing import Any, Optional, Union

from . import algorithm_manager
from . import calibrator
from . import default_policy
from . import model_modifier
from . import params_generator
from . import qtyping
from . import recipe_manager
from . import requant_queue
from .utils import tfl_flatbuffer_utils
from .utils import tflite_flatbuffer

apply_quantize_tensor_transformations = model_modifier.apply_quantize_tensor_transformations

Path = Union[str, pathlib.Path]


class _Flag(int):
  """A bool-valued int that can also be called (`qt.need_calibration` / `qt.need_calibration()`)."""

  def __call__(self) -> bool:
    return bool(self)

  def __repr__(self) -> str:
    return repr(bool(self))


@dataclasses.dataclass(frozen=True)
class QuantizationResult:
  """recipe + the serialized quantized model (ref :59-128)."""
  recipe: qtyping.ModelQuantizationRecipe
  quantized_model: Optional[Any]

  def save(self, save_folder: Path, model_name: str, overwrite: bool = False) -> None:
    os.makedirs(save_folder, exist_ok=True)
    self.export_model(str(pathlib.Path(save_folder) / f"{model_name}.tflite"), overwrite)
    recipe_path = pathlib.Path(save_folder) / (model_name + "_recipe.json")
    tfl_flatbuffer_utils.set_file_contents(recipe_path, json.dumps(self.recipe).encode())

  def export_model(self, filepath: Path, overwrite: bool = False) -> None:
    if self.quantized_model is None:
      raise RuntimeError("No quantized model to save. Make sure .quantize() is called.")
    if os.path.exists(filepath) and not overwrite:
      raise ValueError(
          f"The model {filepath} already exists in the folder. Please consider change the model"
          " name or specify overwrite=True to overwrite the model if needed.")
    tfl_flatbuffer_utils.set_file_contents(filepath, self.quantized_model)


class Quantizer:
  """`float_model`: a `.tflite` path, model bytes, or an already parsed ModelT tree."""

  def __init__(self, float_model: Any,
               quantization_recipe: Optional[Union[Path, qtyping.ModelQuantizationRecipe]] = None):
    if isinstance(float_model, (str, pathlib.Path)):
      self._float_model_buffer = tfl_flatbuffer_utils.get_model_content(float_model)
      self.float_model = tfl_flatbuffer_utils.read_model(self._float_model_buffer)
    elif isinstance(float_model, (bytes, bytearray, memoryview)):
      self._float_model_buffer = memoryview(float_model)
      self.float_model = tfl_flatbuffer_utils.read_model(self._float_model_buffer)
    elif isinstance(float_model, tflite_flatbuffer.TableT):
      self._float_model_buffer = None
      self.float_model = float_model
    else:
      raise ValueError("Unsupported float_model type: %s" % type(float_model).__name__)
    self._recipe_manager = recipe_manager.RecipeManager()
    self._result = QuantizationResult([{}], None)
    self.quantized_model_object: Optional[Any] = None
    if quantization_recipe is not None:
      self.load_quantization_recipe(quantization_recipe)

  def load_quantization_recipe(self, recipe: Union[Path, qtyping.ModelQuantizationRecipe]) -> None:
    """A recipe list, or the path of a recipe .json (ref :195-205)."""
    if isinstance(recipe, (str, pathlib.Path)):
      with open(recipe, "r", encoding="utf-8") as f:
        recipe = json.load(f)
    self._recipe_manager.load_quantization_recipe(recipe)

  def get_quantization_recipe(self) -> qtyping.ModelQuantizationRecipe:
    return self._recipe_manager.get_quantization_recipe()

  @property
  def need_calibration(self) -> "_Flag":
    """A property in the reference (`qt.need_calibration`); also callable here."""
    return _Flag(self._recipe_manager.need_calibration())

  def load_config_policy(self, filename: Path) -> None:
    """A user policy .json replaces the min/max algorithm's config check policy (ref :207-222)."""
    with open(filename, "r", encoding="utf-8") as f:
      policy = default_policy.update_default_config_policy(f.read())
    algorithm_manager.register_config_check_policy_func(
        algorithm_manager.AlgorithmName.MIN_MAX_UNIFORM_QUANT, policy)

  def update_quantization_recipe(self, regex: str, operation_name, op_config=None,
                                 algorithm_key: str = algorithm_manager.AlgorithmName.MIN_MAX_UNIFORM_QUANT):
    """ref :233-262."""
    self._recipe_manager.add_quantization_config(regex, operation_name, op_config, algorithm_key)

  def add_dynamic_config(self, regex: str, operation_name, num_bits: int,
                         granularity=qtyping.QuantGranularity.CHANNELWISE,
                         algorithm_key: str = algorithm_manager.AlgorithmName.MIN_MAX_UNIFORM_QUANT):
    """ref :264-289."""
    self._recipe_manager.add_dynamic_config(regex, operation_name, num_bits, granularity, algorithm_key)

  def add_weight_only_config(self, regex: str, operation_name, num_bits: int,
                             granularity=qtyping.QuantGranularity.CHANNELWISE,
                             algorithm_key: str = algorithm_manager.AlgorithmName.MIN_MAX_UNIFORM_QUANT):
    """ref :291-316."""
    self._recipe_manager.add_weight_only_config(regex, operation_name, num_bits, granularity, algorithm_key)

  def add_static_config(self, regex: str, operation_name, activation_num_bits: int, weight_num_bits: int,
                        weight_granularity=qtyping.QuantGranularity.CHANNELWISE,
                        algorithm_key: str = algorithm_manager.AlgorithmName.MIN_MAX_UNIFORM_QUANT):
    """ref :318-352."""
    self._recipe_manager.add_static_config(regex, operation_name, activation_num_bits, weight_num_bits,
                                           weight_granularity, algorithm_key)

  def calibrate(self, calibration_data: dict, previous_calibration_result: Optional[dict] = None,
                tensor_provider: Optional[Any] = None, hessians: str = "consumed") -> dict[str, qtyping.QSV]:
    """Model QSVs from per-sample tensor contents (ref :369-413). The reference runs the float
    model in the LiteRT interpreter to obtain those tensors; here each sample is the
    {tensor name: ndarray} map itself, or `tensor_provider(signature_key, sample)` returns it.
    `hessians`: see Calibrator ("all" = a GPTQ Hessian for every runtime tensor, as the reference)."""
    if not self.need_calibration:
      return {}
    calib = calibrator.Calibrator(self.float_model, tensor_provider=tensor_provider, hessians=hessians)
    if previous_calibration_result is not None:
      calib.load_model_qsvs(previous_calibration_result)
    calib.calibrate(calibration_data, self._recipe_manager)
    return calib.get_model_qsvs()

  def quantize(self, calibration_result: Optional[dict[str, qtyping.QSV]] = None,
               serialize_to_path: Optional[Path] = None) -> QuantizationResult:
    """The float model is left untouched; the result holds the serialized quantized model
    (also written to `serialize_to_path` when given). `quantized_model_object` keeps the
    quantized ModelT tree for inspection."""
    if not self.get_quantization_recipe():
      raise RuntimeError("Can not quantize without a quantization recipe.")
    generator = params_generator.ParamsGenerator(self.float_model)
    modifier = model_modifier.ModelModifier(self.float_model)
    # One block around the op walk AND the writer when a file is written: the walk's own block joins it, so results
    # stay placeholders until somebody reads them; the writer lays the file out from their sizes, sends every quantized
    # buffer on its way behind its own producer and takes the values the flatbuffer stores (per-channel scales) last
    # (model_modifier.serialize_model, utils/tflite_flatbuffer.serialize_with_external_buffers).
    from . import runtime as rt
    rt.mark("quantize: call")
    try:
      with (requant_queue.batching() if serialize_to_path else contextlib.nullcontext()) as block:
        params = generator.generate_quantization_parameters(self._recipe_manager, calibration_result)
        rt.mark("quantize: parameters generated (host)")
        serialized = modifier.modify_model(params, serialize_to_path=serialize_to_path)
    finally:
      if serialize_to_path:
        rt.release_upload_files()      # (also when the call failed: announced uploads are waited for and dropped)
    rt.mark("quantize: model modified and serialized (host)")
    # launches / tensors of the batched path (with a file written, some launches left from inside the writer)
    self.batch_stats = dict(block.stats) if block is not None else getattr(generator, "batch_stats", None)
    self.quantized_model_object = modifier.quantized_model_object
    self._result = QuantizationResult(self.get_quantization_recipe(), serialized)
    return self._result
