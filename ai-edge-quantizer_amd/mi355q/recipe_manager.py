"""Recipe -> per-op quantization config resolution.

Same data model and lookup rule as ref: recipe_manager.py:40-262: a recipe is a
list of {regex, operation, algorithm_key, op_config}; scopes are matched with
re.search in insertion order and the LAST valid match wins; '*' covers every op.
"""
from __future__ import annotations

import collections
import dataclasses
import re
from typing import Optional

from . import algorithm_manager
from . import qtyping

_Op = qtyping.TFLOperationName
_Cfg = qtyping.OpQuantizationConfig
_TCfg = qtyping.TensorQuantizationConfig
AlgorithmName = algorithm_manager.AlgorithmName


@dataclasses.dataclass
class OpQuantizationRecipe:
  regex: str
  operation: qtyping.TFLOperationName
  algorithm_key: str
  op_config: qtyping.OpQuantizationConfig = dataclasses.field(default_factory=_Cfg)


class RecipeManager:
  def __init__(self):
    self._scope_configs: collections.OrderedDict[str, list[OpQuantizationRecipe]] = (
        collections.OrderedDict())

  def add_quantization_config(self, regex: str, operation_name, op_config: Optional[_Cfg] = None,
                              algorithm_key: str = AlgorithmName.MIN_MAX_UNIFORM_QUANT) -> None:
    """ref :85-153."""
    try:
      AlgorithmName(algorithm_key)
    except ValueError as e:
      raise ValueError(f"Unsupported algorithm key: {algorithm_key}.") from e
    operation_name = _Op(operation_name)
    entry = OpQuantizationRecipe(regex, operation_name, algorithm_key, op_config or _Cfg())
    if operation_name == _Op.ALL_SUPPORTED:
      self._scope_configs[regex] = [entry]
      return
    if algorithm_key != AlgorithmName.NO_QUANTIZE:
      algorithm_manager.check_op_quantization_config(algorithm_key, operation_name, entry.op_config)
    current = self._scope_configs.setdefault(regex, [])
    for i, old in enumerate(current):
      if old.operation == operation_name:
        current[i] = entry
        break
    else:
      current.append(entry)

  def add_dynamic_config(self, regex: str, operation_name, num_bits: int,
                         granularity=qtyping.QuantGranularity.CHANNELWISE,
                         algorithm_key: str = AlgorithmName.MIN_MAX_UNIFORM_QUANT) -> None:
    """Integer weights, float activations quantized at run time (DRQ)."""
    w = _TCfg(num_bits=num_bits, symmetric=True, granularity=granularity)
    self.add_quantization_config(
        regex, operation_name,
        _Cfg(weight_tensor_config=w, compute_precision=qtyping.ComputePrecision.INTEGER),
        algorithm_key)

  def add_weight_only_config(self, regex: str, operation_name, num_bits: int,
                             granularity=qtyping.QuantGranularity.CHANNELWISE,
                             algorithm_key: str = AlgorithmName.MIN_MAX_UNIFORM_QUANT) -> None:
    # integer weights, except float_casting's FP16 (ref :335-341)
    dtype = (qtyping.TensorDataType.FLOAT if algorithm_key == AlgorithmName.FLOAT_CASTING
             else qtyping.TensorDataType.INT)
    w = _TCfg(num_bits=num_bits, symmetric=True, granularity=granularity, dtype=dtype)
    self.add_quantization_config(
        regex, operation_name,
        _Cfg(weight_tensor_config=w, compute_precision=qtyping.ComputePrecision.FLOAT,
             explicit_dequantize=True), algorithm_key)

  def add_static_config(self, regex: str, operation_name, activation_num_bits: int,
                        weight_num_bits: int,
                        weight_granularity=qtyping.QuantGranularity.CHANNELWISE,
                        algorithm_key: str = AlgorithmName.MIN_MAX_UNIFORM_QUANT) -> None:
    act = _TCfg(num_bits=activation_num_bits, symmetric=activation_num_bits == 16,
                granularity=qtyping.QuantGranularity.TENSORWISE)
    w = _TCfg(num_bits=weight_num_bits, symmetric=True, granularity=weight_granularity)
    self.add_quantization_config(
        regex, operation_name,
        _Cfg(activation_tensor_config=act, weight_tensor_config=w,
             compute_precision=qtyping.ComputePrecision.INTEGER), algorithm_key)

  def get_quantization_configs(self, target_op_name, scope_name: str):
    """(algorithm_key, op_config) of the last valid matching entry (ref :157-202)."""
    key, cfg = AlgorithmName.NO_QUANTIZE, _Cfg()
    for scope_regex, entries in self._scope_configs.items():
      if not re.search(scope_regex, scope_name):
        continue
      for entry in entries:
        if entry.operation not in (_Op.ALL_SUPPORTED, target_op_name):
          continue
        if entry.algorithm_key != AlgorithmName.NO_QUANTIZE:
          try:
            algorithm_manager.check_op_quantization_config(entry.algorithm_key, target_op_name,
                                                           entry.op_config)
          except ValueError:
            continue
        key, cfg = entry.algorithm_key, entry.op_config
    return key, cfg

  def get_quantization_recipe(self) -> qtyping.ModelQuantizationRecipe:
    out = []
    for entries in self._scope_configs.values():
      for e in entries:
        out.append({"regex": e.regex, "operation": str(e.operation.value),
                    "algorithm_key": str(getattr(e.algorithm_key, "value", e.algorithm_key)),
                    "op_config": e.op_config.to_dict()})
    return out

  def load_quantization_recipe(self, quantization_recipe: qtyping.ModelQuantizationRecipe) -> None:
    self._scope_configs = collections.OrderedDict()
    for c in quantization_recipe:
      no_q = c["algorithm_key"] == AlgorithmName.NO_QUANTIZE
      self.add_quantization_config(c["regex"], c["operation"],
                                   None if no_q else _Cfg.from_dict(c["op_config"]),
                                   c["algorithm_key"])

  def need_calibration(self) -> bool:
    """SRQ and GPTQ need activation statistics (ref :250-262)."""
    for entries in self._scope_configs.values():
      for e in entries:
        if (e.op_config.compute_precision == qtyping.ComputePrecision.INTEGER
            and e.op_config.activation_tensor_config is not None):
          return True
        if e.algorithm_key == AlgorithmName.GPTQ:
          return True
    return False
