"""Registry of quantization algorithms: the plugin interface of the drop-in.

Same surface as ref: algorithm_manager_api.py:153-441 -- register_quantized_op,
get_quantization_func(alg, op, QuantizeMode), get_update_qsv_func,
get_init_qsv_func, check_op_quantization_config, config-check policy hooks --
so code written against the reference's registry works unchanged.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, Optional

from . import qtyping
from .utils import qsv_utils


@dataclasses.dataclass
class QuantizedOperationInfo:
  tfl_op_key: qtyping.TFLOperationName
  init_qsv_func: Callable[..., Any]
  calibration_func: Callable[..., Any]
  materialize_func: Callable[..., Any]
  update_qsv_func: Callable[..., Any] = qsv_utils.moving_average_update


@dataclasses.dataclass
class QuantizationAlgorithmInfo:
  quantization_algorithm: str
  quantized_ops: dict[qtyping.TFLOperationName, QuantizedOperationInfo]


class AlgorithmManagerApi:
  """Holds {algorithm_key: {op: functions}} plus per-algorithm config checks."""

  def __init__(self):
    self._algorithm_registry: dict[str, QuantizationAlgorithmInfo] = {}
    self._config_check_registry: dict[str, Callable[..., None]] = {}
    self._config_check_policy_registry: dict[str, Optional[qtyping.ConfigCheckPolicyDict]] = {}

  # ---- registration -------------------------------------------------------
  def register_op_quant_config_validation_func(self, algorithm_key: str, config_check_func):
    self._config_check_registry[algorithm_key] = config_check_func

  def register_config_check_policy(self, algorithm_key: str, config_check_policy):
    self._config_check_policy_registry[algorithm_key] = config_check_policy

  def register_quantized_op(self, algorithm_key: str, tfl_op_name: qtyping.TFLOperationName,
                            init_qsv_func, calibration_func, materialize_func,
                            update_qsv_func=qsv_utils.moving_average_update):
    info = self._algorithm_registry.setdefault(
        algorithm_key, QuantizationAlgorithmInfo(algorithm_key, {}))
    info.quantized_ops[tfl_op_name] = QuantizedOperationInfo(
        tfl_op_name, init_qsv_func, calibration_func, materialize_func, update_qsv_func)

  # ---- queries --------------------------------------------------------------
  def is_algorithm_registered(self, quantization_algorithm: str) -> bool:
    return quantization_algorithm in self._algorithm_registry

  def is_op_registered(self, quantization_algorithm: str, tfl_op_name) -> bool:
    return (self.is_algorithm_registered(quantization_algorithm)
            and tfl_op_name in self._algorithm_registry[quantization_algorithm].quantized_ops)

  def get_supported_ops(self, alg_key: str) -> list[qtyping.TFLOperationName]:
    if alg_key not in self._algorithm_registry:
      raise ValueError(f"Unregistered algorithm: {alg_key}")
    return list(self._algorithm_registry[alg_key].quantized_ops.keys())

  def _unsupported(self, algorithm_key, tfl_op_name) -> ValueError:
    return ValueError(
        f"Unsupported operation {tfl_op_name} for Algorithm: {algorithm_key}. Supported ops"
        f" for algorithm {algorithm_key}: {self.get_supported_ops(algorithm_key)}")

  def check_op_quantization_config(self, quantization_algorithm: str, tfl_op_name,
                                   op_quantization_config: qtyping.OpQuantizationConfig) -> None:
    if op_quantization_config.skip_checks:
      return
    if not self.is_op_registered(quantization_algorithm, tfl_op_name):
      raise ValueError(
          f"Unsupported operation {tfl_op_name} for Algorithm: {quantization_algorithm}.")
    if quantization_algorithm not in self._config_check_registry:
      raise ValueError(
          f"Config checking function for  algorithm {quantization_algorithm} is not registered."
          " Please use `register_op_quant_config_validation_func` to register the validation"
          " function.")
    self._config_check_registry[quantization_algorithm](
        tfl_op_name, op_quantization_config,
        self._config_check_policy_registry.get(quantization_algorithm))

  def get_quantization_func(self, algorithm_key: str, tfl_op_name, quantize_mode: qtyping.QuantizeMode):
    if not self.is_op_registered(algorithm_key, tfl_op_name):
      if not self.is_algorithm_registered(algorithm_key):
        raise ValueError(f"Unregistered algorithm: {algorithm_key}")
      raise self._unsupported(algorithm_key, tfl_op_name)
    entry = self._algorithm_registry[algorithm_key].quantized_ops[tfl_op_name]
    fn = {qtyping.QuantizeMode.CALIBRATE: entry.calibration_func,
          qtyping.QuantizeMode.MATERIALIZE: entry.materialize_func}.get(quantize_mode)
    if fn is None:
      raise ValueError(
          f"Cannot retrieve appropriate quantization function for {tfl_op_name} for algorithm"
          f" {algorithm_key} under quantization mode {quantize_mode}. Check if the op is"
          " registed in algorithm_manager.")
    return fn

  def get_update_qsv_func(self, algorithm_key: str, tfl_op_name):
    fn = self._algorithm_registry[algorithm_key].quantized_ops[tfl_op_name].update_qsv_func
    if not fn:
      raise self._unsupported(algorithm_key, tfl_op_name)
    return fn

  def get_init_qsv_func(self, algorithm_key: str, tfl_op_name):
    if not self.is_op_registered(algorithm_key, tfl_op_name):
      raise self._unsupported(algorithm_key, tfl_op_name)
    return self._algorithm_registry[algorithm_key].quantized_ops[tfl_op_name].init_qsv_func
