"""Recipe factories with the reference's names (ref: recipe.py:58-300)."""
from __future__ import annotations

from . import qtyping
from . import recipe_manager
from .algorithm_manager import AlgorithmName

_Op = qtyping.TFLOperationName
_G = qtyping.QuantGranularity


def _dynamic_wix_afp32(num_bits: int, regex: str = ".*", operation_name=_Op.ALL_SUPPORTED, **kwargs):
  rm = recipe_manager.RecipeManager()
  rm.add_dynamic_config(regex=regex, operation_name=operation_name, num_bits=num_bits, **kwargs)
  return rm.get_quantization_recipe()


def _weight_only_wix_afp32(num_bits: int, regex: str = ".*", operation_name=_Op.ALL_SUPPORTED, **kwargs):
  rm = recipe_manager.RecipeManager()
  rm.add_weight_only_config(regex=regex, operation_name=operation_name, num_bits=num_bits, **kwargs)
  return rm.get_quantization_recipe()


def dynamic_wi8_afp32(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  return _dynamic_wix_afp32(8, algorithm_key=algorithm_key)


def dynamic_wi4_afp32(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  return _dynamic_wix_afp32(4, algorithm_key=algorithm_key)


def weight_only_wi8_afp32(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  return _weight_only_wix_afp32(8, algorithm_key=algorithm_key)


def weight_only_wi4_afp32(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  return _weight_only_wix_afp32(4, algorithm_key=algorithm_key)


def static_wi8_ai8(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  rm = recipe_manager.RecipeManager()
  rm.add_static_config(regex=".*", operation_name=_Op.ALL_SUPPORTED, activation_num_bits=8,
                       weight_num_bits=8, algorithm_key=algorithm_key)
  return rm.get_quantization_recipe()


def static_wi8_ai16(algorithm_key=AlgorithmName.MIN_MAX_UNIFORM_QUANT):
  rm = recipe_manager.RecipeManager()
  rm.add_static_config(regex=".*", operation_name=_Op.ALL_SUPPORTED, activation_num_bits=16,
                       weight_num_bits=8, algorithm_key=algorithm_key)
  return rm.get_quantization_recipe()


dynamic_wi8c_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=8, **kw)  # noqa: E731
dynamic_wi4c_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=4, **kw)  # noqa: E731
dynamic_wi2c_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=2, **kw)  # noqa: E731
dynamic_wi8b32_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=8, granularity=_G.BLOCKWISE_32, **kw)  # noqa: E731
dynamic_wi4b32_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=4, granularity=_G.BLOCKWISE_32, **kw)  # noqa: E731
dynamic_wi2b32_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=2, granularity=_G.BLOCKWISE_32, **kw)  # noqa: E731
dynamic_wi8b64_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=8, granularity=_G.BLOCKWISE_64, **kw)  # noqa: E731
dynamic_wi4b64_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=4, granularity=_G.BLOCKWISE_64, **kw)  # noqa: E731
dynamic_wi4b128_afp32 = lambda **kw: _dynamic_wix_afp32(num_bits=4, granularity=_G.BLOCKWISE_128, **kw)  # noqa: E731
dynamic_wi8c_hr_afp32 = lambda **kw: dynamic_wi8c_afp32(algorithm_key=AlgorithmName.DECOMPOSED_HADAMARD_ROTATION, **kw)  # noqa: E731
dynamic_wi4c_hr_afp32 = lambda **kw: dynamic_wi4c_afp32(algorithm_key=AlgorithmName.DECOMPOSED_HADAMARD_ROTATION, **kw)  # noqa: E731


def dynamic_legacy_wi8_afp32():
  """dynamic_wi8_afp32 for models first quantized with the older TFLite tooling: only weights of
  at least 1024 elements are quantized (ref :43-78)."""
  return [dict(regex=".*", operation="*", algorithm_key="min_max_uniform_quantize", op_config=dict(
      weight_tensor_config=dict(num_bits=8, symmetric=True, granularity="CHANNELWISE", dtype="INT",
                                block_size=0),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False,
      min_weight_elements=1024))]
