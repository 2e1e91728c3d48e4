"""Which (op, quantization config) pairs a recipe may select (ref: default_policy.py:25-423).

The reference unrolls a JSON table into lists of OpQuantizationConfig objects and tests
membership; the same acceptance set is stated here as rules: a config family (what the
activation / weight tensor configs look like) -> the ops that support it. `is_supported`
ignores `min_weight_elements` and `algorithm_params` exactly like the reference's check
(algorithms/utils/common_utils.py:104-164); every other field must have the family's value.
"""
from __future__ import annotations

import collections
import copy
import dataclasses
import json
from typing import Any, Optional

from . import qtyping

_Op = qtyping.TFLOperationName
_G = qtyping.QuantGranularity
_P = qtyping.ComputePrecision

_PER_TENSOR_OR_CHANNEL = frozenset({_G.CHANNELWISE, _G.TENSORWISE})
_BLOCKWISE = frozenset({_G.BLOCKWISE_32, _G.BLOCKWISE_64, _G.BLOCKWISE_128, _G.BLOCKWISE_256})


def _ops(*names: str) -> frozenset:
  return frozenset(_Op[n] for n in names)


_MATMUL_LIKE = _ops("BATCH_MATMUL", "CONV_2D", "CONV_2D_TRANSPOSE", "DEPTHWISE_CONV_2D",
                    "EMBEDDING_LOOKUP", "FULLY_CONNECTED")
_STATIC_A16W8 = _ops(
    "ADD", "AVERAGE_POOL_2D", "BATCH_MATMUL", "BROADCAST_TO", "CONCATENATION", "CONV_2D",
    "CONV_2D_TRANSPOSE", "DEPTHWISE_CONV_2D", "DIV", "DYNAMIC_UPDATE_SLICE", "EQUAL",
    "FULLY_CONNECTED", "GATHER", "GATHER_ND", "GELU", "INPUT", "LOGISTIC", "MAXIMUM", "MAX_POOL_2D",
    "MEAN", "MIRROR_PAD", "MUL", "NOT_EQUAL", "OUTPUT", "PACK", "PAD", "PADV2", "REDUCE_MIN", "RELU",
    "RESHAPE", "RESIZE_BILINEAR", "RESIZE_NEAREST_NEIGHBOR", "RSQRT", "SELECT", "SELECT_V2", "SLICE",
    "SOFTMAX", "SPLIT", "SQRT", "STABLEHLO_COMPOSITE", "STRIDED_SLICE", "SUB", "SUM", "TANH",
    "TRANSPOSE", "UNPACK")
_STATIC_A8W8 = _STATIC_A16W8 | _ops("HARD_SWISH", "SPACE_TO_DEPTH", "SQUARED_DIFFERENCE")
_STATIC_W4 = _ops("FULLY_CONNECTED", "CONV_2D", "INPUT", "OUTPUT")


@dataclasses.dataclass(frozen=True)
class _Family:
  """One row of the policy: the shape of a config and the ops that accept it."""
  name: str
  weight_bits: int
  weight_symmetric: frozenset
  weight_granularity: frozenset
  activation_bits: Optional[int]          # None: no activation tensor config
  activation_symmetric: frozenset
  compute_precision: qtyping.ComputePrecision
  explicit_dequantize: bool
  ops: frozenset


_T, _F = frozenset({True}), frozenset({True, False})
FAMILIES = (
    _Family("dynamic_wi8_afp32", 8, _T, _PER_TENSOR_OR_CHANNEL, None, _T, _P.INTEGER, False, _MATMUL_LIKE),
    _Family("dynamic_wi4_afp32", 4, _T, _PER_TENSOR_OR_CHANNEL, None, _T, _P.INTEGER, False,
            _ops("FULLY_CONNECTED", "EMBEDDING_LOOKUP", "CONV_2D")),
    _Family("dynamic_wi4_afp32_blockwise", 4, _T, _BLOCKWISE, None, _T, _P.INTEGER, False,
            _ops("EMBEDDING_LOOKUP", "FULLY_CONNECTED")),
    _Family("dynamic_wi2_afp32", 2, _T, _PER_TENSOR_OR_CHANNEL, None, _T, _P.INTEGER, False,
            _ops("FULLY_CONNECTED", "EMBEDDING_LOOKUP", "CONV_2D")),
    _Family("dynamic_wi2_afp32_blockwise", 2, _T, _BLOCKWISE, None, _T, _P.INTEGER, False,
            _ops("FULLY_CONNECTED")),
    _Family("static_wi8_ai16", 8, _T, _PER_TENSOR_OR_CHANNEL, 16, _T, _P.INTEGER, False, _STATIC_A16W8),
    _Family("static_wi4_ai16", 4, _T, _PER_TENSOR_OR_CHANNEL, 16, _T, _P.INTEGER, False, _STATIC_W4),
    _Family("static_wi8_ai8", 8, _T, _PER_TENSOR_OR_CHANNEL, 8, _F, _P.INTEGER, False, _STATIC_A8W8),
    _Family("static_wi4_ai8", 4, _T, _PER_TENSOR_OR_CHANNEL, 8, _F, _P.INTEGER, False, _STATIC_W4),
    _Family("weightonly_wi8_afp32", 8, _F, _PER_TENSOR_OR_CHANNEL, None, _T, _P.FLOAT, True, _MATMUL_LIKE),
    _Family("weightonly_wi4_afp32", 4, _F, _PER_TENSOR_OR_CHANNEL, None, _T, _P.FLOAT, True,
            _ops("BATCH_MATMUL", "FULLY_CONNECTED", "EMBEDDING_LOOKUP", "CONV_2D")),
)

QUANTIZABLE_COMPOSITES = ("odml.npu_call", "odml.rms_norm", "odml.l2_norm")


def _plain_tensor_config(cfg: qtyping.TensorQuantizationConfig) -> bool:
  """Fields outside the policy's vocabulary must keep their defaults."""
  return cfg.dtype == qtyping.TensorDataType.INT


def _matches(f: _Family, c: qtyping.OpQuantizationConfig) -> bool:
  w, a = c.weight_tensor_config, c.activation_tensor_config
  if w is None or not _plain_tensor_config(w):
    return False
  if (w.num_bits, w.symmetric in f.weight_symmetric, w.granularity in f.weight_granularity) != (
      f.weight_bits, True, True):
    return False
  if (a is None) != (f.activation_bits is None):
    return False
  if a is not None and not (_plain_tensor_config(a) and a.num_bits == f.activation_bits
                            and a.symmetric in f.activation_symmetric
                            and a.granularity == _G.TENSORWISE):
    return False
  return (c.compute_precision == f.compute_precision and c.explicit_dequantize == f.explicit_dequantize
          and not c.skip_checks)


class ConfigCheckPolicy:
  """Duck-typed stand-in for the reference's `ConfigCheckPolicyDict`: `op in policy` and
  `policy.accepts(op, config)`."""

  def __contains__(self, op_name) -> bool:
    return any(op_name in f.ops for f in FAMILIES)

  def keys(self):
    return sorted({op for f in FAMILIES for op in f.ops}, key=lambda o: o.value)

  def accepts(self, op_name, op_quant_config: qtyping.OpQuantizationConfig) -> bool:
    return any(op_name in f.ops and _matches(f, op_quant_config) for f in FAMILIES)


DEFAULT_CONFIG_CHECK_POLICY = ConfigCheckPolicy()


class ListedConfigPolicy(collections.OrderedDict):
  """A policy given as data: op -> the list of accepted configs (what a user's policy .json
  unrolls to; the reference's `ConfigCheckPolicyDict`). A config is accepted when, with
  `min_weight_elements` and the tensors' `algorithm_params` neutralised, it equals one of the
  listed ones (ref algorithms/utils/common_utils.py:130-151)."""

  def accepts(self, op_name, op_quant_config: qtyping.OpQuantizationConfig) -> bool:
    probe = dataclasses.replace(op_quant_config, min_weight_elements=0)
    for side in ("weight_tensor_config", "activation_tensor_config"):
      cfg = getattr(probe, side)
      if cfg is not None:
        probe = dataclasses.replace(probe, **{side: dataclasses.replace(cfg, algorithm_params={})})
    return probe in self[op_name]


def _unroll_json_config(entry: dict) -> list[qtyping.OpQuantizationConfig]:
  """One policy entry lists alternatives for `symmetric` and `granularity` per tensor: every
  combination becomes a config (ref :339-399)."""
  def tensor_configs(spec):
    return [qtyping.TensorQuantizationConfig.from_dict(dict(
        num_bits=spec["num_bits"], symmetric=sym, granularity=gran, dtype=spec["dtype"]))
            for sym in spec["symmetric"] for gran in spec["granularity"]]
  acts = tensor_configs(entry["activation_tensor_config"]) if "activation_tensor_config" in entry else []
  out, seen_weights = [], []
  for w in tensor_configs(entry["weight_tensor_config"]):
    seen_weights.append(w)
    # (without activation configs the reference re-lists every weight config seen so far: the
    # duplicates do not change what the policy accepts, but the order is kept the same)
    for act, weight in ([(a, w) for a in acts] if acts else [(None, sw) for sw in seen_weights]):
      out.append(qtyping.OpQuantizationConfig(
          activation_tensor_config=act, weight_tensor_config=weight,
          compute_precision=entry["compute_precision"], explicit_dequantize=entry["explicit_dequantize"]))
  return out


def update_default_config_policy(raw_json_policy: str) -> ListedConfigPolicy:
  """A policy .json ({"configs": {name: entry}, "ops_per_config": {name: [ops]}}) as a
  ListedConfigPolicy (ref :402-420)."""
  content = json.loads(raw_json_policy)
  policy = ListedConfigPolicy()
  for name, ops in content["ops_per_config"].items():
    unrolled = _unroll_json_config(content["configs"][name])
    for op in ops:
      op_name = qtyping.TFLOperationName(op)
      policy[op_name] = copy.deepcopy(unrolled) + (policy[op_name] if op_name in policy else [])
  return policy


def check_if_valid_op_config(op_name, op_quant_config: qtyping.OpQuantizationConfig,
                             config_check_policy: Optional[ConfigCheckPolicy]) -> None:
  """Raises the reference's ValueError when the policy rejects the pair
  (ref algorithms/utils/common_utils.py:104-164)."""
  if config_check_policy is None:
    why = "No policy was specified at all."
  elif op_name not in config_check_policy:
    why = f"No policy was specified for op: {op_name} with config: {op_quant_config}."
  elif not config_check_policy.accepts(op_name, op_quant_config):
    why = (f"Quantization config for op: {op_name} with config: {op_quant_config!r} was not found"
           " in the policy.")
  else:
    return
  raise ValueError(f"Unsupported op for {op_quant_config.compute_precision}: {op_name}. Error: {why}")


def is_non_quantizable_composite_op(op: Any) -> bool:
  """A STABLEHLO_COMPOSITE op is only quantized when its composite name is on the allow list
  (ref :377-399)."""
  opts = getattr(op, "builtinOptions2", None)
  if isinstance(opts, qtyping.StableHLOCompositeOptionsT) and opts.name is not None:
    name = opts.name.decode("utf-8") if isinstance(opts.name, (bytes, bytearray)) else str(opts.name)
    return name not in QUANTIZABLE_COMPOSITES
  return False
