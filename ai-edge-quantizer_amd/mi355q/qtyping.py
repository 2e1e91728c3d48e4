"""Value types that cross the drop-in boundary.

API-compatible with the reference's `ai_edge_quantizer.qtyping` for the
calibration / requantization path (ref: qtyping.py:82-710): same class, field and
enum member names, same defaults, same value-based equality for
`UniformQuantParams` (params_generator relies on it for buffer-sharing checks,
ref: params_generator.py:516-560). Schema types come from `mi355q.schema` instead
of ai_edge_litert.
"""
from __future__ import annotations

import copy
import dataclasses
import enum
from collections import OrderedDict
from collections.abc import Mapping, MutableMapping
from typing import Any, Callable, Optional, Union

import numpy as np

from . import schema

QSV = MutableMapping[str, Any]
ModelQuantizationRecipe = list[dict[str, Any]]

# Schema re-exports (ref: qtyping.py:37-79).
TensorType = schema.TensorType
TensorT = schema.TensorT
OperatorT = schema.OperatorT
BufferT = schema.BufferT
SubGraphT = schema.SubGraphT
ModelT = schema.ModelT
OperatorCodeT = schema.OperatorCodeT
BuiltinOperator = schema.BuiltinOperator
QuantizationDetails = schema.QuantizationDetails
QuantizationParametersT = schema.QuantizationParametersT
BlockwiseQuantizationT = schema.BlockwiseQuantizationT
FullyConnectedOptionsT = schema.FullyConnectedOptionsT
BatchMatMulOptionsT = schema.BatchMatMulOptionsT
MulOptionsT = schema.MulOptionsT
ActivationFunctionType = schema.ActivationFunctionType
StableHLOCompositeOptionsT = schema.StableHLOCompositeOptionsT
SignatureDefT = schema.SignatureDefT
TensorMapT = schema.TensorMapT
MetadataT = schema.MetadataT
BuiltinOptions = schema.BuiltinOptions
BuiltinOptions2 = schema.BuiltinOptions2


class FrozenMapping(dict):
  """Hashable read-only dict (role of `immutabledict` in the reference)."""

  def __hash__(self):  # type: ignore[override]
    return hash(tuple(sorted((k, repr(v)) for k, v in self.items())))

  def _readonly(self, *_, **__):
    raise TypeError("FrozenMapping is read-only")

  def __deepcopy__(self, memo):
    return FrozenMapping({copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()})

  def __reduce__(self):
    return (FrozenMapping, (dict(self),))

  __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _readonly


class TFLOperationName(str, enum.Enum):
  """ref: qtyping.py:82-136 (same member names / values)."""
  ALL_SUPPORTED = "*"
  INPUT = "INPUT"
  OUTPUT = "OUTPUT"
  FULLY_CONNECTED = "FULLY_CONNECTED"
  BATCH_MATMUL = "BATCH_MATMUL"
  DEPTHWISE_CONV_2D = "DEPTHWISE_CONV_2D"
  CONV_2D = "CONV_2D"
  CONV_2D_TRANSPOSE = "CONV_2D_TRANSPOSE"
  AVERAGE_POOL_2D = "AVERAGE_POOL_2D"
  RESHAPE = "RESHAPE"
  CUSTOM_OP = "CUSTOM_OP"
  EMBEDDING_LOOKUP = "EMBEDDING_LOOKUP"
  SOFTMAX = "SOFTMAX"
  TANH = "TANH"
  TRANSPOSE = "TRANSPOSE"
  GELU = "GELU"
  ADD = "ADD"
  SUB = "SUB"
  MUL = "MUL"
  MEAN = "MEAN"
  RSQRT = "RSQRT"
  CONCATENATION = "CONCATENATION"
  STRIDED_SLICE = "STRIDED_SLICE"
  SPLIT = "SPLIT"
  LOGISTIC = "LOGISTIC"
  SLICE = "SLICE"
  SUM = "SUM"
  SELECT = "SELECT"
  SELECT_V2 = "SELECT_V2"
  DYNAMIC_UPDATE_SLICE = "DYNAMIC_UPDATE_SLICE"
  STABLEHLO_COMPOSITE = "STABLEHLO_COMPOSITE"
  PAD = "PAD"
  SQUARED_DIFFERENCE = "SQUARED_DIFFERENCE"
  MAX_POOL_2D = "MAX_POOL_2D"
  RESIZE_BILINEAR = "RESIZE_BILINEAR"
  RESIZE_NEAREST_NEIGHBOR = "RESIZE_NEAREST_NEIGHBOR"
  GATHER_ND = "GATHER_ND"
  PACK = "PACK"
  UNPACK = "UNPACK"
  DIV = "DIV"
  BROADCAST_TO = "BROADCAST_TO"
  SQRT = "SQRT"
  GATHER = "GATHER"
  HARD_SWISH = "HARD_SWISH"
  MAXIMUM = "MAXIMUM"
  PADV2 = "PADV2"
  REDUCE_MIN = "REDUCE_MIN"
  EQUAL = "EQUAL"
  NOT_EQUAL = "NOT_EQUAL"
  MIRROR_PAD = "MIRROR_PAD"
  SPACE_TO_DEPTH = "SPACE_TO_DEPTH"
  RELU = "RELU"


class QuantizeMode(enum.Enum):
  CALIBRATE = 2
  MATERIALIZE = 3


class OpExecutionMode(str, enum.Enum):
  WEIGHT_ONLY = "WEIGHT_ONLY"
  DRQ = "DRQ"
  SRQ = "SRQ"


class ComputePrecision(str, enum.Enum):
  INTEGER = "INTEGER"
  FLOAT = "FLOAT"


class TensorDataType(str, enum.Enum):
  INT = "INT"
  FLOAT = "FLOAT"


class QuantGranularity(str, enum.Enum):
  TENSORWISE = "TENSORWISE"
  CHANNELWISE = "CHANNELWISE"
  BLOCKWISE_32 = "BLOCKWISE_32"
  BLOCKWISE_64 = "BLOCKWISE_64"
  BLOCKWISE_128 = "BLOCKWISE_128"
  BLOCKWISE_256 = "BLOCKWISE_256"


class QuantTransformation(enum.Enum):
  NO_QUANTIZE = 0
  ADD_QUANTIZE = 1
  ADD_DEQUANTIZE = 2
  QUANTIZE_TENSOR = 3
  EMULATED_SUBCHANNEL = 4
  DUPLICATE_BUFFER = 5
  DUPLICATE_TENSOR = 6
  INSERT_HADAMARD_ROTATION = 7
  INSERT_DECOMPOSED_HADAMARD_ROTATION = 8
  INSERT_MULTIPLY = 9


def _same_array(a: Optional[np.ndarray], b: Optional[np.ndarray]) -> bool:
  if a is None or b is None:
    return a is None and b is None
  if a is b:
    return True
  for x, y in ((a, b), (b, a)):      # a payload whose bytes stayed in another rank's HBM (runtime.RemoteBuffer)
    same = getattr(x, "same_payload", None)
    if same is not None:
      return bool(same(y))
  return np.array_equal(a, b)


def _same_value(a: Any, b: Any) -> bool:
  if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
    return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and np.array_equal(a, b)
  return a == b


@dataclasses.dataclass(frozen=True, eq=False)
class UniformQuantParams:
  """ref: qtyping.py:205-313."""

  class HadamardRotationParams:
    """ref: qtyping.py:227-251."""

    def __init__(self, random_binary_vector: np.ndarray, hadamard_size: int):
      self.random_binary_vector = random_binary_vector
      self.hadamard_size = hadamard_size

    def __eq__(self, other):
      if other.__class__ is not self.__class__:
        return NotImplemented
      return self is other or (
          np.array_equal(self.random_binary_vector, other.random_binary_vector)
          and self.hadamard_size == other.hadamard_size)

    __hash__ = None  # type: ignore[assignment]

  num_bits: int
  quantized_dimension: Optional[int]
  scale: np.ndarray
  zero_point: np.ndarray
  symmetric: bool = True
  quantized_data: Optional[np.ndarray] = None
  block_size: int = 0
  hadamard: Optional[HadamardRotationParams] = None
  custom_algorithm_param: Optional[dict[str, Any]] = None

  @classmethod
  def from_tfl_tensor_details(cls, tensor_detail) -> "UniformQuantParams":
    """ref: qtyping.py:263-296."""
    qp = tensor_detail["quantization_parameters"]
    bits = {np.dtype(np.int8): 8, np.dtype(np.int16): 16, np.dtype(np.int32): 32,
            np.dtype(np.int64): 64}.get(np.dtype(tensor_detail["dtype"]))
    if bits is None:
      raise ValueError(
          f"Unsupported data type: {tensor_detail['dtype']}. Supported types are np.int8,"
          " np.int16, np.int32, np.int64.")
    return cls(quantized_dimension=qp["quantized_dimension"], num_bits=bits,
               scale=qp["scales"], zero_point=qp["zero_points"],
               symmetric=sum(abs(qp["zero_points"])) == 0, block_size=qp["block_size"])

  def __eq__(self, other):
    if other.__class__ is not self.__class__:
      return NotImplemented
    if self is other:
      return True
    a, b = self.custom_algorithm_param, other.custom_algorithm_param
    if (a is None) != (b is None):
      return False
    if a is not None and (a.keys() != b.keys()
                          or not all(_same_value(v, b[k]) for k, v in a.items())):
      return False
    return (self.num_bits == other.num_bits
            and self.quantized_dimension == other.quantized_dimension
            and self.symmetric == other.symmetric
            and self.block_size == other.block_size
            and _same_array(self.scale, other.scale)
            and _same_array(self.zero_point, other.zero_point)
            and _same_array(self.quantized_data, other.quantized_data)
            and self.hadamard == other.hadamard)

  __hash__ = None  # type: ignore[assignment]


@dataclasses.dataclass(frozen=True, eq=False)
class NonLinearQuantParams:
  """ref: qtyping.py:316-339."""
  num_bits: int
  quantized_data: Optional[np.ndarray]
  data_type: TensorDataType = TensorDataType.FLOAT

  def __eq__(self, other):
    if other.__class__ is not self.__class__:
      return NotImplemented
    return self is other or (self.num_bits == other.num_bits
                             and self.data_type == other.data_type
                             and _same_array(self.quantized_data, other.quantized_data))

  __hash__ = None  # type: ignore[assignment]


@dataclasses.dataclass(frozen=True)
class OpToTensorParams:
  subgraph_op_id: int
  transformations: list[QuantTransformation]
  parameters: Union[None, UniformQuantParams, NonLinearQuantParams] = None


@dataclasses.dataclass
class TensorTransformationParams:
  tensor_name: str
  producer: Optional[OpToTensorParams] = None
  consumers: Optional[list[OpToTensorParams]] = None

  def __copy__(self):
    return TensorTransformationParams(
        self.tensor_name, self.producer,
        None if self.consumers is None else list(self.consumers))


_BLOCK_TO_GRANULARITY = {32: QuantGranularity.BLOCKWISE_32, 64: QuantGranularity.BLOCKWISE_64,
                         128: QuantGranularity.BLOCKWISE_128, 256: QuantGranularity.BLOCKWISE_256}


def _plain_dict(obj) -> dict[str, Any]:
  """dataclasses.asdict without None / empty-mapping entries (ref: qtyping.py:415-429)."""
  def factory(items):
    out = {}
    for k, v in items:
      if v is None or (isinstance(v, Mapping) and not v):
        continue
      out[k] = dict(v) if isinstance(v, Mapping) and type(v) is not dict else v
    return out
  return dataclasses.asdict(obj, dict_factory=factory)


@dataclasses.dataclass(frozen=True)
class TensorQuantizationConfig:
  """ref: qtyping.py:384-445."""
  num_bits: int
  symmetric: bool = True
  granularity: QuantGranularity = QuantGranularity.TENSORWISE
  dtype: TensorDataType = TensorDataType.INT
  algorithm_params: Mapping[str, Any] = dataclasses.field(default_factory=FrozenMapping)

  def __post_init__(self):
    if not isinstance(self.algorithm_params, FrozenMapping):
      object.__setattr__(self, "algorithm_params", FrozenMapping(self.algorithm_params))

  def to_dict(self) -> dict[str, Any]:
    return _plain_dict(self)

  @classmethod
  def from_dict(cls, params: dict[str, Any]) -> "TensorQuantizationConfig":
    p = copy.deepcopy(params)
    block = p.pop("block_size", 0)  # legacy recipes (ref: qtyping.py:448-462)
    if block > 0:
      if block not in _BLOCK_TO_GRANULARITY:
        raise ValueError(f"Unsupported block size: {block}")
      p["granularity"] = _BLOCK_TO_GRANULARITY[block]
    known = {f.name for f in dataclasses.fields(cls)}
    algo = dict(p.pop("algorithm_params", {}))
    for key in [k for k in p if k not in known]:
      algo[key] = p.pop(key)
    return cls(algorithm_params=algo, **p)


@dataclasses.dataclass(frozen=True)
class OpQuantizationConfig:
  """ref: qtyping.py:465-551."""
  activation_tensor_config: Optional[TensorQuantizationConfig] = None
  weight_tensor_config: Optional[TensorQuantizationConfig] = None
  compute_precision: ComputePrecision = ComputePrecision.FLOAT
  explicit_dequantize: bool = False
  skip_checks: bool = False
  min_weight_elements: int = 0

  def __post_init__(self):
    act, w = self.activation_tensor_config, self.weight_tensor_config
    if act is None or w is None:
      return
    if act.dtype == TensorDataType.INT and w.dtype == TensorDataType.FLOAT:
      raise ValueError("An op can not be set to have integer activation but float weights!")
    if (act.dtype == TensorDataType.INT and w.dtype == TensorDataType.INT
        and self.compute_precision != ComputePrecision.INTEGER):
      raise ValueError("Op execution mode must be SRQ (static range quantization) if both"
                       " activation and weight tensors are quantized!")

  def to_dict(self) -> dict[str, Any]:
    return _plain_dict(self)

  @classmethod
  def from_dict(cls, params: dict[str, Any]) -> "OpQuantizationConfig":
    p = copy.deepcopy(params)
    p["weight_tensor_config"] = TensorQuantizationConfig.from_dict(p["weight_tensor_config"])
    if "activation_tensor_config" in p:
      p["activation_tensor_config"] = TensorQuantizationConfig.from_dict(
          p["activation_tensor_config"])
    return cls(**p)


@dataclasses.dataclass(frozen=True)
class GraphInfo:
  subgraph_tensors: list[Any]
  buffers: list[Any]


@dataclasses.dataclass(frozen=True)
class OpInfo:
  op: Any
  op_name: TFLOperationName
  subgraph_op_index: int
  op_quant_config: OpQuantizationConfig


@dataclasses.dataclass
class TransformationInst:
  transformation: QuantTransformation
  tensor_id: int
  producer: Optional[int]
  consumers: list[int]
  parameters: Union[None, UniformQuantParams, NonLinearQuantParams] = None


@dataclasses.dataclass
class TensorTransformationInsts:
  tensor_name: str
  subgraph_id: int
  instructions: Optional[list[TransformationInst]]


@dataclasses.dataclass(frozen=True)
class TransformationInfo:
  op_id: int
  num_ops_added: int
  output_tensor_id: int


@dataclasses.dataclass(frozen=True)
class IOOperator:
  inputs: list[int]
  outputs: list[int]
  op_key: TFLOperationName


ConfigCheckPolicyDict = OrderedDict

# ref: qtyping.py:701-710
GetTensorQuantParamsFuncSignature = Callable[
    [OpInfo, TensorQuantizationConfig, Optional[np.ndarray], Optional[dict[str, Any]]],
    UniformQuantParams]
