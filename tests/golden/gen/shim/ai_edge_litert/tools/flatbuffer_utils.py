"""Stand-in for ai_edge_litert.tools.flatbuffer_utils (absent here).

Provides attribute-bag versions of the flatbuffer object-API classes and enum
holders so the reference's *arithmetic* modules import. No serialization.
"""
import pathlib
import types


class _Bag:
  def __init__(self, **kw):
    self.__dict__.update(kw)


class _AutoEnum:
  """Attribute access hands out distinct, stable integer codes."""

  def __init__(self, seed=None):
    object.__setattr__(self, "_codes", dict(seed or {}))

  def __getattr__(self, name):
    if name.startswith("__"):
      raise AttributeError(name)
    codes = object.__getattribute__(self, "_codes")
    if name not in codes:
      codes[name] = 1000 + len(codes)
    return codes[name]


BuiltinOperator = _AutoEnum({
    "ADD": 0, "AVERAGE_POOL_2D": 1, "CONCATENATION": 2, "CONV_2D": 3,
    "DEPTHWISE_CONV_2D": 4, "EMBEDDING_LOOKUP": 7, "FULLY_CONNECTED": 9,
    "CUSTOM": 32, "BATCH_MATMUL": 126,
})
BuiltinOptions = _AutoEnum()
BuiltinOptions2 = _AutoEnum()
ActivationFunctionType = _AutoEnum({"NONE": 0})
QuantizationDetails = _AutoEnum({"NONE": 0, "CustomQuantization": 1,
                                 "BlockwiseQuantization": 2})


class TensorType:
  FLOAT32 = 0
  FLOAT16 = 1
  INT32 = 2
  UINT8 = 3
  INT64 = 4
  STRING = 5
  BOOL = 6
  INT16 = 7
  COMPLEX64 = 8
  INT8 = 9
  FLOAT64 = 10
  COMPLEX128 = 11
  UINT64 = 12
  RESOURCE = 13
  VARIANT = 14
  UINT32 = 15
  UINT16 = 16
  INT4 = 17
  BFLOAT16 = 18
  INT2 = 19


class TensorT(_Bag):
  def __init__(self, **kw):
    super().__init__(name=None, shape=None, type=0, buffer=0, quantization=None,
                     isVariable=False, shapeSignature=None, hasRank=False)
    self.__dict__.update(kw)


class OperatorT(_Bag):
  def __init__(self, **kw):
    super().__init__(opcodeIndex=0, inputs=None, outputs=None,
                     builtinOptions=None, builtinOptionsType=0,
                     builtinOptions2=None, builtinOptions2Type=0,
                     customOptions=None)
    self.__dict__.update(kw)


class BufferT(_Bag):
  def __init__(self, **kw):
    super().__init__(data=None, offset=0, size=0)
    self.__dict__.update(kw)


class QuantizationParametersT(_Bag):
  def __init__(self, **kw):
    super().__init__(min=None, max=None, scale=None, zeroPoint=None,
                     detailsType=0, details=None, quantizedDimension=0)
    self.__dict__.update(kw)


class BlockwiseQuantizationT(_Bag):
  def __init__(self, **kw):
    super().__init__(scales=0, zeroPoints=0, blockSize=0)
    self.__dict__.update(kw)


class SubGraphT(_Bag):
  def __init__(self, **kw):
    super().__init__(tensors=[], inputs=[], outputs=[], operators=[], name=None)
    self.__dict__.update(kw)


class ModelT(_Bag):
  def __init__(self, **kw):
    super().__init__(version=3, operatorCodes=[], subgraphs=[], description=None,
                     buffers=[], metadataBuffer=None, metadata=None,
                     signatureDefs=None)
    self.__dict__.update(kw)


class OperatorCodeT(_Bag):
  def __init__(self, **kw):
    super().__init__(deprecatedBuiltinCode=0, customCode=None, version=1,
                     builtinCode=0)
    self.__dict__.update(kw)


class FullyConnectedOptionsT(_Bag):
  def __init__(self, **kw):
    super().__init__(fusedActivationFunction=0, weightsFormat=0, keepNumDims=False,
                     asymmetricQuantizeInputs=False, quantizedBiasType=0)
    self.__dict__.update(kw)


class StableHLOCompositeOptionsT(_Bag):
  pass


class _Unavailable:
  def __init__(self, *a, **k):
    raise NotImplementedError("flatbuffer (de)serialization is not available in the oracle shim")


Buffer = Model = Operator = OperatorCode = SubGraph = Tensor = _Unavailable
StableHLOCompositeOptions = _Unavailable
schema_fb = types.SimpleNamespace(MulOptions=_Unavailable, MulOptionsT=_Bag)

Path = str | pathlib.Path
BufferType = bytes | bytearray | memoryview
Endiness = _AutoEnum({"LITTLE": 0, "BIG": 1})


def _no(*a, **k):
  raise NotImplementedError("not available in the oracle shim")


read_model = read_model_from_bytearray = write_model = _no


def get_options_as(op, cls):
  """Options object of `op` when it is an instance of `cls`, else None."""
  for attr in ("builtinOptions2", "builtinOptions"):
    o = getattr(op, attr, None)
    if isinstance(o, cls):
      return o
  return None

convert_object_to_bytearray = _no
