"""Minimal stand-ins for the LiteRT flatbuffer object API used on the hot path.

The reference takes these types from `ai_edge_litert.tools.flatbuffer_utils`
(ref: qtyping.py:37-79). Only plain attribute containers are needed around the
calibration / requantization path: tensors, operators, buffers and the
quantization records that transformations/quantize_tensor.py fills in. Field
names follow the TFLite schema's object API so a real `TensorT` etc. can be
passed in unchanged (duck typing).
"""
from __future__ import annotations

import enum
from typing import Any, Optional


class TensorType(enum.IntEnum):
  """TFLite schema TensorType codes."""
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


# NumPy dtype name per tensor type code (what get_tensor_data views buffers as).
NUMPY_DTYPE = {
    TensorType.FLOAT32: "float32", TensorType.FLOAT16: "float16", TensorType.INT32: "int32",
    TensorType.UINT8: "uint8", TensorType.INT64: "int64", TensorType.BOOL: "bool",
    TensorType.INT16: "int16", TensorType.INT8: "int8", TensorType.FLOAT64: "float64",
    TensorType.UINT64: "uint64", TensorType.UINT32: "uint32", TensorType.UINT16: "uint16",
    TensorType.INT4: "int4", TensorType.INT2: "int2",
}


class QuantizationDetails(enum.IntEnum):
  NONE = 0
  CustomQuantization = 1
  BlockwiseQuantization = 2


class BuiltinOperator(enum.IntEnum):
  """Subset of the TFLite BuiltinOperator codes (the ops this build materializes)."""
  ADD = 0
  AVERAGE_POOL_2D = 1
  CONCATENATION = 2
  CONV_2D = 3
  DEPTHWISE_CONV_2D = 4
  EMBEDDING_LOOKUP = 7
  FULLY_CONNECTED = 9
  CUSTOM = 32
  TRANSPOSE_CONV = 67
  BATCH_MATMUL = 126


class _Record:
  _fields: dict[str, Any] = {}

  def __init__(self, **kw):
    for k, v in self._fields.items():
      setattr(self, k, v() if callable(v) else v)
    for k, v in kw.items():
      setattr(self, k, v)

  def __repr__(self):
    body = ", ".join(f"{k}={getattr(self, k)!r}" for k in self._fields)
    return f"{type(self).__name__}({body})"


class BufferT(_Record):
  _fields = {"data": None, "offset": 0, "size": 0}


class BlockwiseQuantizationT(_Record):
  _fields = {"scales": 0, "zeroPoints": 0, "blockSize": 0}


class QuantizationParametersT(_Record):
  _fields = {"min": None, "max": None, "scale": None, "zeroPoint": None,
             "detailsType": QuantizationDetails.NONE, "details": None,
             "quantizedDimension": 0}


class TensorT(_Record):
  _fields = {"shape": None, "type": TensorType.FLOAT32, "buffer": 0, "name": None,
             "quantization": None, "isVariable": False, "shapeSignature": None,
             "hasRank": False}


class FullyConnectedOptionsT(_Record):
  _fields = {"fusedActivationFunction": 0, "weightsFormat": 0, "keepNumDims": False,
             "asymmetricQuantizeInputs": False, "quantizedBiasType": 0}


class OperatorT(_Record):
  _fields = {"opcodeIndex": 0, "inputs": None, "outputs": None, "builtinOptionsType": 0,
             "builtinOptions": None, "customOptions": None}


class OperatorCodeT(_Record):
  _fields = {"deprecatedBuiltinCode": 0, "customCode": None, "version": 1, "builtinCode": 0}


class SubGraphT(_Record):
  _fields = {"tensors": list, "inputs": list, "outputs": list, "operators": list, "name": None}


class ModelT(_Record):
  _fields = {"version": 3, "operatorCodes": list, "subgraphs": list, "description": None,
             "buffers": list, "metadataBuffer": None, "metadata": None, "signatureDefs": None}


def tensor_name(tensor: Any) -> Optional[str]:
  n = tensor.name
  return n.decode("utf-8") if isinstance(n, (bytes, bytearray)) else n
