"""TFLite schema types used around the hot path.

The reference takes these from `ai_edge_litert.tools.flatbuffer_utils` (ref: qtyping.py:37-79).
The table classes (`ModelT`, `TensorT`, ...) are the object tree of this build's own flatbuffer
reader / writer (`utils/tflite_flatbuffer.py`); field names follow the flatbuffers object API, so
models parsed from `.tflite` files and models assembled in memory are the same kind of object.
"""
from __future__ import annotations

import enum
from typing import Any, Optional

from .utils import tflite_flatbuffer as _fb


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


# BuiltinOperator codes 0..161 in schema order (checked against the op each of the reference's
# single-op test models holds), plus the StableHLO composite op.
_BUILTIN_OPERATOR_NAMES = """
    ADD AVERAGE_POOL_2D CONCATENATION CONV_2D DEPTHWISE_CONV_2D DEPTH_TO_SPACE DEQUANTIZE EMBEDDING_LOOKUP
    FLOOR FULLY_CONNECTED HASHTABLE_LOOKUP L2_NORMALIZATION L2_POOL_2D LOCAL_RESPONSE_NORMALIZATION
    LOGISTIC LSH_PROJECTION LSTM MAX_POOL_2D MUL RELU RELU_N1_TO_1 RELU6 RESHAPE RESIZE_BILINEAR
    RNN SOFTMAX SPACE_TO_DEPTH SVDF TANH CONCAT_EMBEDDINGS SKIP_GRAM CALL CUSTOM EMBEDDING_LOOKUP_SPARSE
    PAD UNIDIRECTIONAL_SEQUENCE_RNN GATHER BATCH_TO_SPACE_ND SPACE_TO_BATCH_ND TRANSPOSE MEAN
    SUB DIV SQUEEZE UNIDIRECTIONAL_SEQUENCE_LSTM STRIDED_SLICE BIDIRECTIONAL_SEQUENCE_RNN EXP
    TOPK_V2 SPLIT LOG_SOFTMAX DELEGATE BIDIRECTIONAL_SEQUENCE_LSTM CAST PRELU MAXIMUM ARG_MAX
    MINIMUM LESS NEG PADV2 GREATER GREATER_EQUAL LESS_EQUAL SELECT SLICE SIN TRANSPOSE_CONV SPARSE_TO_DENSE
    TILE EXPAND_DIMS EQUAL NOT_EQUAL LOG SUM SQRT RSQRT SHAPE POW ARG_MIN FAKE_QUANT REDUCE_PROD
    REDUCE_MAX PACK LOGICAL_OR ONE_HOT LOGICAL_AND LOGICAL_NOT UNPACK REDUCE_MIN FLOOR_DIV REDUCE_ANY
    SQUARE ZEROS_LIKE FILL FLOOR_MOD RANGE RESIZE_NEAREST_NEIGHBOR LEAKY_RELU SQUARED_DIFFERENCE
    MIRROR_PAD ABS SPLIT_V UNIQUE CEIL REVERSE_V2 ADD_N GATHER_ND COS WHERE RANK ELU REVERSE_SEQUENCE
    MATRIX_DIAG QUANTIZE MATRIX_SET_DIAG ROUND HARD_SWISH IF WHILE NON_MAX_SUPPRESSION_V4 NON_MAX_SUPPRESSION_V5
    SCATTER_ND SELECT_V2 DENSIFY SEGMENT_SUM BATCH_MATMUL PLACEHOLDER_FOR_GREATER_OP_CODES CUMSUM
    CALL_ONCE BROADCAST_TO RFFT2D CONV_3D IMAG REAL COMPLEX_ABS HASHTABLE HASHTABLE_FIND HASHTABLE_IMPORT
    HASHTABLE_SIZE REDUCE_ALL CONV_3D_TRANSPOSE VAR_HANDLE READ_VARIABLE ASSIGN_VARIABLE BROADCAST_ARGS
    RANDOM_STANDARD_NORMAL BUCKETIZE RANDOM_UNIFORM MULTINOMIAL GELU DYNAMIC_UPDATE_SLICE RELU_0_TO_1
    UNSORTED_SEGMENT_PROD UNSORTED_SEGMENT_MAX UNSORTED_SEGMENT_SUM ATAN2 UNSORTED_SEGMENT_MIN
    SIGN BITCAST BITWISE_XOR RIGHT_SHIFT
""".split()
BuiltinOperator = enum.IntEnum(
    "BuiltinOperator",
    {**{n: i for i, n in enumerate(_BUILTIN_OPERATOR_NAMES)}, "STABLEHLO_COMPOSITE": 206})


class BuiltinOptions(enum.IntEnum):
  """Union codes of the option tables this build reads field by field."""
  NONE = 0
  FullyConnectedOptions = 8
  ReshapeOptions = 17
  MulOptions = 21
  DequantizeOptions = 38
  QuantizeOptions = 89


class BuiltinOptions2(enum.IntEnum):
  NONE = 0
  StableHLOCompositeOptions = 21


ModelT = _fb.ModelT
SubGraphT = _fb.SubGraphT
TensorT = _fb.TensorT
BufferT = _fb.BufferT
OperatorT = _fb.OperatorT
OperatorCodeT = _fb.OperatorCodeT
QuantizationParametersT = _fb.QuantizationParametersT
BlockwiseQuantizationT = _fb.BlockwiseQuantizationT
FullyConnectedOptionsT = _fb.FullyConnectedOptionsT
BatchMatMulOptionsT = _fb.BatchMatMulOptionsT
MulOptionsT = _fb.MulOptionsT
StableHLOCompositeOptionsT = _fb.StableHLOCompositeOptionsT
SignatureDefT = _fb.SignatureDefT
TensorMapT = _fb.TensorMapT
MetadataT = _fb.MetadataT


def tensor_name(tensor: Any) -> Optional[str]:
  n = tensor.name
  return n.decode("utf-8") if isinstance(n, (bytes, bytearray)) else n


class ActivationFunctionType(enum.IntEnum):
  NONE = 0
  RELU = 1
  RELU_N1_TO_1 = 2
  RELU6 = 3
  TANH = 4
  SIGN_BIT = 5
