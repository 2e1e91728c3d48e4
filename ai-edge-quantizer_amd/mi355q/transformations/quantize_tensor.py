"""QUANTIZE_TENSOR transformation: store packed bytes + quantization metadata.

ref: transformations/quantize_tensor.py:29-224. The bit packing runs on the GPU
(pack_data -> mi355q_pack_bits); everything else is flatbuffer bookkeeping.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np

from .. import qtyping
from ..algorithms.uniform_quantize import uniform_quantize_tensor
from . import transformation_utils


def quant_params_to_tflite_type(bitwidth: int) -> Optional[qtyping.TensorType]:
  """ref :29-54."""
  if bitwidth == 2:
    return qtyping.TensorType.INT2
  if bitwidth == 4:
    return qtyping.TensorType.INT4
  if 1 < bitwidth <= 8:
    return qtyping.TensorType.INT8
  if 8 < bitwidth <= 16:
    return qtyping.TensorType.INT16
  if 16 < bitwidth <= 32:
    return qtyping.TensorType.INT32
  if 32 < bitwidth <= 64:
    return qtyping.TensorType.INT64
  raise ValueError(f"Unsupported bitwidth {bitwidth}.I")


def nonlinear_quant_params_to_tflite_type(bitwidth: int) -> Optional[qtyping.TensorType]:
  if bitwidth == 16:
    return qtyping.TensorType.FLOAT16
  if bitwidth == 32:
    return qtyping.TensorType.FLOAT32
  raise ValueError(f"Unsupported nonlinear params: {bitwidth}")


def _perform_channelwise_quantization(ti: transformation_utils.TransformationInput):
  """scale f32[ch], zeroPoint int64[ch], quantizedDimension (ref :76-104)."""
  p = ti.quant_params
  q = qtyping.QuantizationParametersT()
  # (scales still in HBM keep their place in the flatbuffer and are read last: runtime.LateVector)
  q.scale = transformation_utils.rt.late_vector(p.scale, np.dtype(np.float32))
  if p.zero_point is not None:
    q.zeroPoint = np.ravel(p.zero_point).astype(np.int64, copy=False)
  if p.quantized_dimension is not None:
    q.quantizedDimension = p.quantized_dimension
  return q


def _perform_blockwise_quantization(ti: transformation_utils.TransformationInput):
  """f16 `<name>_scales` side tensor, no zero points, blockSize (ref :107-147)."""
  p = ti.quant_params
  q = qtyping.QuantizationParametersT()
  q.detailsType = qtyping.QuantizationDetails.BlockwiseQuantization
  tensor = ti.subgraph.tensors[ti.tensor_id]
  details = qtyping.BlockwiseQuantizationT()
  f16 = getattr(p.scale, "f16", None)     # written by the launch that quantized (requant_queue)
  if f16 is not None and transformation_utils._still_on_device(f16):   # pylint: disable=protected-access
    scales_f16 = f16            # stays in HBM: its buffer is laid out from its size and written by the file writer
  else:
    scales_f16 = np.asarray(f16) if f16 is not None else uniform_quantize_tensor.round_to_bf16(
        np.asarray(p.scale, dtype=np.float32)).astype(np.float16)
  name = tensor.name if isinstance(tensor.name, (bytes, bytearray)) else str(tensor.name).encode()
  details.scales = transformation_utils.add_new_constant_tensor(
      name + b"_scales", scales_f16, qtyping.TensorType.FLOAT16, ti.subgraph, ti.model)
  details.zeroPoints = -1
  details.blockSize = p.block_size
  q.details = details
  q.quantizedDimension = 0  # hard-coded in the reference (b/443830202)
  return q


def quantize_tensor(ti: transformation_utils.TransformationInput) -> qtyping.TransformationInfo:
  """ref :150-224."""
  tensor = ti.subgraph.tensors[ti.tensor_id]
  buffer_id = tensor.buffer
  p = ti.quant_params
  if buffer_id and p.quantized_data is not None:
    origin = ti.buffer_origin.get(buffer_id)
    if origin is not None and origin is p:
      logging.debug("Quantized data for tensor %s already packed to buffer %s", tensor.name,
                    buffer_id)
    else:
      if origin is not None:
        logging.warning("Quantized data for tensor %s is overriding other previously quantized"
                        " data in buffer %s.", tensor.name, buffer_id)
      ti.buffer_origin[buffer_id] = p
      ready = getattr(p.quantized_data, "packed", None)    # packed by the quantizing launch
      if ready is None and p.num_bits not in (2, 4) and hasattr(p.quantized_data, "copy_into") \
          and p.quantized_data.dtype in (np.int8, np.uint8):
        ready = p.quantized_data                           # one byte per value in HBM: its bytes are the buffer (pack_data
                                                           # returns every width but 2 and 4 unchanged, ref :293-353)
      ti.model.buffers[buffer_id].data = ready if ready is not None else transformation_utils.pack_data(
          p.num_bits, np.ravel(np.asarray(p.quantized_data)).view(np.uint8))
  if isinstance(p, qtyping.UniformQuantParams):
    tensor.quantization = (_perform_channelwise_quantization(ti) if p.block_size == 0
                           else _perform_blockwise_quantization(ti))
    tensor.type = quant_params_to_tflite_type(p.num_bits)
  if isinstance(p, qtyping.NonLinearQuantParams):
    tensor.type = nonlinear_quant_params_to_tflite_type(p.num_bits)
  return qtyping.TransformationInfo(0, num_ops_added=0, output_tensor_id=ti.tensor_id)
