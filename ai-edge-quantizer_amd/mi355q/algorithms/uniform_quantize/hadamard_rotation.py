"""Hadamard rotation + OCTAV, GPU backed.

Mirror of ref: algorithms/uniform_quantize/hadamard_rotation.py:45-203 (the
tensor math; the graph-rewrite materializers live with the registry). The
rotation is an in-LDS fast Walsh-Hadamard transform (mi355q_hadamard_rotate_f32),
the rotated weight stays in HBM and feeds the OCTAV + requant kernels directly.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import runtime as rt
from . import octav
from . import uniform_quantize_tensor

CUSTOM_OP_ALGORITHM_KEY = "HADAMARD_ROTATION"
DECOMPOSED_ALGORITHM_KEY = "DECOMPOSED_HADAMARD_ROTATION"


def _make_hadamard_matrix(size: int) -> np.ndarray:
  """Sylvester H_size / sqrt(size), float32 (host; only used to emit graph constants)."""
  size = int(size)
  if size <= 0 or size & (size - 1):
    raise ValueError("Hadamard matrix size must be a power of 2. ")
  h2 = np.array([[1, 1], [1, -1]], dtype=np.int8)
  h, n = h2, 2
  while n < size:
    h, n = np.kron(h, h2), n * 2
  return h / np.sqrt(n, dtype=np.float32)


def hadamard_size_for(last_dim: int, max_size: int | None = None) -> int:
  """Largest power-of-two factor of last_dim, capped (ref :118-123)."""
  h = int(np.gcd(last_dim, 2**30))
  if max_size:
    h = min(h, 1 << (int(max_size).bit_length() - 1))
  return h


def _rotate_with_diagonal_hadamard(tensor_content: np.ndarray, axis: int,
                                   max_size: int | None = None):
  """(rotated, hadamard_size, random_vector). ref :93-134."""
  if axis != tensor_content.ndim - 1:
    raise ValueError("Hadamard rotation is only supported for tensors with quantized"
                     " dimension 0 (rotate last dimension).")
  rotated, h, vec = _rotate_in_hbm(tensor_content, axis, max_size)
  return np.asarray(rotated), h, vec


def _rotate_in_hbm(tensor_content, axis: int, max_size: int | None = None):
  """The same, with the rotated tensor left in HBM (runtime.HbmArray) for the clip search and the
  quantizing launch that follow."""
  if axis != tensor_content.ndim - 1:
    raise ValueError("Hadamard rotation is only supported for tensors with quantized"
                     " dimension 0 (rotate last dimension).")
  h = hadamard_size_for(tensor_content.shape[axis], max_size)
  x = uniform_quantize_tensor._as_f32_exact(tensor_content)  # pylint: disable=protected-access
  rt.require_gpu()
  rotated = ops.hadamard_rotate(rt.to_device(x.reshape(-1)), max(h, 2) if h < 2 else h)
  return rt.HbmArray(rotated.reshape(tuple(tensor_content.shape))), h, np.ones(h, dtype=np.int8)


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: Optional[np.ndarray] = None, tensor_qsv: Optional[dict[str, Any]] = None,
) -> qtyping.UniformQuantParams:
  """rotate -> OCTAV (ref :137-203)."""
  if tensor_content is None:
    raise ValueError("Hadamard rotation is only supported for weight tensors.")
  if tensor_qsv is not None:
    raise ValueError("Hadamard rotation is not supported for static quantization.")
  if tensor_content.ndim < 2:
    raise ValueError("Hadamard rotation is only supported for tensors with rank >= 2.")
  w_rot, h, vec = _rotate_in_hbm(
      tensor_content, axis=tensor_content.ndim - 1,
      max_size=tensor_quant_config.algorithm_params.get("max_hadamard_size"))
  p = octav.get_tensor_quant_params(op_info, tensor_quant_config, w_rot, tensor_qsv)
  return qtyping.UniformQuantParams(
      quantized_dimension=p.quantized_dimension, num_bits=p.num_bits, scale=p.scale,
      zero_point=p.zero_point, symmetric=p.symmetric, quantized_data=p.quantized_data,
      block_size=p.block_size,
      hadamard=qtyping.UniformQuantParams.HadamardRotationParams(
          random_binary_vector=vec, hadamard_size=h))


# ------------------------------------------------------------ materializers ---
# ref :206-500. The weight gets QUANTIZE_TENSOR with the rotated + OCTAV params;
# the tensor on the other side of the matmul gets an INSERT_(DECOMPOSED_)HADAMARD_
# ROTATION instruction carrying the same params (the graph rewrite itself is
# outside the hot path).
from ...utils import tfl_flatbuffer_utils as _fb  # noqa: E402

_T = qtyping.QuantTransformation


def _link(op_info, transformations, params=None):
  return qtyping.OpToTensorParams(subgraph_op_id=op_info.subgraph_op_index,
                                  parameters=params, transformations=transformations)


def _weight_params(op_info, graph_info, weight_tensor, cache):
  cfg = op_info.op_quant_config.weight_tensor_config
  params = cache.lookup(weight_tensor.buffer, cfg) if cache is not None else None
  if not params:
    params = get_tensor_quant_params(op_info, cfg,
                                     _fb.get_tensor_data(weight_tensor, graph_info.buffers), None)
    if cache is not None:
      cache.insert(weight_tensor.buffer, cfg, params)
  return params


def _materialize_fully_connected(op_info, graph_info, tensor_quant_params_cache,
                                 is_decomposed=False, tensor_name_to_qsv=None):
  del tensor_name_to_qsv
  if op_info.op_quant_config.weight_tensor_config is None:
    raise ValueError("Weight tensor quantization config is not provided for Hadamard Rotation"
                     " quantization.")
  tensors = graph_info.subgraph_tensors
  inp, w, bias = (tensors[op_info.op.inputs[k]] for k in range(3))
  out = tensors[op_info.op.outputs[0]]
  params = _weight_params(op_info, graph_info, w, tensor_quant_params_cache)
  rot = _T.INSERT_DECOMPOSED_HADAMARD_ROTATION if is_decomposed else _T.INSERT_HADAMARD_ROTATION
  P = qtyping.TensorTransformationParams
  return [
      P(tensor_name=_fb.get_tensor_name(inp), consumers=[_link(op_info, [rot], params)]),
      P(tensor_name=_fb.get_tensor_name(w), consumers=[_link(op_info, [_T.QUANTIZE_TENSOR], params)]),
      P(tensor_name=_fb.get_tensor_name(bias), consumers=[_link(op_info, [_T.NO_QUANTIZE])]),
      P(tensor_name=_fb.get_tensor_name(out), producer=_link(op_info, [_T.NO_QUANTIZE])),
  ]


def materialize_fully_connected_custom_op(op_info, graph_info, tensor_quant_params_cache,
                                          tensor_name_to_qsv=None):
  return _materialize_fully_connected(op_info, graph_info, tensor_quant_params_cache, False,
                                      tensor_name_to_qsv)


def materialize_fully_connected_decomposed(op_info, graph_info, tensor_quant_params_cache,
                                           tensor_name_to_qsv=None):
  return _materialize_fully_connected(op_info, graph_info, tensor_quant_params_cache, True,
                                      tensor_name_to_qsv)


def _materialize_embedding_lookup(op_info, graph_info, tensor_quant_params_cache,
                                  is_decomposed=False, tensor_name_to_qsv=None):
  del tensor_name_to_qsv
  tensors = graph_info.subgraph_tensors
  lookup, emb = tensors[op_info.op.inputs[0]], tensors[op_info.op.inputs[1]]
  out = tensors[op_info.op.outputs[0]]
  params = _weight_params(op_info, graph_info, emb, tensor_quant_params_cache)
  rot = _T.INSERT_DECOMPOSED_HADAMARD_ROTATION if is_decomposed else _T.INSERT_HADAMARD_ROTATION
  P = qtyping.TensorTransformationParams
  return [
      P(tensor_name=_fb.get_tensor_name(lookup), consumers=[_link(op_info, [_T.NO_QUANTIZE])]),
      P(tensor_name=_fb.get_tensor_name(emb), consumers=[_link(op_info, [_T.QUANTIZE_TENSOR], params)]),
      P(tensor_name=_fb.get_tensor_name(out), producer=_link(op_info, [rot], params)),
  ]


def materialize_embedding_lookup_custom_op(op_info, graph_info, tensor_quant_params_cache,
                                           tensor_name_to_qsv=None):
  return _materialize_embedding_lookup(op_info, graph_info, tensor_quant_params_cache, False,
                                       tensor_name_to_qsv)


def materialize_embedding_lookup_decomposed(op_info, graph_info, tensor_quant_params_cache,
                                            tensor_name_to_qsv=None):
  return _materialize_embedding_lookup(op_info, graph_info, tensor_quant_params_cache, True,
                                       tensor_name_to_qsv)
