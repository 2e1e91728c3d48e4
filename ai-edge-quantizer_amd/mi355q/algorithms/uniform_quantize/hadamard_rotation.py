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
  h = hadamard_size_for(tensor_content.shape[axis], max_size)
  x = uniform_quantize_tensor._as_f32_exact(tensor_content)  # pylint: disable=protected-access
  rt.require_gpu()
  rotated = ops.hadamard_rotate(rt.to_device(x.reshape(-1)), max(h, 2) if h < 2 else h)
  return rt.to_numpy(rotated).reshape(tensor_content.shape), h, np.ones(h, dtype=np.int8)


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
  w_rot, h, vec = _rotate_with_diagonal_hadamard(
      tensor_content, axis=tensor_content.ndim - 1,
      max_size=tensor_quant_config.algorithm_params.get("max_hadamard_size"))
  p = octav.get_tensor_quant_params(op_info, tensor_quant_config, w_rot, tensor_qsv)
  return qtyping.UniformQuantParams(
      quantized_dimension=p.quantized_dimension, num_bits=p.num_bits, scale=p.scale,
      zero_point=p.zero_point, symmetric=p.symmetric, quantized_data=p.quantized_data,
      block_size=p.block_size,
      hadamard=qtyping.UniformQuantParams.HadamardRotationParams(
          random_binary_vector=vec, hadamard_size=h))
