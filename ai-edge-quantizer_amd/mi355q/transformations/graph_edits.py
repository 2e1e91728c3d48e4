"""Transformations that change the graph around a tensor: Q / DQ op insertion and constant
duplication (ref: transformations/quant_insert.py, dequant_insert.py, duplicate_buffer.py,
duplicate_tensor.py). Pure flatbuffer bookkeeping; the arithmetic lives in quantize_tensor.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from .. import qtyping
from ..utils import tfl_flatbuffer_utils
from . import quantize_tensor
from . import transformation_utils

_Input = transformation_utils.TransformationInput


def _raw_name(tensor: Any) -> bytes:
  return tensor.name if isinstance(tensor.name, (bytes, bytearray)) else str(tensor.name).encode()


def _splice_after(ti: _Input, new_tensor_id: int, op: Any) -> int:
  """Re-route the listed consumers (and graph outputs) from the tensor to `new_tensor_id` and
  insert `op` right before the first of them - or right after the producer when that comes
  later. Returns the op's index."""
  sg = ti.subgraph
  for consumer in ti.consumers:
    inputs = sg.operators[consumer].inputs
    for k, tid in enumerate(inputs):
      if tid == ti.tensor_id:
        inputs[k] = new_tensor_id
  for k, tid in enumerate(sg.outputs):
    if tid == ti.tensor_id:
      sg.outputs[k] = new_tensor_id
  at = max(ti.producer + 1, min(ti.consumers))
  sg.operators.insert(at, op)
  return at


def insert_quant(ti: _Input) -> qtyping.TransformationInfo:
  """float tensor -> QUANTIZE -> `<name>_quantized` for the given consumers (ref quant_insert.py)."""
  code = transformation_utils.add_op_code(qtyping.BuiltinOperator.QUANTIZE, ti.model.operatorCodes)
  tensor = ti.subgraph.tensors[ti.tensor_id]
  new_id = transformation_utils.add_new_activation_tensor(
      _raw_name(tensor) + b"_quantized", tensor.shape, qtyping.TensorType.FLOAT32, ti.subgraph)
  quantize_tensor.quantize_tensor(_Input(new_id, ti.model, ti.subgraph, ti.producer, ti.consumers,
                                         ti.quant_params))
  op = qtyping.OperatorT(opcodeIndex=code, inputs=[ti.tensor_id], outputs=[new_id])
  at = _splice_after(ti, new_id, op)
  return qtyping.TransformationInfo(op_id=at, num_ops_added=1, output_tensor_id=new_id)


def insert_dequant(ti: _Input) -> qtyping.TransformationInfo:
  """the tensor becomes integer; DEQUANTIZE -> `<name>_dequant` feeds the given consumers
  (ref dequant_insert.py)."""
  code = transformation_utils.add_op_code(qtyping.BuiltinOperator.DEQUANTIZE, ti.model.operatorCodes)
  tensor = ti.subgraph.tensors[ti.tensor_id]
  new_id = transformation_utils.add_new_activation_tensor(
      _raw_name(tensor) + b"_dequant", tensor.shape, qtyping.TensorType.FLOAT32, ti.subgraph)
  op = qtyping.OperatorT(opcodeIndex=code, inputs=[ti.tensor_id], outputs=[new_id])
  quantize_tensor.quantize_tensor(ti)
  at = _splice_after(ti, new_id, op)
  return qtyping.TransformationInfo(op_id=at, num_ops_added=1, output_tensor_id=new_id)


def _constant_bytes(ti: _Input, what: str):
  tensor = ti.subgraph.tensors[ti.tensor_id]
  data = ti.model.buffers[tensor.buffer].data
  if data is None:
    raise ValueError(f"{what} transformation supports only constant tensors. Tensor"
                     f" {tfl_flatbuffer_utils.get_tensor_name(tensor)} is not constant.")
  return tensor, data


def duplicate_buffer(ti: _Input) -> qtyping.TransformationInfo:
  """Give the tensor a private copy of its (shared) buffer (ref duplicate_buffer.py)."""
  tensor, data = _constant_bytes(ti, "Duplicate Buffer")
  tensor.buffer = transformation_utils.get_constant_buffer(data, ti.model, force_duplicate_buffer=True)
  return qtyping.TransformationInfo(op_id=0, num_ops_added=0, output_tensor_id=ti.tensor_id)


def duplicate_tensor(ti: _Input) -> qtyping.TransformationInfo:
  """`<name>_duplicated_<id>` with its own buffer, consumed by the given ops instead of the
  original (ref duplicate_tensor.py)."""
  tensor, data = _constant_bytes(ti, "Duplicate Tensor")
  name = tfl_flatbuffer_utils.get_tensor_name(tensor)
  new_id = transformation_utils.add_new_constant_tensor(
      f"{name}_duplicated".encode(), data, tensor.type, ti.subgraph, ti.model,
      tensor_shape=tensor.shape, force_duplicate_buffer=True)
  ti.subgraph.tensors[new_id].name += f"_{new_id}".encode()
  for consumer in ti.consumers:
    inputs = ti.subgraph.operators[consumer].inputs
    for k, tid in enumerate(inputs):
      if tid == ti.tensor_id:
        inputs[k] = new_id
        break
  return qtyping.TransformationInfo(op_id=0, num_ops_added=0, output_tensor_id=new_id)


def _op_code_of(ti: _Input, op_index: int) -> int:
  return ti.model.operatorCodes[ti.subgraph.operators[op_index].opcodeIndex].builtinCode


def _sylvester_hadamard(size: int) -> np.ndarray:
  """H_size / sqrt(size) (float64 until the caller casts), ref insert_decomposed_...py:57-79."""
  if size <= 0 or size & (size - 1):
    raise ValueError("Hadamard matrix size must be a power of 2. ")
  h = base = np.array([[1, 1], [1, -1]])
  n = 2
  while n < size:
    h = np.kron(h, base)
    n *= 2
  return h / np.sqrt(size)


_HADAMARD_F32: dict[int, np.ndarray] = {}


def _sylvester_hadamard_f32(size: int) -> np.ndarray:
  """The float32 constant the rotation op multiplies by, built once per size (a model inserts one per
  rotated FULLY_CONNECTED: 126 of order 2048 in an 18-layer decoder; read-only, shared by the writers)."""
  m = _HADAMARD_F32.get(size)
  if m is None:
    m = _sylvester_hadamard(size).astype(np.float32)
    m.setflags(write=False)
    _HADAMARD_F32[size] = m
  return m


def insert_decomposed_hadamard_rotation(ti: _Input) -> qtyping.TransformationInfo:
  """x -> RESHAPE(-1, h) -> FULLY_CONNECTED with H_h / sqrt(h) -> RESHAPE(x.shape), feeding the
  FULLY_CONNECTED consumers (or everything after an EMBEDDING_LOOKUP producer) whose weights
  were rotated by the same matrix (ref insert_decomposed_hadamard_rotation.py:82-265)."""
  p = ti.quant_params
  if not isinstance(p, qtyping.UniformQuantParams):
    raise ValueError("Hadamard rotation supports uniform quantization only")
  if p.hadamard is None:
    raise ValueError("Hadamard rotation quantization params are not set but op insertion is"
                     " requested.")
  sg, model = ti.subgraph, ti.model
  tensor = sg.tensors[ti.tensor_id]
  if tensor.type != qtyping.TensorType.FLOAT32:
    raise ValueError(f"The Hadamard rotation op supports float32 tensors only. Got {tensor.type}"
                     " tensor.")
  name = _raw_name(tensor)
  h = int(p.hadamard.hadamard_size)
  flat = [int(np.prod(tensor.shape)) // h, h]
  i32, f32 = qtyping.TensorType.INT32, qtyping.TensorType.FLOAT32
  pre_shape = transformation_utils.add_new_constant_tensor(
      name + b"_prerotate_shape", np.array(flat, np.int32), i32, sg, model)
  pre_out = transformation_utils.add_new_activation_tensor(name + b"_prerotate_reshaped", flat, f32, sg)
  reshape_code = transformation_utils.add_op_code(qtyping.BuiltinOperator.RESHAPE, model.operatorCodes, "RESHAPE")
  pre = qtyping.OperatorT(opcodeIndex=reshape_code, inputs=[ti.tensor_id, pre_shape], outputs=[pre_out])
  matrix = transformation_utils.add_new_constant_tensor(
      name + b"_hadamard_matrix", _sylvester_hadamard_f32(h), f32, sg, model,
      allow_tensor_sharing=True)
  rotated = transformation_utils.add_new_activation_tensor(name + b"_rotated", flat, f32, sg)
  fc_code = transformation_utils.add_op_code(qtyping.BuiltinOperator.FULLY_CONNECTED, model.operatorCodes,
                                             "FULLY_CONNECTED")
  fc = qtyping.OperatorT(opcodeIndex=fc_code, inputs=[pre_out, matrix], outputs=[rotated],
                         builtinOptionsType=int(qtyping.BuiltinOptions.FullyConnectedOptions),
                         builtinOptions=qtyping.FullyConnectedOptionsT(fusedActivationFunction=0))
  post_code = transformation_utils.add_op_code(qtyping.BuiltinOperator.RESHAPE, model.operatorCodes, "RESHAPE")
  post_shape = transformation_utils.add_new_constant_tensor(
      name + b"_postrotate_shape", np.array(tensor.shape, np.int32), i32, sg, model)
  post_out = transformation_utils.add_new_activation_tensor(name + b"_postrotate_reshaped", tensor.shape, f32, sg)
  post = qtyping.OperatorT(opcodeIndex=post_code, inputs=[rotated, post_shape], outputs=[post_out])

  after_embedding = (ti.producer != -1
                     and _op_code_of(ti, ti.producer) == qtyping.BuiltinOperator.EMBEDDING_LOOKUP)
  if after_embedding:
    for consumer in ti.consumers:
      if consumer == -1:
        continue
      inputs = sg.operators[consumer].inputs
      for k, tid in enumerate(inputs):
        if tid == ti.tensor_id:
          inputs[k] = post_out
    for k, tid in enumerate(sg.outputs):
      if tid == ti.tensor_id:
        sg.outputs[k] = post_out
  else:
    updated = False
    for consumer in ti.consumers:
      if _op_code_of(ti, consumer) == qtyping.BuiltinOperator.FULLY_CONNECTED:
        sg.operators[consumer].inputs[0] = post_out
        updated = True
    if not updated:
      raise ValueError("The Hadamard rotation op supports embedding lookup and fully connected"
                       " ops only, but no such ops were found.")
  at = max(ti.producer + 1, min(ti.consumers))
  sg.operators[at:at] = [pre, fc, post]
  return qtyping.TransformationInfo(op_id=at, num_ops_added=3, output_tensor_id=post_out)


def insert_hadamard_rotation(ti: _Input) -> qtyping.TransformationInfo:
  """x -> CUSTOM "aeq.hadamard_rotation"(x) feeding the FULLY_CONNECTED consumers (or everything after
  an EMBEDDING_LOOKUP producer) whose weights were rotated; the op's options are the FlexBuffer map
  {hadamard_size, random_binary_vector} (ref insert_hadamard_rotation.py:24-156; the FlexBuffer
  itself: utils/flexbuffer.py -- the reference's encoder is a third-party package, its bytes unpinned)."""
  from ..utils import flexbuffer
  p = ti.quant_params
  if not isinstance(p, qtyping.UniformQuantParams):
    raise ValueError("Hadamard rotation supports uniform quantization only")
  if p.hadamard is None:
    raise ValueError("Hadamard rotation quantization params are not set but op insertion is"
                     " requested.")
  sg, model = ti.subgraph, ti.model
  tensor = sg.tensors[ti.tensor_id]
  if tensor.type != qtyping.TensorType.FLOAT32:
    raise ValueError(f"The Hadamard rotation op supports float32 tensors only. Got {tensor.type}"
                     " tensor.")
  code = transformation_utils.add_op_code(qtyping.BuiltinOperator.CUSTOM, model.operatorCodes,
                                          "aeq.hadamard_rotation")
  options = flexbuffer.encode_map({
      "hadamard_size": int(p.hadamard.hadamard_size),
      "random_binary_vector": np.asarray(p.hadamard.random_binary_vector).tolist(),
  })
  rotated = transformation_utils.add_new_activation_tensor(
      _raw_name(tensor) + b"_rotated",
      tensor.shapeSignature if tensor.shapeSignature is not None else tensor.shape,
      qtyping.TensorType.FLOAT32, sg)
  op = qtyping.OperatorT(opcodeIndex=code, inputs=[ti.tensor_id], outputs=[rotated],
                         customOptions=np.frombuffer(options, dtype=np.uint8))
  after_embedding = (ti.producer != -1
                     and _op_code_of(ti, ti.producer) == qtyping.BuiltinOperator.EMBEDDING_LOOKUP)
  if after_embedding:
    for consumer in ti.consumers:
      if consumer == -1:        # a graph output: handled below
        continue
      inputs = sg.operators[consumer].inputs
      for k, tid in enumerate(inputs):
        if tid == ti.tensor_id:
          inputs[k] = rotated
  else:
    updated = False
    for consumer in ti.consumers:
      if _op_code_of(ti, consumer) == qtyping.BuiltinOperator.FULLY_CONNECTED:
        sg.operators[consumer].inputs[0] = rotated
        updated = True
    if not updated:
      raise ValueError("The Hadamard rotation op supports embedding lookup and fully connected"
                       " ops only, but no such ops were found.")
  for k, tid in enumerate(sg.outputs):
    if tid == ti.tensor_id:
      sg.outputs[k] = rotated
  at = max(ti.producer + 1, min(ti.consumers))
  sg.operators.insert(at, op)
  return qtyping.TransformationInfo(op_id=at, num_ops_added=1, output_tensor_id=rotated)


def insert_multiply(ti: _Input) -> qtyping.TransformationInfo:
  """x -> MUL(x, multiplier) feeding the FULLY_CONNECTED consumers whose weight columns OSCAR
  scaled by 1/multiplier; the constant is shared between ops asking for the same vector
  (ref insert_multiply.py:24-129)."""
  p = ti.quant_params
  if not isinstance(p, qtyping.UniformQuantParams):
    raise ValueError("Insert multiply supports uniform quantization only.")
  if p.custom_algorithm_param is None or "multiplier" not in p.custom_algorithm_param:
    raise ValueError('Custom algorithm parameter "multiplier" is not set but multiply op'
                     " insertion is requested.")
  sg, model = ti.subgraph, ti.model
  tensor = sg.tensors[ti.tensor_id]
  if tensor.type != qtyping.TensorType.FLOAT32:
    raise ValueError(f"The insert multiply op supports float32 tensors only. Got {tensor.type}"
                     " tensor.")
  name = _raw_name(tensor)
  f32 = qtyping.TensorType.FLOAT32
  multiplier = transformation_utils.add_new_constant_tensor(
      name + b"_multiplier", np.asarray(p.custom_algorithm_param["multiplier"], dtype=np.float32),
      f32, sg, model, allow_tensor_sharing=True)
  shape = list(tensor.shapeSignature if tensor.shapeSignature is not None else tensor.shape)
  scaled = transformation_utils.add_new_activation_tensor(name + b"_scaled", shape, f32, sg)
  code = transformation_utils.add_op_code(qtyping.BuiltinOperator.MUL, model.operatorCodes, "MUL")
  mul = qtyping.OperatorT(
      opcodeIndex=code, inputs=[ti.tensor_id, multiplier], outputs=[scaled],
      builtinOptionsType=int(qtyping.BuiltinOptions.MulOptions),
      builtinOptions=qtyping.MulOptionsT(
          fusedActivationFunction=int(qtyping.ActivationFunctionType.NONE)))
  updated = False
  for consumer in ti.consumers:
    if _op_code_of(ti, consumer) == qtyping.BuiltinOperator.FULLY_CONNECTED:
      sg.operators[consumer].inputs[0] = scaled
      updated = True
  if not updated:
    raise ValueError("The insert multiply op supports fully connected consumers only, but no"
                     " such ops were found.")
  at = max(ti.producer + 1, min(ti.consumers))
  sg.operators.insert(at, mul)
  return qtyping.TransformationInfo(op_id=at, num_ops_added=1, output_tensor_id=scaled)
