"""INSERT_HADAMARD_ROTATION: the custom-op form of the rotation (ref transformations/insert_hadamard_rotation.py:24-156).
The expectations of the reference's own test (insert_hadamard_rotation_test.py:30-200) on its two fixture
models, the FlexBuffer the op carries (format by the published layout; the reference's encoder is the third-party
`flatbuffers` package, absent here: byte parity unpinned, stated in utils/flexbuffer.py), and a file round trip."""
import os

import numpy as np
import pytest

from mi355q import qtyping as q
from mi355q.transformations import graph_edits, transformation_utils
from mi355q.utils import flexbuffer, tfl_flatbuffer_utils, tflite_flatbuffer

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models")


def _params(vector=None, size=2):
  return q.UniformQuantParams(
      num_bits=8, quantized_dimension=None, scale=np.ones(1), zero_point=np.zeros(1),
      hadamard=q.UniformQuantParams.HadamardRotationParams(
          random_binary_vector=np.ones(1) if vector is None else vector, hadamard_size=size))


def _model(name):
  return tfl_flatbuffer_utils.read_model(os.path.join(MODELS, name))


def _input(model, tensor_id, producer, consumers, params):
  return transformation_utils.TransformationInput(tensor_id, model, model.subgraphs[0], producer, consumers, params)


def test_flexbuffer_known_answer_and_round_trips():
  """{hadamard_size: 2, random_binary_vector: [1, -1]} worked out by hand from the builder's steps: two keys,
  the untyped vector [2 | 1, -1 | INT, INT], the typed key vector [2 | 41, 28], the map
  [2, 1 | 2 | 2, 11 | INT, VECTOR], root offset 4, type MAP << 2, width 1 -- 53 bytes."""
  got = flexbuffer.encode_map({"hadamard_size": 2, "random_binary_vector": [1, -1]})
  want = (b"hadamard_size\x00" + b"random_binary_vector\x00" + bytes([2, 1, 0xFF, 4, 4]) + bytes([2, 41, 28])
          + bytes([2, 1, 2, 2, 11, 4, 40]) + bytes([4, 36, 1]))
  assert got == want and len(got) == 53
  assert flexbuffer.decode(got) == {"hadamard_size": 2, "random_binary_vector": [1, -1]}
  rng = np.random.default_rng(5)
  for size, n in ((2, 1), (64, 64), (2048, 2048), (16384, 16384), (1 << 20, 300)):
    vec = (rng.integers(0, 2, n) * 2 - 1).astype(np.int8).tolist()
    buf = flexbuffer.encode_map({"hadamard_size": size, "random_binary_vector": vec})
    assert flexbuffer.decode(buf) == {"hadamard_size": size, "random_binary_vector": vec}
    # a vector longer than 255 entries needs 2-byte slots: length, elements and offsets all widen together
    assert len(buf) < 3 * n + 80
  # what the reference's test passes (np.ones(1).tolist() is a float), wide ints, keys out of order
  assert flexbuffer.decode(flexbuffer.encode_map({"hadamard_size": 2, "random_binary_vector": [1.0]})) == {
      "hadamard_size": 2, "random_binary_vector": [1.0]}
  odd = {"z": -70000, "a": [0.1, 3, -129], "m": 2 ** 40}
  assert flexbuffer.decode(flexbuffer.encode_map(odd)) == odd and list(flexbuffer.decode(flexbuffer.encode_map(odd))) == ["a", "m", "z"]
  with pytest.raises(TypeError):
    flexbuffer.encode_map({"flag": True})
  with pytest.raises(ValueError):
    flexbuffer.decode(b"\x00")


def test_errors_of_the_reference():
  model = _model("single_fc_bias.tflite")
  with pytest.raises(ValueError, match="uniform quantization"):
    graph_edits.insert_hadamard_rotation(_input(model, 0, -1, [-1], q.NonLinearQuantParams(num_bits=16, quantized_data=None)))
  with pytest.raises(ValueError, match="quantization params are not set"):
    graph_edits.insert_hadamard_rotation(_input(model, 0, -1, [-1], q.UniformQuantParams(
        num_bits=8, quantized_dimension=None, scale=np.ones(1), zero_point=np.zeros(1))))
  model.subgraphs[0].tensors[0].type = q.TensorType.INT32
  with pytest.raises(ValueError, match="float32 tensors"):
    graph_edits.insert_hadamard_rotation(_input(model, 0, -1, [-1], _params()))


def test_custom_op_in_front_of_a_fully_connected():
  model = _model("single_fc_bias.tflite")
  info = graph_edits.insert_hadamard_rotation(_input(model, 0, -1, [-1], _params()))
  sg = model.subgraphs[0]
  assert (info.op_id, info.num_ops_added, info.output_tensor_id) == (0, 1, 4)
  assert len(sg.tensors) == 5 and len(model.operatorCodes) == 2
  assert model.operatorCodes[1].builtinCode == q.BuiltinOperator.CUSTOM
  assert model.operatorCodes[1].customCode in ("aeq.hadamard_rotation", b"aeq.hadamard_rotation")
  assert model.operatorCodes[sg.operators[0].opcodeIndex].builtinCode == q.BuiltinOperator.CUSTOM
  assert sg.operators[0].inputs[0] == 0 and sg.operators[1].inputs[0] == 4
  assert flexbuffer.decode(bytes(np.asarray(sg.operators[0].customOptions, np.uint8))) == {
      "hadamard_size": 2, "random_binary_vector": [1.0]}
  name = sg.tensors[4].name
  assert (name if isinstance(name, bytes) else name.encode()).endswith(b"_rotated")


def test_custom_op_behind_an_embedding_lookup_and_file_round_trip(tmp_path):
  model = _model("embedding_lookup.tflite")
  vec = np.array([1, -1, -1, 1], np.int8)
  info = graph_edits.insert_hadamard_rotation(_input(model, 2, 0, [-1], _params(vec, 4)))
  sg = model.subgraphs[0]
  assert (info.op_id, info.num_ops_added, info.output_tensor_id) == (1, 1, 3)
  assert len(sg.tensors) == 4 and len(model.operatorCodes) == 2
  assert model.operatorCodes[1].builtinCode == q.BuiltinOperator.CUSTOM
  assert model.operatorCodes[sg.operators[1].opcodeIndex].builtinCode == q.BuiltinOperator.CUSTOM
  assert sg.operators[1].inputs[0] == 2 and sg.operators[1].outputs[0] == 3
  assert list(sg.outputs) == [3]          # the rotated tensor is the graph output now
  # the op and its options survive the writer and the reader
  path = str(tmp_path / "rotated.tflite")
  open(path, "wb").write(tflite_flatbuffer.write_model(model))
  back = tfl_flatbuffer_utils.read_model(path)
  op = back.subgraphs[0].operators[1]
  assert back.operatorCodes[op.opcodeIndex].builtinCode == q.BuiltinOperator.CUSTOM
  assert flexbuffer.decode(bytes(np.asarray(op.customOptions, np.uint8))) == {
      "hadamard_size": 4, "random_binary_vector": [1, -1, -1, 1]}
