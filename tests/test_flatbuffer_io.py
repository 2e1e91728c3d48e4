"""The self-contained TFLite flatbuffer reader / writer (CPU; no GPU, no third-party wheel).

Fixtures under tests/golden/models/ are copies of the reference's test-model data files.
"""
import glob
import os
import struct

import numpy as np
import pytest

from mi355q import qtyping as q
from mi355q.utils import tfl_flatbuffer_utils
from mi355q.utils import tflite_flatbuffer as fb

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = sorted(glob.glob(os.path.join(HERE, "golden", "models", "*.tflite")))


def same(a, b, path="model"):
  if isinstance(a, fb.TableT):
    assert type(a) is type(b), path
    for spec in fb.SCHEMA[a._table]:
      if spec[1] != "dead":
        same(getattr(a, spec[0]), getattr(b, spec[0]), f"{path}.{spec[0]}")
  elif isinstance(a, list):
    assert isinstance(b, list) and len(a) == len(b), path
    for i, (x, y) in enumerate(zip(a, b)):
      same(x, y, f"{path}[{i}]")
  elif isinstance(a, np.ndarray):
    assert isinstance(b, np.ndarray) and a.dtype == b.dtype and a.tobytes() == b.tobytes(), path
  else:
    assert a == b, (path, a, b)


def test_fixture_models_present():
  assert len(MODELS) >= 12


@pytest.mark.parametrize("path", MODELS, ids=[os.path.basename(p)[:-7] for p in MODELS])
def test_read_accounts_for_every_byte_and_round_trips(path):
  data = open(path, "rb").read()
  m = fb.read_model(data)                      # verify=True: 100 % byte coverage or it raises
  assert m.version == 3 and m.subgraphs and m.buffers
  out = fb.write_model(m)
  assert bytes(out[4:8]) == b"TFL3"
  m2 = fb.read_model(out)
  same(m, m2)
  assert fb.write_model(m2) == out             # writer is a fixed point
  ext = fb.serialize_with_external_buffers(m)
  same(m, fb.read_model(ext))
  # constants are zero-copy views of the input bytes
  big = [b for b in m.buffers if b.data is not None and b.data.size]
  for b in big:
    assert not b.data.flags.owndata and not b.data.flags.writeable


def test_known_model_content():
  m = tfl_flatbuffer_utils.read_model(os.path.join(HERE, "golden", "models", "conv_fc_mnist.tflite"))
  codes = [c.builtinCode for c in m.operatorCodes]
  B = q.BuiltinOperator
  assert codes == [B.CONV_2D, B.AVERAGE_POOL_2D, B.RESHAPE, B.FULLY_CONNECTED, B.SOFTMAX]
  sg = m.subgraphs[0]
  assert isinstance(sg.inputs, list) and isinstance(sg.operators[0].inputs, list)
  fc = [op for op in sg.operators if m.operatorCodes[op.opcodeIndex].builtinCode == B.FULLY_CONNECTED][0]
  assert isinstance(fc.builtinOptions, q.FullyConnectedOptionsT)
  w = tfl_flatbuffer_utils.get_tensor_data(sg.tensors[fc.inputs[1]], m.buffers)
  assert w.dtype == np.float32 and list(w.shape) == list(sg.tensors[fc.inputs[1]].shape)
  assert m.signatureDefs and m.signatureDefs[0].signatureKey == b"serving_default"
  # an already blockwise-quantized model: details union + side tensors survive
  m = tfl_flatbuffer_utils.read_model(
      os.path.join(HERE, "golden", "models", "single_fc_bias_sub_channel_weight_only_sym_weight.tflite"))
  qs = [t.quantization for t in m.subgraphs[0].tensors if t.quantization is not None and t.quantization.scale is not None]
  assert qs
  # composite op: typed BuiltinOptions2 with a string and a byte vector
  m = tfl_flatbuffer_utils.read_model(os.path.join(HERE, "golden", "models", "simple_composite.tflite"))
  comp = [op for sg in m.subgraphs for op in sg.operators if op.builtinOptions2 is not None][0]
  assert comp.builtinOptions2Type == 21 and comp.builtinOptions2.name and comp.builtinOptions2.compositeAttributes.size


def test_build_from_scratch_and_external_layout():
  w = np.arange(64 * 96, dtype=np.float32).reshape(64, 96)
  model = q.ModelT(version=3, description=b"scratch")
  model.buffers = [q.BufferT(), q.BufferT(data=w.view(np.uint8).reshape(-1)), q.BufferT(data=np.arange(7, dtype=np.uint8))]
  sg = q.SubGraphT(name=b"main", inputs=[0], outputs=[2])
  qp = q.QuantizationParametersT(scale=np.array([0.5, 0.25], np.float32), zeroPoint=np.array([0, -3], np.int64),
                                 quantizedDimension=1, detailsType=2,
                                 details=q.BlockwiseQuantizationT(scales=3, zeroPoints=-1, blockSize=32))
  sg.tensors = [q.TensorT(name=b"x", shape=[1, 96], buffer=0), q.TensorT(name=b"w", shape=[64, 96], buffer=1),
                q.TensorT(name=b"y", shape=[1, 64], buffer=0, shapeSignature=[-1, 64], hasRank=True),
                q.TensorT(name=b"s", shape=[7], buffer=2, type=int(q.TensorType.UINT8), quantization=qp)]
  sg.operators = [q.OperatorT(inputs=[0, 1, -1], outputs=[2], builtinOptionsType=8,
                              builtinOptions=q.FullyConnectedOptionsT(fusedActivationFunction=1))]
  model.operatorCodes = [q.OperatorCodeT(builtinCode=9, deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serve", subgraphIndex=0,
                                         inputs=[q.TensorMapT(name=b"x", tensorIndex=0)],
                                         outputs=[q.TensorMapT(name=b"y", tensorIndex=2)])]
  inline = fb.write_model(model)
  m = fb.read_model(inline)
  same(model.subgraphs[0].tensors[3].quantization, m.subgraphs[0].tensors[3].quantization)
  assert m.subgraphs[0].tensors[2].shapeSignature == [-1, 64] and m.subgraphs[0].tensors[2].hasRank is True
  assert np.array_equal(m.buffers[1].data.view(np.float32).reshape(64, 96), w)
  assert m.subgraphs[0].operators[0].builtinOptions.fusedActivationFunction == 1
  assert m.signatureDefs[0].outputs[0].tensorIndex == 2
  # the inline weight vector is 16-byte aligned (schema: force_align 16)
  assert inline.find(w.tobytes()) % 16 == 0
  # external layout: weights behind the flatbuffer, 16-byte aligned, recorded in offset/size
  ext = fb.serialize_with_external_buffers(model, min_size_bytes=1024)
  pos = bytes(ext).find(w.tobytes())
  assert pos % 16 == 0 and len(ext) < len(inline) + 64
  raw = fb._Reader(ext)
  root = raw.table(struct.unpack_from("<I", ext, 0)[0], "Model")
  assert (root.buffers[1].offset, root.buffers[1].size) == (pos, w.nbytes) and root.buffers[1].data is None
  assert root.buffers[2].data is not None and root.buffers[2].offset == 0       # small buffers stay inline
  m2 = fb.read_model(ext)
  assert np.array_equal(m2.buffers[1].data.view(np.float32).reshape(64, 96), w) and m2.buffers[1].offset == 0
  assert model.buffers[1].offset == 0 and model.buffers[1].data is not None     # input tree restored


def test_rejects_what_it_cannot_carry():
  data = bytearray(open(os.path.join(HERE, "golden", "models", "single_fc.tflite"), "rb").read())
  with pytest.raises(fb.FlatbufferError):
    fb.read_model(data[:40])
  with pytest.raises(fb.FlatbufferError):
    fb.read_model(b"\0" * 4)
  # a table with a field this module's schema does not know (newer schema) is refused loudly
  m = fb.read_model(bytes(data))
  extra = fb.SCHEMA["Tensor"]
  fb.SCHEMA["Tensor"] = extra[:3]              # pretend we only know shape/type/buffer
  try:
    with pytest.raises(fb.FlatbufferError, match="unknown fields"):
      fb.read_model(bytes(data))
  finally:
    fb.SCHEMA["Tensor"] = extra
  # an opaque option table that hides an offset leaves bytes unaccounted for -> refused
  saved = dict(fb.SCHEMA["Operator"][4][1][2])
  reshape = os.path.join(HERE, "golden", "models", "reshape_with_empty_shape.tflite")
  src = open(reshape, "rb").read()
  has_vec = any(isinstance(op.builtinOptions, fb.ReshapeOptionsT) and op.builtinOptions.newShape is not None
                for sg in fb.read_model(src).subgraphs for op in sg.operators)
  if has_vec:
    del fb.SCHEMA["Operator"][4][1][2][17]     # treat ReshapeOptions as opaque
    try:
      with pytest.raises(fb.FlatbufferError, match="unaccounted"):
        fb.read_model(src)
    finally:
      fb.SCHEMA["Operator"][4][1][2].update(saved)
  with pytest.raises(ValueError, match="Unsupported tflite_model type"):
    tfl_flatbuffer_utils.read_model(1234)


def test_constant_buffer_sharing_follows_reference_rule():
  from mi355q.transformations import transformation_utils as tu
  model = q.ModelT(version=3)
  a = np.arange(40, dtype=np.uint8)
  model.buffers = [q.BufferT(), q.BufferT(data=a.copy()), q.BufferT(data=a.copy()), q.BufferT(data=a[:39].copy())]
  sg = q.SubGraphT(tensors=[])
  assert tu.get_constant_buffer(a.copy(), model) == 2          # later equal buffer shadows the earlier one
  assert tu.get_constant_buffer(a[:39].copy(), model) == 3
  new = tu.get_constant_buffer(np.arange(5, dtype=np.float32), model)
  assert new == 4 and len(model.buffers) == 5
  assert tu.get_constant_buffer(np.arange(5, dtype=np.float32), model) == 4   # found again, not re-added
  assert tu.get_constant_buffer(np.arange(5, dtype=np.float32), model, force_duplicate_buffer=True) == 5
  tid = tu.add_new_constant_tensor(b"t", np.arange(5, dtype=np.float32), q.TensorType.FLOAT32, sg, model)
  assert sg.tensors[tid].buffer == 5 and sg.tensors[tid].shape == [5]


def test_corrupted_files_are_refused_cleanly():
  """Byte flips, truncation and random 32-bit words: the reader either parses (the damage hit
  payload bytes or padding) or raises FlatbufferError - never another exception, never a hang
  (uoffsets only point forward, every access is bounds-checked first)."""
  rng = np.random.default_rng(0)
  small = [p for p in MODELS if os.path.getsize(p) < 6000]
  outcomes = {"ok": 0, "refused": 0}
  for it in range(1500):
    data = bytearray(open(small[it % len(small)], "rb").read())
    mode = it % 3
    if mode == 0:
      for _ in range(int(rng.integers(1, 4))):
        data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    elif mode == 1:
      data = data[:int(rng.integers(0, len(data)))]
    else:
      i = int(rng.integers(0, len(data) - 4))
      data[i:i + 4] = int(rng.integers(0, 2**32)).to_bytes(4, "little")
    try:
      m = fb.read_model(bytes(data))
      fb.write_model(m)
      outcomes["ok"] += 1
    except fb.FlatbufferError:
      outcomes["refused"] += 1
  assert outcomes["refused"] > 500 and outcomes["ok"] > 100
