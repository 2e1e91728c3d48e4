"""Model-level batching of the per-op weight loop (mi355q/requant_queue.py; ref
params_generator.py:110-183): what leaves through one batched launch per shape group must be
the bytes the one-launch-per-tensor path produces."""
import hashlib
import os
import sys

import numpy as np
import pytest

from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import qtyping, quantizer, recipe, requant_queue, runtime
  from mi355q.algorithms.uniform_quantize import naive_min_max_quantize
  return types.SimpleNamespace(torch=torch, q=qtyping, mm=naive_min_max_quantize, rq=requant_queue,
                               rt=runtime, quantizer=quantizer, recipe=recipe)


def _info_cfg(m, bits, gran, op="FULLY_CONNECTED"):
  q = m.q
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  return q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName[op], subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg)), cfg


CASES = [  # (shape, bits, granularity): several shape groups, small and large, ragged widths
    ((512, 1024), 8, "CHANNELWISE"), ((512, 1024), 8, "CHANNELWISE"), ((512, 1024), 4, "CHANNELWISE"),
    ((2048, 2048), 4, "BLOCKWISE_128"), ((2048, 2048), 4, "BLOCKWISE_128"), ((2048, 2048), 4, "BLOCKWISE_32"),
    ((300, 260), 8, "CHANNELWISE"), ((300, 256), 2, "BLOCKWISE_32"),
    ((257, 1001), 4, "CHANNELWISE"), ((257, 1001), 4, "CHANNELWISE"), ((1024, 4096), 2, "CHANNELWISE"),
    ((4096, 4096), 8, "CHANNELWISE"), ((64, 64), 8, "CHANNELWISE")]


def test_queued_results_equal_immediate_results_and_oracle(m):
  rng = np.random.default_rng(2024)
  ws = [rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05 + i) for i, (shape, _, _) in enumerate(CASES)]
  ws[1][3] = 0.0                                   # an all-zero row
  ws[3][5, 7] = np.nan                             # NaN propagates into that block's scale
  immediate = []
  for w, (_, bits, gran) in zip(ws, CASES):
    info, cfg = _info_cfg(m, bits, gran)
    immediate.append(m.mm.get_tensor_quant_params(info, cfg, w))
  with m.rq.batching() as queue:
    queued = []
    for w, (_, bits, gran) in zip(ws, CASES):
      info, cfg = _info_cfg(m, bits, gran)
      queued.append(m.mm.get_tensor_quant_params(info, cfg, w))
    assert queue.stats["launches"] == 0            # nothing has run yet
    assert isinstance(queued[0].scale, m.rq.PendingArray) and queued[0].scale.shape == (512, 1)
  groups = len({(c[0], c[1], c[2]) for c in CASES if c[0] != (64, 64)})
  assert queue.stats["launches"] == groups and queue.stats["tensors"] == len(CASES) - 1
  assert queue.stats["scale_copies"] == 1
  for a, b, w, (_, bits, gran) in zip(immediate, queued, ws, CASES):
    if "BLOCKWISE" in gran and w.size * 4 >= 64 << 10:   # blockwise scales stay in HBM, f16 patterns beside them
      assert isinstance(b.scale, m.rt.HbmArray)
      assert np.array_equal(np.asarray(b.scale.f16).view(np.uint16), O.blockwise_scale_f16(np.asarray(a.scale)).view(np.uint16),
                            ) or np.isnan(w).any()
    else:
      assert type(b.scale) is np.ndarray
    assert b.scale.dtype == np.float32 and b.scale.shape == a.scale.shape
    assert np.array_equal(a.scale, b.scale, equal_nan=True)
    assert np.array_equal(a.zero_point, b.zero_point) and b.zero_point.dtype == np.int8
    qa, qb = np.asarray(a.quantized_data), np.asarray(b.quantized_data)
    assert qb.dtype == np.int8 and qb.shape == w.shape and np.array_equal(qa, qb)
    pa, pb = getattr(a.quantized_data, "packed", None), getattr(b.quantized_data, "packed", None)
    assert (pa is None) == (pb is None)
    if pb is not None:
      assert np.array_equal(np.asarray(pa), np.asarray(pb))
      assert np.array_equal(np.asarray(pb), O.pack_data(bits, qb.reshape(-1).view(np.uint8)))
    if not np.isnan(w).any():
      assert a == b                                 # value equality of the parameter records
      ref = O.min_max_quant_params(w, bits, True, gran)
      assert np.array_equal(b.scale, ref["scale"]) and np.array_equal(qb, ref["quantized_data"])


def test_reading_a_pending_value_flushes_and_budget_flushes_in_waves(m):
  rng = np.random.default_rng(5)
  info, cfg = _info_cfg(m, 8, "CHANNELWISE")
  ws = [rng.standard_normal((256, 512), dtype=np.float32) for _ in range(7)]
  with m.rq.batching(budget_tensors=3) as queue:
    ps = [m.mm.get_tensor_quant_params(info, cfg, w) for w in ws]
    assert queue.stats["flushes"] == 2 and queue.stats["launches"] == 2      # two waves of three
    assert isinstance(ps[0].scale, m.rq.PendingArray) and ps[0].scale.resolved and not ps[6].scale.resolved
    s6 = np.asarray(ps[6].scale)                                             # a read flushes the rest
    assert queue.stats["flushes"] == 3 and ps[6].scale.resolved
    assert np.array_equal(s6, O.min_max_quant_params(ws[6], 8, True, "CHANNELWISE")["scale"])
  assert queue.stats["flushes"] == 3
  assert all(type(p.scale) is np.ndarray for p in ps)                        # swapped in on exit
  for p, w in zip(ps, ws):
    assert np.array_equal(np.asarray(p.quantized_data), O.min_max_quant_params(w, 8, True, "CHANNELWISE")["quantized_data"])


def test_hbm_resident_weights_and_lazily_unpacked_int4(m):
  """Inputs that already live in HBM are not copied; an int4 group writes only packed bytes and
  the int8 containers appear when asked for."""
  torch = m.torch
  rng = np.random.default_rng(6)
  info, cfg = _info_cfg(m, 4, "BLOCKWISE_128")
  ws = [rng.standard_normal((1024, 2048), dtype=np.float32) * np.float32(0.02) for _ in range(4)]
  dev = [m.rt.HbmArray(torch.from_numpy(w).cuda()) for w in ws]
  with m.rq.batching() as queue:
    ps = [m.mm.get_tensor_quant_params(info, cfg, d) for d in dev]
  assert queue.stats["launches"] == 1
  for p, w in zip(ps, ws):
    ref = O.min_max_quant_params(w, 4, True, "BLOCKWISE_128")
    qd = p.quantized_data
    assert isinstance(qd, m.rq.PendingArray) and not qd.resolved and qd.packed.resolved
    assert isinstance(p.scale, m.rq.PendingArray) and p.scale._host is None  # nothing came to the host
    assert np.array_equal(np.asarray(qd.packed), O.pack_data(4, ref["quantized_data"].reshape(-1).view(np.uint8)))
    assert not qd.resolved                                                   # still only the packed bytes
    assert np.array_equal(np.asarray(qd), ref["quantized_data"]) and qd.resolved
    assert np.array_equal(p.scale, ref["scale"])


def _sha(path):
  h = hashlib.sha256()
  with open(path, "rb") as f:
    for chunk in iter(lambda: f.read(1 << 24), b""):
      h.update(chunk)
  return h.hexdigest()


def test_c3_model_32_layers_two_launches_and_identical_file(m, tmp_path):
  """BASELINE C3 as a model file: 32 FULLY_CONNECTED layers of 4096 x 11008 FP32 through
  Quantizer.quantize(dynamic int4 blockwise-128). The file equals, byte for byte, the one the launch-per-tensor
  path writes. Launches: the weights of a 5.8 GB file arrive through the upload ring while the writer asks for the
  payloads in file order, so a launch takes the tensors that have arrived (at most one launch per tensor; round 3
  waited for all uploads and needed two); resident weights of one shape still leave 16 at a time
  (test_hbm_resident_weights_and_lazily_unpacked_int4, bench.py api_resident)."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import file_bench
  src = str(tmp_path / "c3.tflite")
  file_bench.build_model(src, 32, 4096, 11008)
  rcp = m.recipe.dynamic_wi4b128_afp32()
  outs = []
  slices = []
  arena_slice = m.rt._UploadArena.slice

  def counted(self, offset, n):
    out = arena_slice(self, offset, n)
    slices.append(out is not None)
    return out
  m.rt._UploadArena.slice = counted
  for enabled in (True, False):
    m.rq.ENABLED = enabled
    try:
      dst = str(tmp_path / f"c3_{int(enabled)}.tflite")
      qz = m.quantizer.Quantizer(src, rcp)
      qz.quantize(serialize_to_path=dst)
      outs.append((dst, qz.batch_stats))
    finally:
      m.rq.ENABLED = True
  m.rt._UploadArena.slice = arena_slice
  # the 32 announced weights of each run landed in ONE allocation made by a helper thread (runtime._UploadArena)
  assert len(slices) == 64 and all(slices), slices
  (batched, stats), (single, stats_off) = outs
  assert stats["tensors"] == 32 and stats["launches"] <= 32, stats
  assert stats_off["tensors"] == 0
  assert os.path.getsize(batched) == os.path.getsize(single) > 32 * 4096 * 11008 // 2
  assert _sha(batched) == _sha(single)
  # and the bytes are the reference's: layer 0 against the oracle
  from mi355q.utils import tfl_flatbuffer_utils
  model = tfl_flatbuffer_utils.read_model(tfl_flatbuffer_utils.get_model_content(batched))
  w0 = np.random.default_rng(0).standard_normal((4096, 11008), dtype=np.float32) * np.float32(0.02)
  ref = O.min_max_quant_params(w0, 4, True, "BLOCKWISE_128")
  t = next(t for t in model.subgraphs[0].tensors if t.name in (b"w0", "w0"))
  assert np.array_equal(np.asarray(model.buffers[t.buffer].data),
                        O.pack_data(4, ref["quantized_data"].reshape(-1).view(np.uint8)))
