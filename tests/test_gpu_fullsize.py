"""Parity at (or near) BASELINE sizes: size-independent properties and oracle
comparisons that finish in seconds on the host."""
import warnings

import numpy as np
import pytest

import parity_rates

from golden_util import gen_c2
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import distributed, ops, qtyping
  from mi355q.algorithms.uniform_quantize import gptq, hadamard_rotation, naive_min_max_quantize, octav
  return types.SimpleNamespace(torch=torch, ops=ops, q=qtyping, gptq=gptq, had=hadamard_rotation,
                               mm=naive_min_max_quantize, octav=octav, dist=distributed)


def info_cfg(m, bits, gran, **algo):
  q = m.q
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran],
                                   algorithm_params=algo)
  return q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg)), cfg


def test_c2_requant_is_idempotent_on_dequantized_weights(m):
  """quantize(dequantize(q)) == q with the same scales: a size-independent property of a3."""
  w = gen_c2()
  x = m.torch.from_numpy(w).cuda()
  r = m.ops.requant_sym(x, 0, 8)
  deq = r["q"].to(m.torch.float32) * r["scale"].unsqueeze(1)
  q2 = m.ops.quantize(deq, 1, 4096, 4096, r["scale"], None, 8, True)
  assert m.torch.equal(q2, r["q"])
  # |w - deq| <= scale/2 (+ FP32 rounding of w/s and q*s) -- min/max scales never clip
  err = (x - deq).abs()
  assert bool((err <= r["scale"].unsqueeze(1) * 0.5001).all())


def test_c3_shape_blockwise_scales_bound_every_block(m):
  w = (np.random.default_rng(1003).standard_normal((1024, 11008), dtype=np.float32) * np.float32(0.02))
  x = m.torch.from_numpy(w).cuda()
  r = m.ops.requant_sym(x, 128, 4, want_packed=True, want_scale_f16=True)
  ref = O.min_max_quant_params(w, 4, True, "BLOCKWISE_128")
  assert np.array_equal(r["q"].cpu().numpy(), ref["quantized_data"])
  assert np.array_equal(r["scale"].cpu().numpy(), ref["scale"])
  q = r["q"].cpu().numpy()
  assert q.min() >= -8 and q.max() <= 7
  lo = (r["packed"].cpu().numpy() & 0xF).astype(np.int8)
  hi = (r["packed"].cpu().numpy() >> 4).astype(np.int8)
  unpacked = np.stack([lo, hi], 1).reshape(-1)
  unpacked = np.where(unpacked > 7, unpacked - 16, unpacked).astype(np.int8)
  assert np.array_equal(unpacked, q.reshape(-1))  # pack/unpack round trip


def test_c4_parity_subset_16_samples_through_gather_and_replay(m):
  """First 16 samples of the C4 workload (32 activation tensors of [1,256,4096], 4 MiB each, SURVEY
  8d's parity subset), sentinels planted as there; world_size 1 path of the multi-GPU layer."""
  rng = np.random.default_rng(44)
  names = [f"act{i}" for i in range(32)]
  samples = []
  for s in range(16):
    d = {}
    for i, n in enumerate(names):
      x = rng.standard_normal((1, 256, 4096), dtype=np.float32) * np.float32(1 + i / 8)
      if (s * 32 + i) % 97 == 0:
        x.reshape(-1)[:4] = [np.inf, -np.inf, 3.39e38, -3.39e38]
      d[n] = x
    samples.append(d)
  stats = m.dist.local_activation_stats(samples, names)
  assert stats.shape == (16, 32, 2)
  stats = m.dist.gather_sample_stats(stats)
  qsvs = m.dist.replay_qsv_updates(stats, names, [samples[0][n].shape for n in names])
  ref = {}
  for smp in samples:
    for n in names:
      r = O.activation_min_max(smp[n], -3e38, 3e38)
      ref[n] = O.moving_average_update(ref.get(n), r)
  for n in names:
    assert np.array_equal(qsvs[n]["min"], ref[n]["min"]) and np.array_equal(qsvs[n]["max"], ref[n]["max"])
  mm = m.dist.allreduce_min_max(stats)
  for i, n in enumerate(names):
    assert mm[i, 0] == min(O.activation_min_max(s[n], -3e38, 3e38)["min"].item() for s in samples)


def test_hadamard_octav_4096_against_oracle(m):
  """Hadamard(h=4096) + OCTAV int4 on 512 x 4096 (the C5 Hadamard shape, fewer rows so the
  NumPy oracle finishes in seconds). T2: scales 1e-6 rel, ints +-1 on a tiny fraction."""
  w = np.random.default_rng(55).standard_normal((512, 4096), dtype=np.float32)
  info, cfg = info_cfg(m, 4, "CHANNELWISE")
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.hadamard_quant_params(w, 4, "CHANNELWISE")
  p = m.had.get_tensor_quant_params(info, cfg, w)
  assert p.hadamard.hadamard_size == 4096
  np.testing.assert_allclose(p.scale, ref["scale"], rtol=1e-6)
  parity_rates.check("hadamard(h=4096)+octav int4 512x4096 vs oracle", p.quantized_data, ref["quantized_data"], parity_rates.T2)


@pytest.mark.parametrize("rows,cols,h", [(256, 2048, 2048), (64, 16384, 16384)])
def test_hadamard_octav_gemma_shapes_against_oracle(m, rows, cols, h):
  """The C5 Hadamard shapes: h = 2048 (q / k / v / o / gate / up rows of a Gemma-2B layer) and
  h = 16384 (down_proj rows): rotation + OCTAV + int4 against the oracle's sgemm rotation
  (ref hadamard_rotation.py:137-203). T2: scales 1e-6 rel, integers +-1 on <= 1e-5."""
  w = np.random.default_rng(57 + h).standard_normal((rows, cols), dtype=np.float32) * np.float32(0.02)
  info, cfg = info_cfg(m, 4, "CHANNELWISE")
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.hadamard_quant_params(w, 4, "CHANNELWISE")
  p = m.had.get_tensor_quant_params(info, cfg, w)
  assert p.hadamard.hadamard_size == h
  np.testing.assert_allclose(p.scale, ref["scale"], rtol=1e-6)
  parity_rates.check(f"hadamard(h={h})+octav int4 {rows}x{cols} vs oracle", p.quantized_data, ref["quantized_data"],
                     parity_rates.T2)


def test_octav_4096_rows_bit_exact(m):
  w = np.random.default_rng(56).standard_normal((256, 4096), dtype=np.float32)
  w[3] *= 40
  info, cfg = info_cfg(m, 4, "CHANNELWISE")
  ref = O.octav_quant_params(w, 4, "CHANNELWISE")
  p = m.octav.get_tensor_quant_params(info, cfg, w)
  assert np.array_equal(p.scale, ref["scale"]) and np.array_equal(p.quantized_data, ref["quantized_data"])


def test_gptq_gemma_attention_shape_against_oracle(m):
  """d = 2048 (Gemma-2B q/o projection), 256 output rows, Hessian from 4096 tokens."""
  rng = np.random.default_rng(57)
  d, rows = 2048, 256
  w = (rng.standard_normal((rows, d)) * 0.02).astype(np.float32)
  x = rng.standard_normal((8, 512, d)).astype(np.float32)
  x[..., :16] *= 6
  hg = m.gptq.hessian_of(x, np.array(8))
  h = O.gptq_hessian(x)
  assert np.max(np.abs(hg - h)) <= 3e-6 * np.abs(h).max()
  info, cfg = info_cfg(m, 4, "CHANNELWISE")
  ref = O.gptq_quant_params(w, 4, True, "CHANNELWISE",
                            {"activation_tensor_qsv": {"hessian": h, "num_samples": 8}})
  p = m.gptq.get_tensor_quant_params(info, cfg, w,
                                     {"activation_tensor_qsv": {"hessian": hg, "num_samples": 8}})
  assert np.array_equal(p.scale, ref["scale"])
  parity_rates.check("gptq end to end [256,2048] int4 (own Hessian + inverse) vs oracle", p.quantized_data,
                     ref["quantized_data"], parity_rates.T2)


@pytest.mark.parametrize("gran", ["CHANNELWISE", "BLOCKWISE_32"])
def test_oscar_layer_sized_weight_against_oracle(gran):
  """OSCAR at a layer-sized weight (2048 x 4096): 4096-element LDS sort tiles and the
  wave-per-row scan (channelwise), rank-major sorted blocks and the lane-per-block scan
  (blockwise), 8192-chunk pairwise sums over 2048 rows -- bit for bit against the oracle."""
  from mi355q import qtyping as q
  from mi355q.algorithms.uniform_quantize import oscar
  rng = np.random.default_rng(99)
  w = rng.standard_normal((2048, 4096)).astype(np.float32) * np.float32(0.02)
  w[:, :64] *= 12.0
  mu2 = np.exp(rng.normal(size=4096) * 1.5)
  cfg = q.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  res = oscar.get_tensor_quant_params(info, cfg, w, {"mu2": mu2})
  ref = O.oscar_quant_params(w, mu2, 4, gran)
  assert np.array_equal(res.custom_algorithm_param["multiplier"], ref["multiplier"])
  assert res.scale.dtype == ref["scale"].dtype and np.array_equal(res.scale, ref["scale"])
  assert np.array_equal(res.quantized_data, ref["quantized_data"])


@pytest.fixture(scope="module")
def embedding_table():
  """A Gemma-sized embedding table: 256000 x 2048 float32 = 2 097 152 000 B (byte offsets pass
  2^31 within the tensor)."""
  rng = np.random.default_rng(4242)
  base = rng.standard_normal((4000, 2048), dtype=np.float32) * np.float32(0.05)
  w = np.empty((256000, 2048), np.float32)
  for i in range(64):                     # 64 differently scaled copies: rows stay distinct in range
    np.multiply(base, np.float32(0.5 + i / 16), out=w[i * 4000:(i + 1) * 4000])
  w[-1, -1] = 3.0                         # the very last element must be seen
  return w


_ROW_SLICES = (slice(0, 96), slice(131000, 131200), slice(255904, 256000))


@pytest.mark.parametrize("alg,bits,gran", [
    ("min_max", 8, "CHANNELWISE"), ("min_max", 4, "BLOCKWISE_32"), ("min_max", 2, "BLOCKWISE_128"),
    ("octav", 4, "CHANNELWISE"), ("mse", 4, "CHANNELWISE")])
def test_embedding_table_over_2gib_matches_oracle_on_row_slices(m, embedding_table, alg, bits, gran):
  from mi355q import qtyping as q
  from mi355q.algorithms.uniform_quantize import mse
  w = embedding_table
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.EMBEDDING_LOOKUP, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  mod = {"min_max": m.mm, "octav": m.octav, "mse": mse}[alg]
  res = mod.get_tensor_quant_params(info, cfg, w)
  got_q = np.asarray(res.quantized_data)
  assert got_q.shape == w.shape and res.scale.shape[0] == w.shape[0]
  for rows in _ROW_SLICES:                # rows are independent groups for these granularities
    part = w[rows]
    if alg == "min_max":
      ref = O.min_max_quant_params(part, bits, True, gran, op="EMBEDDING_LOOKUP")
    elif alg == "octav":
      ref = O.octav_quant_params(part, bits, gran, op="EMBEDDING_LOOKUP")
    else:
      ref = O.mse_quant_params(part, bits, gran, op="EMBEDDING_LOOKUP")
    assert res.scale.dtype == ref["scale"].dtype
    assert np.array_equal(res.scale[rows], ref["scale"])
    assert np.array_equal(got_q[rows], ref["quantized_data"])
  if alg == "min_max" and gran == "CHANNELWISE":
    assert got_q[-1, -1] == 127 and abs(int(got_q[-1].astype(np.int32)[:-1].max())) < 127
  packed = getattr(res.quantized_data, "packed", None)
  if packed is not None:                  # sub-byte results ride with their packed bytes
    tail = np.asarray(packed).reshape(-1)[-(2048 * bits // 8):]
    assert np.array_equal(tail, O.pack_data(bits, got_q[-1]))
