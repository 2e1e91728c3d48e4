"""Seeded randomized sweep of the weight path against the oracle: shapes (2-D .. 4-D, ragged and
vector-unfriendly widths), ops, bit widths, granularities, symmetric / asymmetric, data with
outliers, zero channels, NaN and infinities. Everything is compared bit for bit."""
import warnings

import numpy as np
import pytest

from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu

OPS = {
    "FULLY_CONNECTED": lambda r: (int(r.integers(1, 70)), int(r.choice([1, 3, 32, 96, 127, 128, 260, 1024]))),
    "EMBEDDING_LOOKUP": lambda r: (int(r.integers(1, 50)), int(r.choice([32, 64, 96, 256]))),
    "CONV_2D": lambda r: (int(r.integers(1, 20)), int(r.integers(1, 4)), int(r.integers(1, 4)), int(r.integers(1, 9))),
    "DEPTHWISE_CONV_2D": lambda r: (1, int(r.integers(1, 4)), int(r.integers(1, 4)), int(r.integers(1, 40))),
    "CONV_2D_TRANSPOSE": lambda r: (int(r.integers(1, 12)), int(r.integers(1, 4)), int(r.integers(1, 4)), int(r.integers(1, 6))),
    "BATCH_MATMUL": lambda r: (int(r.integers(1, 4)), int(r.integers(1, 40)), int(r.integers(1, 40))),
}


def _data(rng, shape, kind):
  w = rng.standard_normal(shape).astype(np.float32)
  flat = w.reshape(-1)
  if kind == "outliers" and flat.size:
    flat[rng.integers(0, flat.size, max(1, flat.size // 200))] *= 50
  elif kind == "tiny":
    w *= np.float32(1e-12)
  elif kind == "zeros" and flat.size:
    w[tuple(slice(0, 1) for _ in shape)] = 0
    if w.ndim >= 2:
      w[0] = 0
  elif kind == "special" and flat.size >= 4:
    flat[rng.integers(0, flat.size, 3)] = [np.nan, np.inf, -np.inf]
  elif kind == "halves":                      # exact .5 quotients: rint ties
    w = (rng.integers(-9, 10, shape) * 0.5).astype(np.float32)
  return w


@pytest.mark.parametrize("seed", range(120))
def test_min_max_weight_path_random(seed):
  import __graft_entry__ as g
  g.build()
  from mi355q import qtyping as q
  from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as mm
  rng = np.random.default_rng(10_000 + seed)
  op = str(rng.choice(list(OPS)))
  shape = OPS[op](rng)
  bits = int(rng.choice([2, 4, 8]))
  sym = bool(rng.integers(0, 2)) or bits < 8          # asymmetric weights only at 8 bits here
  gran = str(rng.choice(["TENSORWISE", "CHANNELWISE", "BLOCKWISE_32"]))
  if gran.startswith("BLOCKWISE") and (op not in ("FULLY_CONNECTED", "EMBEDDING_LOOKUP") or shape[1] % 32):
    gran = "CHANNELWISE"
  if gran.startswith("BLOCKWISE"):
    sym = True
  adj_y = bool(rng.integers(0, 2)) if op == "BATCH_MATMUL" else False
  w = _data(rng, shape, str(rng.choice(["normal", "outliers", "tiny", "zeros", "special", "halves"])))
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=sym, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(builtinOptions=q.BatchMatMulOptionsT(adjY=adj_y)), op_name=q.TFLOperationName[op],
                  subgraph_op_index=0, op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    qdim = O.weight_quantized_dim(gran, op, w.ndim, adj_y)
    mmv = O.init_tensor_min_max(w, gran, qdim)
    zp, scale = O.zp_scale_from_min_max(mmv["min"], mmv["max"], bits, sym, gran, None)
    ref_q = O.uniform_quantize(w, scale, zp, bits, sym, quantized_dim=qdim, block_size=O.block_size_of(gran),
                               is_blockwise_quant=O.is_blockwise(gran))
    p = mm.get_tensor_quant_params(info, cfg, w)
  assert p.quantized_dimension == qdim
  assert p.scale.shape == scale.shape and np.array_equal(p.scale, scale, equal_nan=True), (op, shape, bits, gran)
  assert np.array_equal(p.zero_point, zp)
  assert p.quantized_data.shape == w.shape and np.array_equal(p.quantized_data, ref_q), (op, shape, bits, gran)


@pytest.mark.parametrize("seed", range(80))
def test_octav_and_mse_random(seed):
  """The order-exact reductions on random layouts: contiguous, channel-last and middle-axis
  units, rows longer than NumPy's 8192 buffer, dense and sparse selections."""
  import __graft_entry__ as g
  g.build()
  from mi355q import qtyping as q
  from mi355q.algorithms.uniform_quantize import mse, octav
  rng = np.random.default_rng(20_000 + seed)
  op = str(rng.choice(["FULLY_CONNECTED", "EMBEDDING_LOOKUP", "CONV_2D", "DEPTHWISE_CONV_2D", "BATCH_MATMUL"]))
  if op == "FULLY_CONNECTED":
    shape = (int(rng.integers(1, 12)), int(rng.choice([5, 64, 200, 1024, 4097, 9000, 17000])))
  elif op == "BATCH_MATMUL":
    shape = (int(rng.integers(1, 4)), int(rng.integers(1, 70)), int(rng.choice([3, 17, 64, 300])))
  else:
    shape = OPS[op](rng)
  bits = int(rng.choice([4, 8]))
  gran = "CHANNELWISE"
  if op in ("FULLY_CONNECTED", "EMBEDDING_LOOKUP") and shape[1] % 32 == 0 and rng.integers(0, 3) == 0:
    gran = "BLOCKWISE_32"
  adj_y = bool(rng.integers(0, 2)) if op == "BATCH_MATMUL" else False
  w = _data(rng, shape, str(rng.choice(["normal", "outliers", "zeros", "halves"])))
  w = w * np.float32(rng.choice([0.01, 1.0, 30.0]))           # dense vs sparse selections at guess = 1
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(builtinOptions=q.BatchMatMulOptionsT(adjY=adj_y)), op_name=q.TFLOperationName[op],
                  subgraph_op_index=0, op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.octav_quant_params(w, bits, gran, op=op, adj_y=adj_y)
    p = octav.get_tensor_quant_params(info, cfg, w)
    assert np.array_equal(p.scale, ref["scale"], equal_nan=True), (op, shape, bits, gran)
    assert np.array_equal(p.quantized_data, ref["quantized_data"]), (op, shape, bits, gran)
    if gran == "CHANNELWISE" and op != "BATCH_MATMUL":          # MSE is not registered for BMM
      ref = O.mse_quant_params(w, bits, gran, op=op)
      p = mse.get_tensor_quant_params(info, cfg, w, {"min": w.min(), "max": w.max()})
      assert np.array_equal(p.scale, ref["scale"], equal_nan=True), (op, shape, bits)
      assert np.array_equal(p.quantized_data, ref["quantized_data"]), (op, shape, bits)


@pytest.mark.parametrize("seed", range(30))
def test_activation_statistics_random(seed):
  import __graft_entry__ as g
  g.build()
  from mi355q.algorithms.uniform_quantize import common_quantize
  rng = np.random.default_rng(30_000 + seed)
  shape = tuple(int(x) for x in rng.integers(1, 40, int(rng.integers(1, 5))))
  x = rng.standard_normal(shape).astype(np.float32) * np.float32(rng.choice([1e-3, 1, 1e6]))
  flat = x.reshape(-1)
  k = str(rng.choice(["plain", "sentinels", "all_masked", "nan"]))
  if k == "sentinels":
    flat[rng.integers(0, flat.size, 3)] = [np.inf, -np.inf, 3.39e38]
  elif k == "all_masked":
    flat[:] = rng.choice([np.inf, -np.inf, 3.2e38], flat.size)
  elif k == "nan":
    flat[rng.integers(0, flat.size)] = np.nan
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.activation_min_max(x, -3e38, 3e38)
    got = common_quantize.get_activation_min_max(x, -3e38, 3e38)
  for key in ("min", "max"):
    assert got[key].shape == ref[key].shape and np.array_equal(got[key], ref[key], equal_nan=True), (k, shape)


# ------------------------------------------------------------------------------ OSCAR (f4) ---
@pytest.mark.parametrize("seed", range(60))
def test_fuzz_oscar_any_shape(seed):
  """FULLY_CONNECTED weights of random shape / granularity / bit width with random activation
  masses (dead channels, huge dynamic range, sometimes none at all): channel scales, FP64 or
  bf16-rounded scales, ints and the float32 multiplier equal the oracle's, bit for bit."""
  from mi355q.algorithms.uniform_quantize import oscar
  rng = np.random.default_rng(9000 + seed)
  gran = ["CHANNELWISE", "CHANNELWISE", "BLOCKWISE_32", "BLOCKWISE_64", "BLOCKWISE_128", "BLOCKWISE_256",
          "TENSORWISE"][int(rng.integers(0, 7))]
  bits = [4, 4, 8, 2][int(rng.integers(0, 4))]
  rows = int(rng.integers(1, 400))
  if gran.startswith("BLOCKWISE"):
    cols = int(gran.split("_")[1]) * int(rng.integers(1, 9))
  elif gran == "TENSORWISE":
    rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 300))
  else:
    cols = int(rng.integers(1, 2500))
  w = rng.standard_normal((rows, cols)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 2))
  style = int(rng.integers(0, 5))
  if style == 0:
    w[:, : max(1, cols // 10)] *= 30.0
  elif style == 1:
    w[rng.random((rows, cols)) < 0.3] = 0.0
  elif style == 2:
    w = (np.round(w / np.abs(w).max() * 6) / 6).astype(np.float32)          # heavy ties
  mu2 = np.exp(rng.normal(size=cols) * rng.uniform(0.1, 3.0))
  if style == 2 or rng.random() < 0.15:
    mu2 = None                          # ties + non-uniform masses depend on NumPy's unstable argsort
  elif rng.random() < 0.3:
    mu2[rng.random(cols) < 0.2] = 0.0
  from mi355q import qtyping as q
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  res = oscar.get_tensor_quant_params(info, cfg, w, None if mu2 is None else {"mu2": mu2})
  ref = O.oscar_quant_params(w, mu2, bits, gran)
  assert np.array_equal(res.custom_algorithm_param["multiplier"], ref["multiplier"])
  assert res.scale.dtype == ref["scale"].dtype and np.array_equal(res.scale, ref["scale"])
  assert np.array_equal(res.quantized_data, ref["quantized_data"])
