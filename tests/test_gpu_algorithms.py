"""GPU parity of the drop-in layer: get_tensor_quant_params of every algorithm,
called exactly as the reference's ParamsGenerator calls it, against the fixtures
recorded from the real reference and against the oracle."""
import warnings

import numpy as np
import pytest

import parity_rates

from golden_util import case_names, sha
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import qtyping
  from mi355q.algorithms.uniform_quantize import (hadamard_rotation, mse,
                                                 naive_min_max_quantize, octav,
                                                 uniform_quantize_tensor)
  return types.SimpleNamespace(qtyping=qtyping, mm=naive_min_max_quantize, octav=octav, mse=mse,
                               had=hadamard_rotation, uqt=uniform_quantize_tensor)


def op_info(m, op, cfg):
  q = m.qtyping
  return q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName[op], subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))


def cfg_of(m, c, **algo):
  q = m.qtyping
  return q.TensorQuantizationConfig(num_bits=c["num_bits"], symmetric=c["symmetric"],
                                    granularity=q.QuantGranularity[c["granularity"]],
                                    algorithm_params=algo)


def check(arrays, name, p, c, exact_q=True):
  assert p.quantized_dimension == c["quantized_dimension"]
  assert p.block_size == c["block_size"]
  assert p.num_bits == c["num_bits"] and p.symmetric == c["symmetric"]
  s, z = arrays[f"{name}/scale"], arrays[f"{name}/zero_point"]
  assert p.scale.shape == s.shape and p.scale.dtype == s.dtype
  assert np.array_equal(p.scale, s, equal_nan=True)
  assert p.zero_point.shape == z.shape and np.array_equal(p.zero_point, z)
  if f"{name}/q" in arrays and exact_q:
    q = arrays[f"{name}/q"]
    assert p.quantized_data.dtype == q.dtype and p.quantized_data.shape == q.shape
    assert np.array_equal(p.quantized_data, q)


@pytest.mark.parametrize("name", case_names("min_max"))
def test_min_max_get_tensor_quant_params(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  cfg = cfg_of(m, c)
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = m.mm.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, arrays[f"{name}/w"])
  check(arrays, name, p, c)
  assert p.zero_point.dtype == arrays[f"{name}/zero_point"].dtype


@pytest.mark.parametrize("name", case_names("min_max_qsv"))
def test_min_max_params_from_qsv(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  cfg = cfg_of(m, c)
  p = m.mm.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, None,
                                   {"min": arrays[f"{name}/min"], "max": arrays[f"{name}/max"]})
  assert p.quantized_data is None
  check(arrays, name, p, c)


def test_min_max_missing_stats_errors(m):
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=8)
  with pytest.raises(ValueError, match="not found in tensor_name_to_qsv"):
    m.mm.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg, None, None)
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED,
                  subgraph_op_index=0, op_quant_config=q.OpQuantizationConfig())
  with pytest.raises(ValueError, match="min and max must be provided"):
    m.mm.get_tensor_quant_params(info, cfg, np.ones((4, 4), np.float32), None)
  cfgb = q.TensorQuantizationConfig(num_bits=4, granularity=q.QuantGranularity.BLOCKWISE_32)
  with pytest.raises(ValueError, match="is not divisible by block size"):
    m.mm.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfgb), cfgb,
                                 np.ones((4, 33), np.float32))


@pytest.mark.parametrize("name", case_names("octav"))
def test_octav_get_tensor_quant_params_bit_exact(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  cfg = cfg_of(m, c)
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = m.octav.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, arrays[f"{name}/w"])
  check(arrays, name, p, c)


@pytest.mark.parametrize("name", case_names("octav"))
def test_octav_clipping_constants_bit_exact(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  w = arrays[f"{name}/w"]
  gran = c["granularity"]
  if "BLOCKWISE" in gran:
    b = c["block_size"]
    data, axis = w.reshape(w.shape[0], w.shape[1] // b, b), 2
  elif gran == "CHANNELWISE":
    data, axis = w, (1,)
  else:
    data, axis = w, None
  for early, key in ((True, "clip"), (False, "clip_no_early_stop")):
    got = m.octav._guess_clipping_with_octav(data, c["num_bits"], axis, 10, 3.0, early_stop=early)
    ref = arrays[f"{name}/{key}"]
    assert got.shape == ref.shape and np.array_equal(got, ref), key


def test_octav_dense_and_degenerate_rows_bit_exact(m):
  """Long runs (all-positive rows, constant rows, c -> 0) exercise the pairwise paths."""
  rng = np.random.default_rng(3)
  rows = [np.abs(rng.standard_normal(4096)).astype(np.float32) + 1.5,   # every element selected
          np.full(4096, 0.75, np.float32), rng.standard_normal(4096).astype(np.float32) * 1e-3,
          -np.abs(rng.standard_normal(4096)).astype(np.float32) * 3,
          rng.standard_normal(4096).astype(np.float32)]
  w = np.stack(rows)
  for bits in (4, 8):
    ref = O.octav_clip(w, bits, (1,), 10, 3.0)
    got = m.octav._guess_clipping_with_octav(w, bits, (1,), 10, 3.0)
    assert np.array_equal(got, ref)
  for n in (9000, 11008, 16384, 20000):  # rows longer than NumPy's 8192-element buffer
    w = np.abs(rng.standard_normal((3, n))).astype(np.float32)
    assert np.array_equal(m.octav._guess_clipping_with_octav(w, 4, (1,), 10, 3.0),
                          O.octav_clip(w, 4, (1,), 10, 3.0))


@pytest.mark.parametrize("shape,op,adj_y", [
    ((1, 3, 3, 64), "DEPTHWISE_CONV_2D", False), ((1, 5, 5, 700), "DEPTHWISE_CONV_2D", False),
    ((96, 130), "BATCH_MATMUL", False), ((4, 33, 257), "BATCH_MATMUL", False),
    ((5000, 8), "BATCH_MATMUL", False),
    ((2, 16, 8), "BATCH_MATMUL", True), ((3, 40, 300), "BATCH_MATMUL", True),
    ((2, 5, 9000), "BATCH_MATMUL", True), ((7, 130, 5), "BATCH_MATMUL", True)])
def test_octav_and_mse_strided_units_bit_exact(m, shape, op, adj_y):
  """Quantized dimension = last or a middle axis: NumPy keeps one running total per channel
  and adds the segments x[o, c, :] to it in order; the column / segment kernels keep that
  order (oracle = the reference's own np.sum calls)."""
  q = m.qtyping
  rng = np.random.default_rng(sum(shape))
  w = rng.standard_normal(shape).astype(np.float32)
  qd = len(shape) - 2 if adj_y else len(shape) - 1
  idx = [slice(None)] * len(shape)
  idx[qd] = 1
  w[tuple(idx)] *= 40.0                  # a channel with outliers
  idx[qd] = 2
  w[tuple(idx)] = 0.0                    # an all-zero channel
  ax = tuple(d for d in range(len(shape)) if d != qd)
  for bits in (4, 8):
    for early in (True, False):
      ref = O.octav_clip(w, bits, ax, 10, 3.0, early_stop=early)
      got = m.octav._guess_clipping_with_octav(w, bits, ax, 10, 3.0, early_stop=early)
      assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True)
    cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity.CHANNELWISE)
    info = q.OpInfo(op=q.OperatorT(builtinOptions=q.BatchMatMulOptionsT(adjY=adj_y)),
                    op_name=q.TFLOperationName[op], subgraph_op_index=0,
                    op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
    for mod, ref in ((m.octav, O.octav_quant_params(w, bits, "CHANNELWISE", op=op, adj_y=adj_y)),
                     (m.mse, O.mse_quant_params(w, bits, "CHANNELWISE", op=op, adj_y=adj_y))):
      p = mod.get_tensor_quant_params(info, cfg, w, {"min": w.min(), "max": w.max()})
      assert p.quantized_dimension == qd == ref["quantized_dimension"]
      assert np.array_equal(p.scale, ref["scale"], equal_nan=True)
      assert np.array_equal(p.quantized_data, ref["quantized_data"])


def test_octav_anchor_digest(m, ref_digests):
  d = ref_digests["octav_anchor"]
  w = np.random.default_rng(d["seed"]).standard_normal(tuple(d["shape"]), dtype=np.float32)
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=4, granularity=q.QuantGranularity.CHANNELWISE)
  p = m.octav.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg, w)
  assert sha(p.quantized_data) == d["q"] and sha(p.scale) == d["scale"]


def test_octav_errors(m):
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=4, symmetric=False,
                                   granularity=q.QuantGranularity.CHANNELWISE)
  with pytest.raises(ValueError, match="Unsupported symmetry"):
    m.octav.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg,
                                    np.ones((4, 4), np.float32))


@pytest.mark.parametrize("name", case_names("mse"))
def test_mse_get_tensor_quant_params_bit_exact(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  cfg = cfg_of(m, c)
  p = m.mse.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, arrays[f"{name}/w"])
  check(arrays, name, p, c)


def test_mse_errors(m):
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=4, granularity=q.QuantGranularity.BLOCKWISE_32)
  with pytest.raises(ValueError, match="Blockwise quantization is not supported for MSE"):
    m.mse.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg,
                                  np.ones((4, 32), np.float32))


@pytest.mark.parametrize("name", case_names("hadamard"))
def test_hadamard_get_tensor_quant_params(m, ref_cases, name):
  """Rotation is T2 (sgemm vs butterfly add order): rotated values within 2e-6 of
  max|row|; given identical rotated values everything downstream is exact, so the
  integers may differ by at most 1 on <= 1e-3 of the entries at these tiny sizes."""
  arrays, cases = ref_cases
  c = cases[name]
  w = arrays[f"{name}/w"]
  algo = {} if c["max_hadamard_size"] is None else {"max_hadamard_size": c["max_hadamard_size"]}
  cfg = cfg_of(m, c, **algo)
  rot, h, vec = m.had._rotate_with_diagonal_hadamard(w, w.ndim - 1, c["max_hadamard_size"])
  assert h == c["hadamard_size"] and np.array_equal(vec, arrays[f"{name}/random_binary_vector"])
  ref_rot = arrays[f"{name}/rotated"]
  assert np.max(np.abs(rot - ref_rot)) <= 2e-6 * np.max(np.abs(ref_rot))
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = m.had.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, w)
  assert p.hadamard.hadamard_size == c["hadamard_size"]
  np.testing.assert_allclose(p.scale, arrays[f"{name}/scale"], rtol=1e-6)
  parity_rates.check(f"hadamard+octav reference case {name}", p.quantized_data, arrays[f"{name}/q"], parity_rates.T2)


def test_hadamard_known_answers(m, known_answers):
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=8, symmetric=True,
                                   granularity=q.QuantGranularity.CHANNELWISE)
  for c in known_answers["hadamard_goldens"]["cases"]:
    x = np.tile(np.array(c["input_tile"], dtype=c["input_dtype"]), c["input_reps"])
    exp = np.tile(np.array(c["expected_tile"]), c["expected_reps"])
    if c["reshape"]:
      x, exp = x.reshape(c["reshape"]), exp.reshape(c["reshape"])
    p = m.had.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg, x, None)
    np.testing.assert_array_equal(p.quantized_data, exp)


def test_hadamard_errors(m):
  q = m.qtyping
  cfg = q.TensorQuantizationConfig(num_bits=8, granularity=q.QuantGranularity.CHANNELWISE)
  info = op_info(m, "FULLY_CONNECTED", cfg)
  with pytest.raises(ValueError, match="only supported for weight tensors"):
    m.had.get_tensor_quant_params(info, cfg, None, None)
  with pytest.raises(ValueError, match="static quantization"):
    m.had.get_tensor_quant_params(info, cfg, np.ones((4, 4), np.float32), {})
  with pytest.raises(ValueError, match="rank >= 2"):
    m.had.get_tensor_quant_params(info, cfg, np.ones(4, np.float32), None)


def test_hadamard_rotation_is_orthogonal_at_full_size(m):
  """Size-independent property at BASELINE sizes: H/sqrt(h) is an involution."""
  rng = np.random.default_rng(0)
  for shape, mx in (((64, 4096), None), ((8, 16384), None), ((16, 11008), None)):
    w = rng.standard_normal(shape).astype(np.float32)
    rot, h, _ = m.had._rotate_with_diagonal_hadamard(w, 1, mx)
    assert h == (shape[1] & -shape[1])
    back, _, _ = m.had._rotate_with_diagonal_hadamard(rot, 1, mx)
    assert np.max(np.abs(back - w)) < 2e-5
    ref = (w.reshape(-1, h)[:2] @ O.hadamard_matrix(h)).reshape(-1)
    assert np.max(np.abs(rot.reshape(-1)[: ref.size] - ref)) < 1e-5 * np.max(np.abs(ref))


@pytest.mark.parametrize("alg,bits,gran", [
    ("min_max", 8, "CHANNELWISE"), ("min_max", 4, "BLOCKWISE_32"), ("min_max", 8, "TENSORWISE"),
    ("octav", 4, "CHANNELWISE"), ("octav", 4, "BLOCKWISE_64"), ("mse", 4, "CHANNELWISE"),
    ("hadamard", 8, "CHANNELWISE"), ("oscar", 4, "CHANNELWISE"), ("oscar", 4, "BLOCKWISE_32"),
    ("gptq", 4, "CHANNELWISE"), ("dwr", 4, "CHANNELWISE")])
def test_weight_resident_in_hbm_gives_the_same_result(alg, bits, gran):
  """`tensor_content` handed over as runtime.HbmArray (a float32 weight that already lives on the
  GPU) is read where it is; results equal those for the same weight handed over as ndarray."""
  import torch
  import __graft_entry__ as g
  g.build()
  from mi355q import qtyping as q, runtime as rt
  from mi355q.algorithms.uniform_quantize import (dequantized_weight_recovery, gptq, hadamard_rotation,
                                                  mse, naive_min_max_quantize, octav, oscar)
  rng = np.random.default_rng(2718)
  w = (rng.standard_normal((96, 256)) * 0.05).astype(np.float32)
  qsv = None
  if alg == "dwr":
    w = (np.rint(w / np.float32(0.02)).clip(-7, 7) * np.float32(0.02)).astype(np.float32)
  if alg == "oscar":
    qsv = {"mu2": np.exp(rng.normal(size=256))}
  if alg == "gptq":
    x = rng.standard_normal((512, 256))
    qsv = {"activation_tensor_qsv": {"hessian": 2.0 / 512 * x.T @ x, "num_samples": np.array(512)}}
  mod = dict(min_max=naive_min_max_quantize, octav=octav, mse=mse, hadamard=hadamard_rotation,
             oscar=oscar, gptq=gptq, dwr=dequantized_weight_recovery)[alg]
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(inputs=[0, 1, -1], outputs=[2]), op_name=q.TFLOperationName.FULLY_CONNECTED,
                  subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg, skip_checks=alg == "dwr"))
  host = mod.get_tensor_quant_params(info, cfg, w, qsv)
  resident = rt.HbmArray(torch.from_numpy(w).cuda())
  dev = mod.get_tensor_quant_params(info, cfg, resident, qsv)
  assert resident._host is None, "the resident weight was pulled to the host"  # pylint: disable=protected-access
  for field in ("scale", "zero_point", "quantized_data", "hadamard"):
    a, b = getattr(host, field, None), getattr(dev, field, None)
    if a is None:
      assert b is None
    elif field == "hadamard":
      assert a.hadamard_size == b.hadamard_size
    else:
      assert np.array_equal(np.asarray(a), np.asarray(b)), field


@pytest.mark.parametrize("cols", [8, 96, 128, 256, 1000, 257, 2816, 4096, 5120, 8192, 8200, 11008,
                                  14336, 16384, 16391, 7, 24576])
def test_mse_row_sums_follow_numpy_order_for_every_row_length(m, cols):
  """Channelwise MSE scales (float32 sum of squares per row) are bit-identical to NumPy's for row
  lengths whose pairwise tree is complete (lane-butterfly kernel: 4096 = 128 << 5, 11008 = 8192 +
  (88 << 5), ...), for ragged ones (leaf-table kernel) and for rows of several 8192-chunks."""
  rng = np.random.default_rng(cols)
  w = (rng.standard_normal((37, cols)) * rng.uniform(0.01, 3.0, size=(37, 1))).astype(np.float32)
  cfg = m.qtyping.TensorQuantizationConfig(num_bits=4, symmetric=True,
                                           granularity=m.qtyping.QuantGranularity.CHANNELWISE)
  p = m.mse.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg, w)
  ref = O.mse_quant_params(w, 4, "CHANNELWISE")
  assert np.array_equal(p.scale, ref["scale"])
  assert np.array_equal(np.asarray(p.quantized_data), ref["quantized_data"])


@pytest.mark.parametrize("gran", ["CHANNELWISE", "BLOCKWISE_128"])
def test_octav_row_with_nan_gets_nan_scale_and_zero_integers(m, gran):
  """OCTAV's masked sums skip NaN (comparisons are false), so the clipping constant of a row that
  holds a NaN stays finite while its max|x| is NaN: np.clip(NaN, -c, c) is NaN, the scale is NaN and
  every integer of the row / block is 0 (ref uniform_quantize_tensor.py:529-563; fminf / fmaxf
  would have returned the clip instead)."""
  rng = np.random.default_rng(21)
  w = rng.standard_normal((8, 4096), dtype=np.float32) * np.float32(0.05)
  w[2, 100] = np.nan
  w[5, 4095] = np.nan
  cfg = m.qtyping.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=m.qtyping.QuantGranularity[gran])
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.octav_quant_params(w, 4, gran)
    p = m.octav.get_tensor_quant_params(op_info(m, "FULLY_CONNECTED", cfg), cfg, w)
  assert np.isnan(ref["scale"]).sum() == 2 and np.array_equal(p.scale, ref["scale"], equal_nan=True)
  assert np.array_equal(np.asarray(p.quantized_data), ref["quantized_data"])


@pytest.mark.parametrize("seed", range(6))
def test_octav_masked_sums_with_designed_run_lengths(m, seed):
  """Selected elements come in runs whose lengths are drawn from {1..12, 55..70, 120..140, 300}
  and whose positions fall anywhere relative to the kernel's 64-lane batches and NumPy's
  8192-element chunks: runs that end exactly at a batch end, cross one with a total of 7 / 8 / 9
  elements, span several batches, or straddle a chunk boundary (where NumPy cuts them). Unselected
  elements are tiny, so the masks stay put over the iterations and every iteration exercises the
  same pattern; rows of several lengths, both signs."""
  rng = np.random.default_rng(1000 + seed)
  lengths = [64, 100, 127, 128, 1000, 1024, 1100, 2048, 4096, 5000, 8191, 8192, 8192 + 37, 8192 + 64 + 7, 11008,
             16000, 16383, 16384, 20000]
  pool = np.concatenate([np.arange(1, 13), np.arange(55, 71), np.arange(120, 141), [300]])
  for n in lengths:
    rows = []
    for _ in range(6):
      sign = np.empty(n, np.float32)
      big = np.zeros(n, bool)
      i, on = 0, bool(rng.integers(2))
      while i < n:
        run = int(rng.choice(pool)) if on else int(rng.integers(1, 9))
        big[i:i + run] = on
        sign[i:i + run] = 1.0 if rng.integers(2) else -1.0
        i += run
        on = not on
      mag = np.where(big, rng.uniform(1.0, 2.0, n), rng.uniform(1e-4, 2e-4, n)).astype(np.float32)
      rows.append(mag * sign)
    for cut in (7, 8, 9, 63, 64, 65):                  # runs ending right around the first batch end
      row = np.full(n, 1e-4, np.float32)
      row[max(0, 64 - cut):min(n, 64 + (cut % 5))] = 1.5
      rows.append(row)
    w = np.stack(rows)
    for bits in (4, 8):
      got = m.octav._guess_clipping_with_octav(w, bits, (1,), 10, 3.0)
      ref = O.octav_clip(w, bits, (1,), 10, 3.0)
      assert np.array_equal(got, ref), (n, bits)


@pytest.mark.parametrize("block", [32, 64, 128, 256, 512])
def test_octav_blockwise_groups_kernel_bit_exact(m, block):
  """Blockwise units of 32 .. 512 elements go through octav_groups_kernel (4096 contiguous elements =
  8 .. 128 whole units per workgroup, csrc/reduce_exact.hip) when the tensor is a whole number of
  groups. Designed content: runs of selected elements of every length up to the block size, ending
  at / crossing piece (16) and block boundaries -- where a run must stop --, all-selected blocks
  (runs longer than 128 in blocks of 256 / 512), all-small blocks, single outliers at a block's
  first and last element, NaN / inf, and plain weight-like data; blocks differ, so their guesses
  move independently and reach their fixed points in different iterations."""
  rng = np.random.default_rng(500 + block)
  rows, cols = 24, 2048                                # 49 152 elements = 12 groups
  w = (rng.standard_normal((rows, cols)) * 0.02).astype(np.float32)
  flat = w.reshape(-1, block)
  nb = flat.shape[0]
  for u in range(0, nb, 3):                            # every third block gets a designed pattern
    sel = np.zeros(block, bool)
    i, on = 0, bool(rng.integers(2))
    while i < block:
      run = int(rng.integers(1, min(block, 150) + 1)) if on else int(rng.integers(1, 9))
      sel[i:i + run] = on
      i += run
      on = not on
    sign = np.where(rng.integers(0, 2, block) > 0, 1.0, -1.0)
    flat[u] = np.where(sel, rng.uniform(1.0, 2.0, block), rng.uniform(1e-4, 2e-4, block)).astype(np.float32) * sign
  flat[1] = np.abs(flat[1]) + 1.5                      # every element selected, one sign
  flat[4] = -np.abs(flat[4]) - 0.5
  flat[7] = 1e-5
  flat[10, 0] = 3.0
  flat[13, block - 1] = -3.0
  flat[16, :17 % block] = 2.0                          # a run over the first piece boundary
  flat[19, block - 9:] = 2.5                           # ... up to the block's end (the next block starts small)
  flat[22, 5] = np.inf
  flat[25, 9] = np.nan
  w = flat.reshape(rows, cols)
  data = w.reshape(rows, cols // block, block)
  for bits in (4, 8):
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      ref = O.octav_clip(data, bits, 2, 10, 3.0)
      got = m.octav._guess_clipping_with_octav(data, bits, 2, 10, 3.0)
      ref_all = O.octav_clip(data, bits, 2, 10, 3.0, early_stop=False)
      got_all = m.octav._guess_clipping_with_octav(data, bits, 2, 10, 3.0, early_stop=False)
    assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True), bits
    assert np.array_equal(got_all, ref_all, equal_nan=True), bits
  # a tensor that is not a whole number of groups takes the one-wave-per-unit kernel: same numbers
  odd = w[:5, :cols - block].copy()
  data = odd.reshape(5, (cols - block) // block, block)
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    assert np.array_equal(m.octav._guess_clipping_with_octav(data, 4, 2, 10, 3.0), O.octav_clip(data, 4, 2, 10, 3.0),
                          equal_nan=True)


@pytest.mark.parametrize("shape,qd,block,bits,symmetric", [((64, 10), 0, 32, 8, True), ((4, 96, 6), 1, 32, 4, True),
                                                           ((3, 64, 5, 2), 1, 16, 8, False), ((128, 3), 0, 64, 4, True)])
def test_blockwise_along_an_axis_that_is_not_the_innermost(m, shape, qd, block, bits, symmetric):
  """uniform_quantize / uniform_dequantize with blocks along an inner axis (ref uniform_quantize_tensor.py:164-270 reshapes;
  no op of the reference's tables asks for it -- TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM is 1 for the two 2-D ops -- a
  direct caller may): the blocked axis is moved last on the device and the results moved back; bit-exact against the
  oracle's broadcast form."""
  q_ = m.qtyping
  rng = np.random.default_rng(sum(shape) + block)
  x = (rng.standard_normal(shape) * 0.3).astype(np.float32)
  sshape = list(shape)
  sshape[qd] //= block
  split = list(shape[:qd]) + [shape[qd] // block, block] + list(shape[qd + 1:])
  absmax = np.abs(x.reshape(split)).max(axis=qd + 1)
  qmax = (1 << (bits - 1)) - 1
  scale = (np.maximum(absmax, 1e-9) / qmax).astype(np.float32).reshape(sshape)
  zp = (np.zeros(sshape, np.int32) if symmetric else rng.integers(-3, 4, sshape).astype(np.int32))
  p = q_.UniformQuantParams(scale=scale, zero_point=zp, num_bits=bits, symmetric=symmetric, quantized_dimension=qd, block_size=block)
  got = m.uqt.uniform_quantize(x, p, is_blockwise_quant=True)
  want = O.uniform_quantize(x, scale, zp, bits, symmetric, quantized_dim=qd, block_size=block, is_blockwise_quant=True)
  assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)
  if qd != 0:       # (quantized_dimension 0 means axis 1 to uniform_dequantize: ref :379-387, b/443830202)
    back = m.uqt.uniform_dequantize(got, p)
    ref = O.uniform_dequantize(want, scale, zp, quantized_dim=qd, block_size=block)
    assert back.dtype == ref.dtype and np.array_equal(back, ref)


@pytest.mark.parametrize("shape,axis,bits", [((6, 40, 24), (1,), 4), ((5, 300, 8), 1, 8), ((3, 16, 4, 64), (1, 3), 4),
                                             ((4, 8, 5, 6, 32), (1, 3), 4), ((2, 1, 9, 128, 3), (3,), 2), ((7, 33, 1, 2), (1,), 4)])
def test_octav_kept_axes_separated_by_a_reduced_one(m, shape, axis, bits):
  """ref octav.py:55-61: np.sum takes any axis tuple. A reduced axis BETWEEN kept ones has no kernel of its own -- the
  reduced axes are moved last on the device and each unit is summed as one contiguous run: NumPy's pairwise order for a
  contiguous run, not the strided walk NumPy makes of the original layout, so the clipping constants are held to
  SURVEY 7's tolerance for OCTAV (T2: 1e-6 relative) where every other layout is bit-exact. Shape, dtype, NaN / inf
  placement and the (global) early stop are the reference's."""
  rng = np.random.default_rng(sum(shape) + bits)
  x = (rng.standard_normal(shape) * 0.02).astype(np.float32)
  flat = x.reshape(-1)
  flat[::97] *= 40.0                       # outliers the first guess selects
  flat[5::211] = 0.0
  flat[11] = np.nan
  for early in (True, False):
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      ref = O.octav_clip(x, bits, axis, 10, 3.0, early_stop=early)
      got = m.octav._guess_clipping_with_octav(x, bits, axis, 10, 3.0, early_stop=early)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    rel = np.abs(got[fin].astype(np.float64) - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30)
    parity_rates.note(f"OCTAV clip, kept axes separated by a reduced one {shape} axis={axis} int{bits} early_stop={early}",
                      "max_rel_error", float(rel.max()), 1e-6)


@pytest.mark.parametrize("shape,block,bits", [((4, 64, 6), 32, 4), ((3, 128, 2, 5), 64, 8), ((2, 256, 3), 128, 4)])
def test_blockwise_min_max_along_an_axis_that_is_not_the_innermost(m, shape, block, bits):
  """ref common_quantize.py:1336-1352: blockwise min / max of a FULLY_CONNECTED weight of rank > 2 -- blocks run along
  axis 1 (TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM), which is then not the innermost one. The blocked axis is moved last
  on the device; a minimum does not depend on where its elements sit: bit-exact against the oracle, and so are the
  parameters and the integers that follow from them (uniform_quantize moves the same axis, see above)."""
  q_ = m.qtyping
  from mi355q.algorithms.uniform_quantize import common_quantize
  rng = np.random.default_rng(sum(shape))
  w = (rng.standard_normal(shape) * 0.1).astype(np.float32)
  w[1, 3] = 0.0
  gran = f"BLOCKWISE_{block}"
  cfg = q_.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q_.QuantGranularity[gran])
  got = common_quantize.init_tensor_min_max(w, op_info(m, "FULLY_CONNECTED", cfg))
  ref = O.init_tensor_min_max(w, gran, 1)
  for k in ("min", "max"):
    assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype and np.array_equal(got[k], ref[k])
  zp, scale = O.zp_scale_from_min_max(ref["min"], ref["max"], bits, True, gran, None)
  zp2, scale2 = m.uqt.tensor_zp_scale_from_min_max(got["min"], got["max"], bits, True, q_.QuantGranularity[gran], None)
  assert np.array_equal(scale, scale2) and np.array_equal(zp, zp2)
  p = q_.UniformQuantParams(scale=scale2, zero_point=zp2, num_bits=bits, symmetric=True, quantized_dimension=1, block_size=block)
  want = O.uniform_quantize(w, scale, zp, bits, True, quantized_dim=1, block_size=block, is_blockwise_quant=True)
  assert np.array_equal(m.uqt.uniform_quantize(w, p, is_blockwise_quant=True), want)


@pytest.mark.parametrize("shape,pshape,bits,symmetric", [((4, 5, 6), (4, 1, 6), 8, True), ((2, 4, 5, 6), (1, 4, 1, 6), 4, False),
                                                         ((3, 7, 1, 5, 2), (3, 1, 1, 5, 1), 8, False), ((6, 9, 8), (6, 1, 8), 16, True)])
def test_parameters_that_vary_over_non_adjacent_dimensions(m, shape, pshape, bits, symmetric):
  """ref uniform_quantize_tensor.py:273-409 is NumPy broadcasting and takes scales of any broadcastable shape of the
  tensor's rank; the kernels address one run of adjacent dimensions, so parameters with a broadcast dimension in between
  ([4, 1, 6] over [4, 5, 6]) are repeated over it first. Elementwise arithmetic with the same operands: bit-exact."""
  q_ = m.qtyping
  rng = np.random.default_rng(sum(shape) + bits)
  x = (rng.standard_normal(shape) * 0.5).astype(np.float32)
  scale = (rng.random(pshape) * 0.01 + 0.002).astype(np.float32)
  zp = np.zeros(pshape, np.int32) if symmetric else rng.integers(-5, 6, pshape).astype(np.int32)
  p = q_.UniformQuantParams(scale=scale, zero_point=zp, num_bits=bits, symmetric=symmetric, quantized_dimension=None)
  got = m.uqt.uniform_quantize(x, p)
  want = O.uniform_quantize(x, scale, zp, bits, symmetric)
  assert got.dtype == want.dtype and np.array_equal(got, want)
  back = m.uqt.uniform_dequantize(got, p)
  ref = O.uniform_dequantize(want, scale, zp)
  assert back.dtype == ref.dtype and np.array_equal(back, ref)
