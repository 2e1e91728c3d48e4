"""GPU parity of the C-ABI entry points against the oracle (bit-exact) and the
fixtures recorded from the real reference."""
import warnings

import numpy as np
import pytest

from golden_util import case_names, gen_c2, gen_c3, sha
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
  import torch
  assert torch.cuda.is_available(), "GPU tests need a GPU"
  import __graft_entry__ as g
  g.build()
  from mi355q import ops as _ops
  return _ops


def dev(a):
  import torch
  return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
  return t.cpu().numpy()


def rand(seed, shape, kind="normal"):
  rng = np.random.default_rng(seed)
  w = rng.standard_normal(shape, dtype=np.float32)
  if kind == "outlier":
    w.reshape(-1)[rng.integers(0, w.size, max(1, w.size // 500))] *= 50
  elif kind == "small":
    w *= np.float32(0.02)
  elif kind == "tiny":
    w *= np.float32(1e-7)
  elif kind == "huge":
    w *= np.float32(1e7)
  elif kind == "zeros":
    w[1::3] = 0
  return w


SHAPES_CW = [(1, 4), (3, 8), (7, 36), (5, 33), (16, 256), (9, 260), (4, 1024), (3, 2048),
             (2, 4096), (2, 4100), (2, 8192), (1, 11008), (1, 16384), (1, 16388), (1, 20000),
             (300, 64), (2, 3)]


@pytest.mark.parametrize("bits", [8, 4, 2])
@pytest.mark.parametrize("shape", SHAPES_CW)
def test_requant_channelwise_matches_oracle(ops, shape, bits):
  for kind in ("normal", "outlier", "zeros"):
    w = rand(hash((shape, bits)) % 1000, shape, kind)
    r = ops.requant_sym(dev(w), block=0, bits=bits)
    ref = O.min_max_quant_params(w, bits, True, "CHANNELWISE")
    assert np.array_equal(host(r["scale"]).reshape(-1, 1), ref["scale"])
    assert np.array_equal(host(r["q"]), ref["quantized_data"])


@pytest.mark.parametrize("bits", [8, 4, 2])
@pytest.mark.parametrize("block", [32, 64, 128, 256])
@pytest.mark.parametrize("shape", [(1, 256), (3, 512), (8, 768), (5, 1280), (64, 256)])
def test_requant_blockwise_matches_oracle(ops, shape, block, bits):
  for kind in ("normal", "small", "tiny", "huge", "zeros"):
    w = rand(hash((shape, block, bits)) % 1000, shape, kind)
    want_packed = (w.size * bits) % 8 == 0
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      ref = O.min_max_quant_params(w, bits, True, f"BLOCKWISE_{block}")
    r = ops.requant_sym(dev(w), block=block, bits=bits, want_packed=want_packed,
                        want_scale_f16=True)
    assert np.array_equal(host(r["scale"]), ref["scale"], equal_nan=True), kind
    assert np.array_equal(host(r["q"]), ref["quantized_data"]), kind
    assert np.array_equal(host(r["scale_f16"]), O.blockwise_scale_f16(ref["scale"]), equal_nan=True)
    if want_packed:
      assert np.array_equal(host(r["packed"]),
                            O.pack_data(bits, np.ravel(ref["quantized_data"]).view(np.uint8)))


@pytest.mark.parametrize("bits", [4, 2])
@pytest.mark.parametrize("block", [32, 128])
def test_requant_blockwise_half_integer_quotients(ops, block, bits):
  """The sub-byte blockwise kernel divides once per block and multiplies by the
  reciprocal, falling back to the IEEE division near half-integer quotients. Plant
  quotients k + 0.5 (+- a few ulps) everywhere except at the block maxima."""
  rng = np.random.default_rng(block * bits)
  w = rng.standard_normal((16, 4 * block)).astype(np.float32)
  ref = O.min_max_quant_params(w, bits, True, f"BLOCKWISE_{block}")
  scale = np.repeat(ref["scale"], block, axis=1)
  qmax = 2 ** (bits - 1) - 1
  is_max = np.abs(w) == np.repeat(np.abs(w).reshape(16, 4, block).max(axis=2), block, axis=1)
  k = rng.integers(-qmax - 1, qmax + 1, size=w.shape).astype(np.float32)
  planted = ((k + np.float32(0.5)) * scale).astype(np.float32)
  ulps = rng.integers(-3, 4, size=w.shape).astype(np.int32)
  planted = (planted.view(np.int32) + ulps).view(np.float32)
  keep = is_max | (np.abs(planted) >= np.abs(w).reshape(16, 4, block).max(axis=2).repeat(block, axis=1))
  w2 = np.where(keep, w, planted).astype(np.float32)
  ref2 = O.min_max_quant_params(w2, bits, True, f"BLOCKWISE_{block}")
  assert np.array_equal(ref2["scale"], ref["scale"])  # maxima untouched
  r = ops.requant_sym(dev(w2), block=block, bits=bits, want_packed=True)
  assert np.array_equal(host(r["q"]), ref2["quantized_data"])
  assert np.array_equal(host(r["packed"]), O.pack_data(bits, np.ravel(ref2["quantized_data"]).view(np.uint8)))


def test_requant_odd_block_falls_back_to_generic_kernel(ops):
  # block sizes outside {32,64,128,256} are not AEQ granularities but the ABI takes them
  w = rand(5, (6, 96))
  r = ops.requant_sym(dev(w), block=48, bits=4)
  s = O.blockwise_scale_round(np.max(np.abs(w.reshape(6, 2, 48)), axis=2) / np.float32(7))
  assert np.array_equal(host(r["scale"]), s)
  q = O.uniform_quantize(w, s, np.zeros_like(s, dtype=np.int8), 4, True, quantized_dim=1,
                         block_size=48, is_blockwise_quant=True)
  assert np.array_equal(host(r["q"]), q)


def test_requant_nan_and_inf_rows(ops):
  w = rand(11, (6, 256))
  w[0, 5] = np.nan
  w[1, 7] = np.inf
  w[2, 9] = -np.inf
  w[3, :] = 0
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.min_max_quant_params(w, 8, True, "CHANNELWISE")
    r = ops.requant_sym(dev(w), block=0, bits=8)
    assert np.array_equal(host(r["scale"]).reshape(-1, 1), ref["scale"], equal_nan=True)
    assert np.array_equal(host(r["q"]), ref["quantized_data"])
    ref = O.min_max_quant_params(w, 4, True, "BLOCKWISE_32")
    r = ops.requant_sym(dev(w), block=32, bits=4)
    assert np.array_equal(host(r["scale"]), ref["scale"], equal_nan=True)
    assert np.array_equal(host(r["q"]), ref["quantized_data"])


def test_requant_with_clip_matches_oracle(ops):
  w = rand(21, (16, 512), "outlier")
  for gran, block, bits in (("CHANNELWISE", 0, 4), ("BLOCKWISE_64", 64, 4), ("CHANNELWISE", 0, 8)):
    qdim = O.weight_quantized_dim(gran)
    mm = O.init_tensor_min_max(w, gran, qdim)
    clip = (np.abs(rand(22, mm["min"].shape)) + np.float32(0.5)).astype(np.float32)
    zp, scale = O.zp_scale_from_min_max(mm["min"], mm["max"], bits, True, gran, clip)
    q = O.uniform_quantize(w, scale, zp, bits, True, quantized_dim=qdim, block_size=block,
                           is_blockwise_quant=block > 0)
    r = ops.requant_sym(dev(w), block=block, bits=bits, clip=dev(clip.reshape(-1)))
    assert np.array_equal(host(r["scale"]).reshape(scale.shape), scale)
    assert np.array_equal(host(r["q"]), q)


def test_requant_batched_matches_single(ops):
  ws = [rand(30 + i, (64, 512)) for i in range(5)]
  for block, bits in ((0, 8), (128, 4)):
    b = ops.RequantBatch([dev(w) for w in ws], block, bits, want_q=True,
                         want_packed=True, want_scale_f16=True)
    b.run()
    for i, w in enumerate(ws):
      gran = "CHANNELWISE" if block == 0 else f"BLOCKWISE_{block}"
      ref = O.min_max_quant_params(w, bits, True, gran)
      assert np.array_equal(host(b.q[i]), ref["quantized_data"])
      assert np.array_equal(host(b.scale[i]).reshape(ref["scale"].shape), ref["scale"])
      assert np.array_equal(host(b.packed[i]),
                            O.pack_data(bits, np.ravel(ref["quantized_data"]).view(np.uint8)))


@pytest.mark.parametrize("name", [n for n in case_names("min_max")
                                  if "_asym" not in n and "_tw_" not in n
                                  and "conv" not in n and "3d" not in n])
def test_requant_matches_reference_fixtures(ops, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  w = arrays[f"{name}/w"]
  r = ops.requant_sym(dev(w), block=c["block_size"], bits=c["num_bits"])
  assert np.array_equal(host(r["scale"]).reshape(arrays[f"{name}/scale"].shape),
                        arrays[f"{name}/scale"], equal_nan=True)
  assert np.array_equal(host(r["q"]), arrays[f"{name}/q"])


# ------------------------------------------------------------------ K1 ---
@pytest.mark.parametrize("view", [(1, 1, 5000), (1, 1, 1 << 20), (1, 37, 129), (1, 4096, 64),
                                  (300, 24, 1), (9, 24, 1), (3, 5, 7), (1, 2000, 32)])
def test_minmax_matches_numpy(ops, view):
  outer, ch, inner = view
  w = rand(sum(view), (outer, ch, inner), "outlier")
  mn, mx = ops.minmax(dev(w), outer, ch, inner)
  assert np.array_equal(host(mn), w.min(axis=(0, 2)))
  assert np.array_equal(host(mx), w.max(axis=(0, 2)))


def test_minmax_propagates_nan(ops):
  w = rand(1, (1, 8, 300))
  w[0, 3, 17] = np.nan
  mn, mx = ops.minmax(dev(w), 1, 8, 300)
  assert np.array_equal(host(mn), w.min(axis=(0, 2)), equal_nan=True)
  assert np.array_equal(host(mx), w.max(axis=(0, 2)), equal_nan=True)


# ------------------------------------------------------------------ K3 ---
@pytest.mark.parametrize("name", case_names("min_max"))
def test_quantize_given_params_matches_reference_fixtures(ops, ref_cases, name):
  """T1: with the reference's own scale / zero point the integers are bit-identical."""
  import torch
  arrays, cases = ref_cases
  c = cases[name]
  w, scale, zp = arrays[f"{name}/w"], arrays[f"{name}/scale"], arrays[f"{name}/zero_point"]
  qd, bs = c["quantized_dimension"], c["block_size"]
  if bs:
    outer, ch, inner = 1, scale.size, bs
  elif qd is None:
    outer, ch, inner = 1, 1, w.size
  else:
    outer = int(np.prod(w.shape[:qd]))
    ch = w.shape[qd]
    inner = int(np.prod(w.shape[qd + 1:]))
  narrow = c["symmetric"] and c["num_bits"] >= 8
  q = ops.quantize(dev(w), outer, ch, inner, dev(scale.reshape(-1)),
                   dev(zp.reshape(-1).astype(np.int32)), c["num_bits"], narrow,
                   zp_via_f64=zp.dtype.itemsize >= 4)
  assert q.dtype == torch.int8
  assert np.array_equal(host(q), arrays[f"{name}/q"])


@pytest.mark.parametrize("name", case_names("uniform_quantize"))
def test_quantize_asymmetric_zero_point_paths(ops, ref_cases, name):
  arrays, cases = ref_cases
  x, scale, zp = arrays[f"{name}/x"], arrays[f"{name}/scale"], arrays[f"{name}/zero_point"]
  q = ops.quantize(dev(x), 1, 1, x.size, dev(scale.reshape(-1)), dev(zp.reshape(-1).astype(np.int32)),
                   8, False, zp_via_f64=zp.dtype.itemsize >= 4)
  assert np.array_equal(host(q), arrays[f"{name}/q"])


def test_quantize_float64_scale_known_answers(ops, known_answers):
  # the reference's own vectors use float64 scales (uniform_quantize_tensor_test.py:120-169)
  for c in known_answers["uniform_quantize"]["cases"]:
    x = np.array(c["tensor"], dtype=np.float32)
    q = ops.quantize(dev(x), 1, 1, x.size, dev(np.array(c["scale"], np.float64)),
                     dev(np.array(c["zero_point"], np.int32)), c["num_bits"],
                     c["symmetric"] and c["num_bits"] >= 8)
    assert host(q).tolist() == c["expected"]


def test_quantize_wide_outputs(ops):
  x = (rand(3, (4, 64)) * 1000).astype(np.float32)
  for bits, dt in ((16, np.int16), (32, np.int32)):
    s = np.array([0.01], np.float32)
    zp = np.zeros(1, np.int32)
    ref = O.uniform_quantize(x, s.reshape(1, 1), zp.reshape(1, 1), bits, True)
    q = ops.quantize(dev(x), 1, 1, x.size, dev(s), dev(zp), bits, True, zp_via_f64=True)
    assert host(q).dtype == dt and np.array_equal(host(q), ref)


def test_dequantize_matches_reference_fixture(ops, ref_cases):
  arrays, _ = ref_cases
  q, s, zp = arrays["dq_cw/q"], arrays["dq_cw/scale"], arrays["dq_cw/zero_point"]
  out = ops.dequantize(dev(q), 1, q.shape[0], q.shape[1], dev(s.reshape(-1)),
                       dev(zp.reshape(-1).astype(np.int32)), diff_bits=8)
  assert np.array_equal(host(out), arrays["dq_cw/out"])


def test_dequantize_int8_difference_wraps_like_numpy(ops):
  q = np.array([[-128, 127, 5, -7]], np.int8)
  zp = np.array([[100]], np.int8)
  s = np.array([[0.5]], np.float32)
  ref = O.uniform_dequantize(q, s, zp)
  out = ops.dequantize(dev(q), 1, 1, 4, dev(s.reshape(-1)), dev(zp.reshape(-1).astype(np.int32)), 8)
  assert np.array_equal(host(out), ref)


# ------------------------------------------------------------------ K4 ---
@pytest.mark.parametrize("name", case_names("pack"))
def test_pack_matches_reference_fixtures(ops, ref_cases, name):
  arrays, cases = ref_cases
  out = ops.pack_bits(dev(arrays[f"{name}/data"]), cases[name]["num_bits"])
  assert np.array_equal(host(out), arrays[f"{name}/packed"])


def test_pack_known_answers(ops, known_answers):
  for c in known_answers["pack"]["cases"]:
    out = ops.pack_bits(dev(np.array(c["data"], np.int8)), c["num_bits"])
    assert host(out).tolist() == c["expected"]


@pytest.mark.parametrize("bits,n", [(4, 1 << 20), (2, (1 << 20) + 3), (4, 999999)])
def test_pack_large_roundtrip(ops, bits, n):
  lo, hi = -(2 ** (bits - 1)), 2 ** (bits - 1)
  data = np.random.default_rng(n).integers(lo, hi, size=n).astype(np.int8)
  out = host(ops.pack_bits(dev(data), bits))
  assert np.array_equal(out, O.pack_data(bits, data.view(np.uint8)))
  per = 8 // bits
  unpacked = np.stack([(out >> (bits * k)) & ((1 << bits) - 1) for k in range(per)], 1).reshape(-1)[:n]
  signed = ((unpacked.astype(np.int16) ^ (1 << (bits - 1))) - (1 << (bits - 1))).astype(np.int8)
  assert np.array_equal(signed, data)


# ------------------------------------------------------------------ K7 ---
def test_act_minmax_matches_reference_fixtures(ops, ref_cases):
  arrays, cases = ref_cases
  names = [n for n in case_names("activation_min_max") if arrays[f"{n}/x"].dtype == np.float32]
  out = host(ops.act_minmax([dev(arrays[f"{n}/x"].reshape(-1)) for n in names]))
  for i, n in enumerate(names):
    assert out[i, 0] == arrays[f"{n}/min"].item(), n
    assert out[i, 1] == arrays[f"{n}/max"].item(), n


def test_act_minmax_large_with_sentinels(ops):
  rng = np.random.default_rng(9)
  xs = []
  for i in range(6):
    x = rng.standard_normal((1, 64, 1000 + i), dtype=np.float32) * (1 + i)
    if i % 2:
      x.reshape(-1)[rng.integers(0, x.size, 5)] = [np.inf, -np.inf, 3.39e38, -3.39e38, np.inf]
    xs.append(x)
  out = host(ops.act_minmax([dev(x.reshape(-1)) for x in xs]))
  for i, x in enumerate(xs):
    ref = O.activation_min_max(x, -3e38, 3e38)
    assert out[i, 0] == ref["min"].item() and out[i, 1] == ref["max"].item()
  out = host(ops.act_minmax([dev(x.reshape(-1)) for x in xs], None, None))
  for i, x in enumerate(xs):
    assert out[i, 0] == x.min() and out[i, 1] == x.max()


# --------------------------------------------- BASELINE sizes (digests) ---
def test_c2_full_size_digest(ops, ref_digests):
  d = ref_digests["c2"]
  w = gen_c2()
  r = ops.requant_sym(dev(w), block=0, bits=8)
  assert sha(host(r["q"])) == d["q"]
  assert sha(host(r["scale"])) == d["scale"]
  v = ref_digests["c2_variant"]
  w[7, :] = 0
  w[9, 5] = 1e4
  r = ops.requant_sym(dev(w), block=0, bits=8)
  assert sha(host(r["q"])) == v["q"] and sha(host(r["scale"])) == v["scale"]
  d4 = ref_digests["c2_int4"]
  r = ops.requant_sym(dev(gen_c2()), block=0, bits=4, want_packed=True)
  assert sha(host(r["q"])) == d4["q"] and sha(host(r["packed"])) == d4["packed"]
  assert sha(host(r["scale"])) == d4["scale"]


@pytest.mark.parametrize("layer", [0, 1])
def test_c3_full_size_digest(ops, ref_digests, layer):
  d = ref_digests[f"c3_layer{layer}"]
  w = gen_c3(layer)
  assert sha(w) == d["w"]
  r = ops.requant_sym(dev(w), block=128, bits=4, want_packed=True, want_scale_f16=True)
  assert sha(host(r["packed"])) == d["packed"]
  assert sha(host(r["q"])) == d["q"]
  assert sha(host(r["scale"])) == d["scale"]
  assert sha(host(r["scale_f16"])) == d["scale_f16"]


@pytest.mark.parametrize("h", [2, 16, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_hadamard_rotate_all_sizes_partial_tiles_and_in_place(h):
  """Both FWHT kernels (radix-2 in LDS below 256 / above 4096, radix-16 tile kernel between)
  against the dense H / sqrt(h) product in float64: T2 (only the float32 add order differs),
  including a vector count that leaves the last 4096-element tile partly empty, and out == x."""
  import torch
  from mi355q import ops
  import oracle.aeq_oracle as O
  rng = np.random.default_rng(h)
  n_vec = max(3, (3 * 4096) // h + 5)               # not a whole number of tiles
  x = rng.standard_normal((n_vec, h)).astype(np.float32)
  hm = O.hadamard_matrix(h).astype(np.float64)
  want = x.astype(np.float64) @ hm
  xd = torch.from_numpy(x).cuda()
  got = ops.hadamard_rotate(xd, h).cpu().numpy()
  assert got.shape == x.shape and got.dtype == np.float32
  tol = 2e-6 * np.abs(x).max(axis=1, keepdims=True) * np.sqrt(h)
  assert np.all(np.abs(got - want) <= tol)
  # an involution up to rounding: H / sqrt(h) is symmetric and orthogonal
  back = ops.hadamard_rotate(torch.from_numpy(got).cuda(), h).cpu().numpy()
  assert np.all(np.abs(back - x) <= 2 * tol + 1e-6)
  # in place through the C ABI (out == x is allowed by include/mi355q.h)
  from mi355q import _ffi, runtime as rt
  inplace = torch.from_numpy(x).cuda()
  _ffi.check(_ffi.lib().mi355q_hadamard_rotate_f32(rt.ptr(inplace), n_vec, h, rt.ptr(inplace), rt.stream_ptr()))
  assert np.array_equal(inplace.cpu().numpy(), got)


def test_cast_f16_matches_numpy_astype():
  """float_casting's cast: round to nearest even, overflow -> inf, subnormal halves, NaN."""
  import torch
  from mi355q import ops, runtime as rt
  rng = np.random.default_rng(77)
  x = np.concatenate([
      rng.standard_normal(100003).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 100003).astype(np.float32),
      np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 70000.0, -1e9, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5, 6.09e-5,
                np.inf, -np.inf, np.nan, 1.0009765625, 1.00048828125, 1.00146484375], np.float32)])
  for arr in (x, x[1:], x[:7], x[3:4]):           # aligned, misaligned, tiny
    got = rt.to_numpy(ops.cast_f16(rt.to_device(arr)))
    want = arr.astype(np.float16)
    assert got.dtype == np.float16 and np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_device_alloc_clock_probe_and_the_held_inverse_workspace(ops):
  """The round-4 plumbing entry points: mi355q_device_alloc / _free hand out device memory the kernels can use (the
  workspace ops.HinvWorkspace allocates on a helper thread: an inverse computed in it equals the one computed in the
  framework's allocation, bit for bit, also when the batched call borrows it); mi355q_clock_probe counts a plausible
  shader clock."""
  import ctypes
  import torch
  from mi355q import _ffi
  O_ = ops
  L = _ffi.lib()
  p = ctypes.c_void_p()
  _ffi.check(L.mi355q_device_alloc(1 << 20, ctypes.byref(p)))
  assert p.value
  _ffi.check(L.mi355q_device_free(p))
  assert L.mi355q_device_alloc(1 << 20, None) == -1 and L.mi355q_device_free(None) == 0
  out = torch.zeros(2, dtype=torch.int64, device="cuda")
  _ffi.check(L.mi355q_clock_probe(0.01, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
  clocks, ticks = out.cpu().tolist()
  assert 0.9e6 <= ticks <= 1.5e6 and 300.0 <= 100.0 * clocks / ticks <= 3500.0, (clocks, ticks)
  assert L.mi355q_clock_probe(2.0, ctypes.c_void_p(out.data_ptr()), None) == -1
  d = 4096
  x = torch.randn((6000, d), device="cuda")
  prod = O_.gptq_xtx_accum(x, None)
  plain, info = O_.gptq_hinv_from_product(prod, 2.0 / 6000)
  small = [torch.randn((900, 320), device="cuda", dtype=torch.float64) for _ in range(3)]
  hs = [s.T @ s / 900 for s in small]
  plain_small = O_.gptq_hinv_batched(hs)
  ws = O_.HinvWorkspace(d)
  with ws:
    assert O_.HinvWorkspace.current is ws and ws.pointer(ws.nbytes) is not None and ws.pointer(ws.nbytes + 1) is None
    held, info2 = O_.gptq_hinv_from_product(prod, 2.0 / 6000)
    held_small = O_.gptq_hinv_batched(hs)
    torch.cuda.synchronize()
  assert O_.HinvWorkspace.current is None
  assert int(info.item()) == int(info2.item()) == 0 and torch.equal(plain, held)
  for (a, _), (b, _) in zip(plain_small, held_small):
    assert torch.equal(a, b)
