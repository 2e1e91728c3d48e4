"""Pins oracle/aeq_oracle.py against (i) the reference's own known-answer test
vectors and (ii) outputs of the real reference recorded by
tests/golden/gen/make_golden.py. CPU only."""
import json
import os
import warnings

import numpy as np
import pytest

from golden_util import case_names, gen_c2, gen_c3, num, sha
from oracle import aeq_oracle as O


def _eq(a, b):
  a, b = np.asarray(a), np.asarray(b)
  assert a.shape == b.shape, (a.shape, b.shape)
  assert a.dtype == b.dtype, (a.dtype, b.dtype)
  assert np.array_equal(a, b, equal_nan=True)


# ---------------------------------------------------------------- known answers
def test_quantized_range(known_answers):
  for bits, signed, lo, hi in known_answers["quantized_range"]["cases"]:
    assert O.qrange(bits, signed) == (lo, hi)


def test_uniform_quantize_known_answers(known_answers):
  for c in known_answers["uniform_quantize"]["cases"]:
    q = O.uniform_quantize(np.array(c["tensor"]), np.array(c["scale"]),
                           np.array(c["zero_point"]), c["num_bits"], c["symmetric"],
                           quantized_dim=0)
    assert q.dtype == np.int8
    assert q.tolist() == c["expected"]


def test_uniform_quantize_errors():
  x = np.array([-3.0, 1.3, 2.4, 16.0])
  with pytest.raises(ValueError, match=r"Ranks of scales \(3\) and zps \(2\)"):
    O.uniform_quantize(x, np.array([[[1.2666667]]]), np.array([[-6]]), 4, True, 0)
  with pytest.raises(ValueError, match="zero_points need to be"):
    O.uniform_quantize(x, np.array([1.0]), np.array([0.5]), 8, True, 0)
  with pytest.raises(ValueError, match="single element for scalar tensor"):
    O.uniform_quantize(np.array(6.66), np.array([1.0, 2.0]), np.array([0, 0]), 8, True, 0)
  with pytest.raises(ValueError, match="is not divisible by block size"):
    O.init_tensor_min_max(np.ones((4, 33), np.float32), "BLOCKWISE_32", 1)


def test_uniform_dequantize_known_answers(known_answers):
  ka = known_answers["uniform_dequantize"]
  for c in ka["cases"]:
    out = O.uniform_dequantize(np.array(c["quantized"]), np.array(c["scale"]),
                               np.array(c["zero_point"]), quantized_dim=0)
    np.testing.assert_almost_equal(out, c["expected"], decimal=ka["places"])
  kb = known_answers["uniform_dequantize_blockwise"]
  out = O.uniform_dequantize(np.array(kb["quantized"]), np.array(kb["scale"]),
                             np.array(kb["zero_point"]),
                             quantized_dim=kb["quantized_dimension"],
                             block_size=kb["block_size"])
  np.testing.assert_almost_equal(out, kb["expected"], decimal=kb["places"])


def test_zp_scale_known_answers(known_answers):
  k = known_answers["zp_scale_with_clipping"]
  for c in k["cases"]:
    zp, scale = O.zp_scale_from_min_max(np.array(k["min"]), np.array(k["max"]),
                                        c["num_bits"], c["symmetric"], "TENSORWISE",
                                        np.array(k["clipping"]))
    assert zp.shape == scale.shape == (1, 1)
    if c["symmetric"]:
      assert zp[0] == 0
    assert scale[0] == np.array(k["clipping"]) / c["quantized_bound"]
  k = known_answers["zp_scale_basic"]
  data = np.array(k["data"])
  mn, mx = np.min(data, keepdims=True), np.max(data, keepdims=True)
  for c in k["cases"]:
    bits, sym = c["num_bits"], c["symmetric"]
    zp, scale = O.zp_scale_from_min_max(mn, mx, bits, sym, "TENSORWISE")
    assert zp.shape == scale.shape
    max_q = 2**bits / 2 - 1
    assert abs(scale[0] * (max_q - zp[0]) - mx) < 1e-3
    min_q = -(2**bits) / 2 + (1 if sym else 0)
    cmin = scale[0] * (min_q - zp[0])
    if sym:
      assert abs(cmin + mx) < 1e-3
    else:
      assert cmin == 0


def test_bias_known_answers(known_answers):
  k = known_answers["bias"]
  bias = np.array(k["bias"])
  for c in k["cases"]:
    ch = c["channels"]
    w_scale = np.array([k["weight_scale"]] * ch, dtype=np.float32)
    q, scale, zp, bits, qdim = O.quantize_bias(bias, np.array(k["input_scale"]),
                                               w_scale, c["activation_num_bits"])
    assert bits == (32 if c["activation_num_bits"] == 8 else 64)
    assert scale.ndim == 1 and zp.ndim == 1 and len(scale) == ch
    assert scale[0] == np.array(k["input_scale"])[0] * w_scale[0]
    assert zp[0] == 0
    assert qdim == (0 if ch == 2 else None)
    deq = O.uniform_dequantize(q, scale, zp, quantized_dim=qdim)
    np.testing.assert_almost_equal(deq.flatten(), bias, decimal=5)
    if bits == 64:
      assert q.dtype == np.int64
      assert q.tolist() == q.astype(np.int32).tolist()


def test_blockwise32_scales_property():
  """ref naive_min_max_quantize_test.py:162-205, with an independent bf16 (torch)."""
  import torch
  x = np.random.default_rng(3).uniform(-10, 10, size=(4, 32)).astype(np.float32)
  r = O.min_max_quant_params(x, 4, True, "BLOCKWISE_32")
  assert r["zero_point"].shape == (4, 1) and not r["zero_point"].any()
  exp = np.max(np.abs(x), axis=1, keepdims=True) / np.float32(7.0)
  exp = torch.from_numpy(exp).to(torch.bfloat16).to(torch.float16).to(torch.float32).numpy()
  assert r["scale"].shape == (4, 1)
  assert np.array_equal(r["scale"], exp)
  assert r["block_size"] == 32 and r["quantized_dimension"] == 1
  assert r["quantized_data"].shape == x.shape


def test_blockwise_reshape(known_answers):
  k = known_answers["blockwise_reshape"]
  assert O.split_blocks(k["shape"], k["quantized_dim"], k["block"]) == k["expected_shape"]


def test_activation_min_max_known_answers(known_answers):
  k = known_answers["activation_min_max"]
  for c in k["cases"]:
    x = np.array([num(v) for v in c["x"]], dtype=np.float32)
    q = O.activation_min_max(x, c["lo"], c["hi"])
    assert q["min"].item() == c["min"] and q["max"].item() == c["max"]
  q = O.activation_min_max(np.array(k["int_case"]["x"], np.int32))
  assert (q["min"].item(), q["max"].item()) == (k["int_case"]["min"], k["int_case"]["max"])


def test_hadamard_known_answers(known_answers):
  for c in known_answers["hadamard_goldens"]["cases"]:
    x = np.tile(np.array(c["input_tile"], dtype=c["input_dtype"]), c["input_reps"])
    exp = np.tile(np.array(c["expected_tile"]), c["expected_reps"])
    if c["reshape"]:
      x, exp = x.reshape(c["reshape"]), exp.reshape(c["reshape"])
    r = O.hadamard_quant_params(x, 8, "CHANNELWISE")
    np.testing.assert_array_equal(r["quantized_data"], exp)
    assert r["hadamard_size"] == 2


def test_gptq_known_answers(known_answers):
  k = known_answers["gptq_hessian"]
  val = k["val"]
  for key in ("input", "output"):
    x = np.array([[[num(v, val) for v in row] for row in m] for m in k[key]])
    q = O.activation_qsv(x)
    assert q["min"].item() == k[f"{key}_min"] and q["max"].item() == k[f"{key}_max"]
    assert q["num_samples"] == k["num_samples"]
    h = O.gptq_hessian(x)
    x2 = x.reshape(-1, 3)
    np.testing.assert_allclose(h, 2.0 * x2.T @ x2)
  k = known_answers["gptq_goldens"]
  w = np.array(k["weights"], dtype=np.float32)
  for c in k["cases"]:
    qsv = {"activation_tensor_qsv": {"hessian": np.array(k["hessian"], np.float32),
                                     "num_samples": 1}}
    if c["qsv_min"] is not None:
      qsv["min"], qsv["max"] = np.array(c["qsv_min"]), np.array(c["qsv_max"])
    r = O.gptq_quant_params(w, 8, True, "TENSORWISE", qsv)
    assert r["quantized_data"].dtype == np.int8
    np.testing.assert_allclose(r["scale"], np.array([[c["expected_scale"]]]), rtol=1e-6)
    np.testing.assert_array_equal(r["quantized_data"], np.array(c["expected"]))
  # params only when there is no Hessian / no content
  r = O.gptq_quant_params(None, 8, True, "TENSORWISE",
                          {"min": np.array([[-1.1]]), "max": np.array([[2.2]])})
  assert r["quantized_data"] is None
  np.testing.assert_allclose(r["scale"], np.array([[2.2 / 127]]))


def test_gptq_blockwise_known_answer(known_answers):
  k = known_answers["gptq_blockwise"]
  w = np.array(k["weights_times_127"], np.float32) / 127
  a = np.array(k["qsv_abs"])
  zp, scale = O.zp_scale_from_min_max(-a, a, 8, True, "BLOCKWISE_32")
  exp_scale = O.blockwise_scale_round((a / 127).astype(np.float32))
  np.testing.assert_allclose(scale, exp_scale)
  q = O.gptq_apply(w, scale, zp, 8, True, np.eye(4, dtype=np.float32), "BLOCKWISE_32",
                   block_size=k["block"])
  assert (q == k["expected_all"]).all() and q.dtype == np.int8


def test_pack_known_answers(known_answers):
  for c in known_answers["pack"]["cases"]:
    out = O.pack_data(c["num_bits"], np.array(c["data"], dtype=np.int8).view(np.uint8))
    assert out.dtype == np.uint8 and out.tolist() == c["expected"]


def test_qsv_known_answers(known_answers):
  k = known_answers["qsv_moving_average"]
  for c in k["cases"]:
    r = O.moving_average_update(k["old"], k["new"], c["smoothing_factor"])
    assert r["min"] == pytest.approx(c["min"]) and r["max"] == pytest.approx(c["max"])
  a = k["array_case"]
  r = O.moving_average_update({"min": np.array(a["old_min"]), "max": np.array(a["old_max"])},
                              {"min": np.array(a["new_min"]), "max": np.array(a["new_max"])},
                              a["smoothing_factor"])
  np.testing.assert_array_almost_equal(r["min"], a["min"])
  np.testing.assert_array_almost_equal(r["max"], a["max"])
  assert O.moving_average_update(None, k["new"]) == k["new"]
  for c in known_answers["qsv_min_max_update"]["cases"]:
    r = O.min_max_update({"min": np.array(c["old_min"]), "max": np.array(c["old_max"])},
                         {"min": np.array(c["new_min"]), "max": np.array(c["new_max"])})
    np.testing.assert_array_equal(r["min"], np.array(c["min"]))
    np.testing.assert_array_equal(r["max"], np.array(c["max"]))
  k = known_answers["qsv_gptq_merge"]
  for c in k["cases"]:
    old = dict(k["old"], hessian=np.array(k["old_hessian"]), num_samples=c["old_ns"])
    new = dict(k["new"], hessian=np.array(k["new_hessian"]), num_samples=c["new_ns"])
    r = O.gptq_and_moving_average_update(old, new)
    np.testing.assert_array_almost_equal(r["hessian"], np.array(c["hessian"]))
    assert r["num_samples"] == c["ns"]
    assert r["min"] == pytest.approx(0.95 * 1.0) and r["max"] == pytest.approx(0.95 * 10 + 0.05 * 12)


# ------------------------------------------- recorded outputs of the reference
def _check_params(arrays, name, r):
  _eq(r["scale"], arrays[f"{name}/scale"])
  _eq(r["zero_point"], arrays[f"{name}/zero_point"])
  if f"{name}/q" in arrays:
    _eq(r["quantized_data"], arrays[f"{name}/q"])


@pytest.mark.parametrize("name", case_names("min_max"))
def test_min_max_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    r = O.min_max_quant_params(arrays[f"{name}/w"], c["num_bits"], c["symmetric"],
                               c["granularity"], c["op"])
  assert r["quantized_dimension"] == c["quantized_dimension"]
  assert r["block_size"] == c["block_size"]
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("min_max_qsv"))
def test_min_max_qsv_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  r = O.min_max_quant_params(None, c["num_bits"], c["symmetric"], c["granularity"],
                             c["op"], qsv={"min": arrays[f"{name}/min"],
                                           "max": arrays[f"{name}/max"]})
  assert r["quantized_data"] is None
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("octav"))
def test_octav_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    r = O.octav_quant_params(arrays[f"{name}/w"], c["num_bits"], c["granularity"], c["op"])
  _eq(np.asarray(r["clip"]).reshape(-1), arrays[f"{name}/clip"].reshape(-1))
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("mse"))
def test_mse_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  r = O.mse_quant_params(arrays[f"{name}/w"], c["num_bits"], c["granularity"], c["op"])
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("hadamard"))
def test_hadamard_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    r = O.hadamard_quant_params(arrays[f"{name}/w"], c["num_bits"], c["granularity"],
                                c["op"], c["max_hadamard_size"])
  assert r["hadamard_size"] == c["hadamard_size"]
  _eq(r["random_binary_vector"], arrays[f"{name}/random_binary_vector"])
  _eq(r["rotated"], arrays[f"{name}/rotated"])   # same BLAS, same box -> bit equal
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("gptq"))
def test_gptq_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  h = O.gptq_hessian(arrays[f"{name}/x"])
  _eq(h, arrays[f"{name}/hessian"])
  _eq(O.gptq_hessian_inverse(h), arrays[f"{name}/hinv"])
  qsv = {"activation_tensor_qsv": {"hessian": h, "num_samples": 1}}
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    r = O.gptq_quant_params(arrays[f"{name}/w"], c["num_bits"], c["symmetric"],
                            c["granularity"], qsv, c["op"])
  _check_params(arrays, name, r)


@pytest.mark.parametrize("name", case_names("activation_min_max"))
def test_activation_min_max_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  q = O.activation_min_max(arrays[f"{name}/x"], c["lo"], c["hi"])
  _eq(q["min"], arrays[f"{name}/min"])
  _eq(q["max"], arrays[f"{name}/max"])


def test_qsv_replay_matches_reference(ref_cases):
  arrays, _ = ref_cases
  mins, maxs = arrays["qsv_replay/mins"], arrays["qsv_replay/maxs"]
  q = O.replay_moving_average(list(mins), list(maxs), 0.95)
  _eq(q["min"], arrays["qsv_replay/ema_min"])
  _eq(q["max"], arrays["qsv_replay/ema_max"])
  m = None
  for a, b in zip(mins, maxs):
    m = O.min_max_update(m, {"min": a, "max": b})
  _eq(m["min"], arrays["qsv_replay/mm_min"])
  _eq(m["max"], arrays["qsv_replay/mm_max"])


def test_qsv_hessian_merge_matches_reference(ref_cases):
  arrays, cases = ref_cases
  ns = cases["qsv_hessian"]["num_samples"]
  q = None
  for k, n in enumerate(ns):
    q = O.gptq_and_moving_average_update(
        q, {"min": np.float32(-1), "max": np.float32(1),
            "hessian": arrays[f"qsv_hessian/h{k}"], "num_samples": n})
  _eq(q["hessian"], arrays["qsv_hessian/merged"])
  assert q["num_samples"] == cases["qsv_hessian"]["total"]


@pytest.mark.parametrize("name", case_names("pack"))
def test_pack_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  out = O.pack_data(cases[name]["num_bits"], arrays[f"{name}/data"].view(np.uint8))
  _eq(out, arrays[f"{name}/packed"])


@pytest.mark.parametrize("name", case_names("uniform_quantize"))
def test_uniform_quantize_matches_reference(ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  q = O.uniform_quantize(arrays[f"{name}/x"], arrays[f"{name}/scale"],
                         arrays[f"{name}/zero_point"], c["num_bits"], c["symmetric"],
                         quantized_dim=c["quantized_dimension"])
  _eq(q, arrays[f"{name}/q"])


def test_dequantize_and_bias_match_reference(ref_cases):
  arrays, cases = ref_cases
  out = O.uniform_dequantize(arrays["dq_cw/q"], arrays["dq_cw/scale"],
                             arrays["dq_cw/zero_point"], quantized_dim=0)
  _eq(out, arrays["dq_cw/out"])
  for name in ("bias_i32", "bias_i64"):
    c = cases[name]
    q, scale, _, bits, qdim = O.quantize_bias(arrays[f"{name}/bias"],
                                              arrays[f"{name}/in_scale"],
                                              arrays[f"{name}/w_scale"], c["in_num_bits"])
    _eq(q, arrays[f"{name}/q"])
    _eq(scale, arrays[f"{name}/scale"])
    assert bits == c["num_bits"] and qdim == c["quantized_dimension"]


# -------------------------------------------------- BASELINE sizes (digests)
def test_c2_digests(ref_digests):
  d = ref_digests["c2"]
  w = gen_c2()
  assert sha(w) == d["w"]
  r = O.min_max_quant_params(w, 8, True, "CHANNELWISE")
  assert sha(r["quantized_data"]) == d["q"]
  assert sha(r["scale"]) == d["scale"]
  assert sha(r["zero_point"]) == d["zero_point"]
  assert r["quantized_data"][0, :8].tolist() == d["q_head"]
  v = ref_digests["c2_variant"]
  w[7, :] = 0
  w[9, 5] = 1e4
  r = O.min_max_quant_params(w, 8, True, "CHANNELWISE")
  assert sha(r["quantized_data"]) == v["q"] and sha(r["scale"]) == v["scale"]
  assert float(r["scale"][7, 0]) == v["scale_7"] and float(r["scale"][9, 0]) == v["scale_9"]
  assert not r["quantized_data"][7].any()


def test_c2_int4_digests(ref_digests):
  d = ref_digests["c2_int4"]
  r = O.min_max_quant_params(gen_c2(), 4, True, "CHANNELWISE")
  assert sha(r["quantized_data"]) == d["q"] and sha(r["scale"]) == d["scale"]
  assert sha(O.pack_data(4, np.ravel(r["quantized_data"]).view(np.uint8))) == d["packed"]


def test_c3_layer0_digests(ref_digests):
  d = ref_digests["c3_layer0"]
  w = gen_c3(0)
  assert sha(w) == d["w"]
  r = O.min_max_quant_params(w, 4, True, "BLOCKWISE_128")
  assert sha(r["quantized_data"]) == d["q"]
  assert sha(r["scale"]) == d["scale"]
  assert sha(O.blockwise_scale_f16(r["scale"])) == d["scale_f16"]
  packed = O.pack_data(4, np.ravel(r["quantized_data"]).view(np.uint8))
  assert sha(packed) == d["packed"] and packed[:4].tolist() == d["packed_head"]


def test_octav_anchor_digest(ref_digests):
  d = ref_digests["octav_anchor"]
  w = np.random.default_rng(d["seed"]).standard_normal(tuple(d["shape"]), dtype=np.float32)
  r = O.octav_quant_params(w, 4, "CHANNELWISE")
  assert sha(r["quantized_data"]) == d["q"] and sha(r["scale"]) == d["scale"]


# ------------------------------------------------------------------ OSCAR (f4) ---
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_oscar_cases.json")) as _f:
  _OSCAR = {c["name"]: c for c in json.load(_f)["cases"]}


@pytest.fixture(scope="module")
def oscar_arrays():
  return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_oscar_cases.npz"))


@pytest.mark.parametrize("name", sorted(n for n in _OSCAR if n != "oscar_calib"))
def test_oscar_matches_reference(oscar_arrays, name):
  """Channel scales, clip bounds, scale (FP64, or bf16-rounded float32 for blockwise), int
  weights and the float32 multiplier of the real reference's OSCAR, stage by stage."""
  z, c = oscar_arrays, _OSCAR[name]
  w = z[f"{name}/w"]
  mu2 = z[f"{name}/mu2"] if c["has_mu2"] else None
  s64 = np.ones(w.shape[1])
  if mu2 is not None:
    block = O.block_size_of(c["granularity"]) if O.is_blockwise(c["granularity"]) else 0
    s, gain = O.oscar_channel_scales(np.asarray(w, np.float64), np.asarray(mu2, np.float64), block)
    assert (s is not None) == c["scaled"]
    if s is not None:
      _eq(s, z[f"{name}/s"])
      s64 = s
  _eq(O.oscar_clip_bounds(np.asarray(w, np.float64) * s64, None if mu2 is None else mu2 / (s64 * s64),
                          c["num_bits"], c["granularity"]), z[f"{name}/bounds"])
  r = O.oscar_quant_params(w, mu2, c["num_bits"], c["granularity"])
  _eq(r["scale"], z[f"{name}/scale"])
  assert str(r["scale"].dtype) == c["scale_dtype"]
  _eq(r["quantized_data"], z[f"{name}/q"])
  _eq(r["multiplier"], z[f"{name}/multiplier"])
  assert np.array_equal(r["zero_point"], z[f"{name}/zero_point"])
  assert r["quantized_dimension"] == c["quantized_dimension"] and r["block_size"] == c["block_size"]


def test_oscar_calibration_statistic_and_merge_match_reference(oscar_arrays):
  z, c = oscar_arrays, _OSCAR["oscar_calib"]
  q = None
  for i in range(c["steps"]):
    x = z[f"oscar_calib/x{i}"]
    new = O.activation_min_max(x, -3e38, 3e38)
    new["num_samples"] = np.array(x.shape[0])
    new["mu2"] = O.oscar_mu2(x)
    _eq(new["mu2"], z[f"oscar_calib/mu2_{i}"])
    q = O.oscar_and_moving_average_update(q, new)
    _eq(np.asarray(q["mu2"]), z[f"oscar_calib/merged_mu2_{i}"])
    _eq(q["min"], z[f"oscar_calib/merged_min_{i}"])
  assert int(q["num_samples"]) == c["num_samples"]


# ------------------------------------------------------------ dequantized weight recovery ---
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_dwr_cases.json")) as _f:
  _DWR = {c["name"]: c for c in json.load(_f)["cases"]}


@pytest.mark.parametrize("name", sorted(_DWR))
def test_dequantized_weight_recovery_matches_reference(name):
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_dwr_cases.npz"))
  c = _DWR[name]
  w = z[f"{name}/w"]
  if "error" in c:
    with pytest.raises(RuntimeError, match="Failed to recover the original quantized values"):
      O.dwr_quant_params(w, c["num_bits"], c["granularity"], c["op"])
    return
  r = O.dwr_quant_params(w, c["num_bits"], c["granularity"], c["op"])
  _eq(r["scale"], z[f"{name}/scale"])
  assert str(r["scale"].dtype) == c["scale_dtype"]
  _eq(r["quantized_data"], z[f"{name}/q"])
  assert np.array_equal(r["zero_point"], z[f"{name}/zero_point"])
  assert r["quantized_dimension"] == c["quantized_dimension"] and r["block_size"] == c["block_size"]
