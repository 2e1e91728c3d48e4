"""The opt-in one-read OCTAV kernel (ops.octav_mode("fast"), mi355q_octav_clip_fast_f32) against the reference's
fixtures and the oracle, tolerance class T2: scales within 1e-6 relative, integers +-1 on <= 1e-5 of the
elements. The default (exact) kernels stay bit-exact and are tested in test_gpu_algorithms.py; what is checked here is
that the fast kernel computes the SAME iteration (ref octav.py:30-112: the two masks, zeros in both at guess 0, NaNs in
neither, the float64 `s * N` term, the GLOBAL early stop) with another summation order, and that it is only used where
it was asked for.
"""
import warnings

import numpy as np
import pytest

import parity_rates
from oracle import aeq_oracle as O
from golden_util import case_names
from test_gpu_algorithms import cfg_of, op_info

pytestmark = pytest.mark.gpu

CLIP_RTOL = 1e-6          # on the clipping constants themselves; recorded 2e-7 ... 4e-7 (tools/octav_fast_bench.py)


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import ops, qtyping, runtime
  from mi355q.algorithms.uniform_quantize import hadamard_rotation, octav
  return types.SimpleNamespace(torch=torch, ops=ops, qtyping=qtyping, octav=octav, had=hadamard_rotation, rt=runtime)


def _rel(got, ref):
  got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
  return np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)


@pytest.mark.parametrize("name", case_names("octav"))
def test_reference_fixtures_within_t2(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  cfg = cfg_of(m, c)
  w = arrays[f"{name}/w"]
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    with m.ops.octav_mode("fast"):
      p = m.octav.get_tensor_quant_params(op_info(m, c["op"], cfg), cfg, w)
  ref_scale, ref_q = arrays[f"{name}/scale"], arrays[f"{name}/q"]
  assert p.scale.shape == ref_scale.shape and p.scale.dtype == np.float32
  ok = np.isfinite(ref_scale)
  assert np.array_equal(np.isfinite(p.scale), ok)
  # blockwise scales are rounded to bfloat16 (7 bits of mantissa: a 1e-7 difference in front of the rounding moves a
  # scale by one bf16 step or not at all); per-row scales follow the clipping constant
  if c["block_size"]:
    moved = (np.asarray(p.scale) != ref_scale) & ok
    assert moved.mean() <= 1e-3 and (_rel(p.scale, ref_scale)[ok].max() <= 2.0 ** -7 if moved.any() else True)
  else:
    assert _rel(p.scale, ref_scale)[ok].max() <= CLIP_RTOL
  parity_rates.check(f"OCTAV one-read kernel, reference fixture {name}", p.quantized_data, ref_q,
                     max(parity_rates.T2, 2.0 / ref_q.size) if not c["block_size"] else 2e-3)


@pytest.mark.parametrize("shape,axis,bits", [
    ((512, 4096), (1,), 4), ((64, 11008), (1,), 4), ((16, 16384), (1,), 4), ((7, 65536), (1,), 8), ((300, 40), (1,), 2),
    ((7, 12), (1,), 4), ((256, 32, 128), 2, 4), ((4096, 8, 32), 2, 4), ((33, 5, 256), 2, 8), ((9, 2048), (1,), 8)])
def test_clipping_constants_against_the_oracle(m, shape, axis, bits):
  rng = np.random.default_rng(sum(shape) + bits)
  w = (rng.standard_normal(shape) * 0.02).astype(np.float32)
  flat = w.reshape(-1, shape[-1])
  flat[1] *= 60.0                           # a unit with outliers: the first guess (1) selects some of it
  flat[2] = 0.0                             # all zeros: every iterate is 0 / (s N) = 0, zeros counted twice
  flat[3, :3] = [np.nan, np.inf, -np.inf]   # NaNs are in neither mask; infinities are selected
  flat[4] = np.abs(flat[4]) + 1.5           # everything selected from the first guess on
  flat[5, ::3] = 0.0                        # a third of the unit exactly zero (guess 0: in both masks)
  for early in (True, False):
    ref, ref_iters = O.octav_clip(w, bits, axis, 10, 3.0, early_stop=early, return_iters=True)
    with m.ops.octav_mode("fast"):
      got = m.octav._guess_clipping_with_octav(w, bits, axis, 10, 3.0, early_stop=early)
    exact = m.octav._guess_clipping_with_octav(w, bits, axis, 10, 3.0, early_stop=early)
    assert np.array_equal(exact, ref, equal_nan=True)                     # the default path is untouched
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    rel = _rel(got[fin], ref[fin])
    parity_rates.note(f"OCTAV one-read kernel clip vs oracle {shape} int{bits} early_stop={early}", "max_rel_error",
                      float(rel.max()), CLIP_RTOL)
    # the iteration count is the reference's: the early stop is global and taken at the same iterate
    xd = m.rt.to_device(np.ascontiguousarray(w).reshape(-1))
    with m.ops.octav_mode("fast"):
      _, iters = m.ops.octav_clip(xd, w.size // shape[-1], shape[-1], bits, 10, 3.0, early)
    assert int(iters.item()) == ref_iters


def test_full_size_rows_and_the_public_entry_point(m):
  """4096 x 4096 int4 channelwise and Hadamard + OCTAV through get_tensor_quant_params: T2 on scales and integers."""
  q = m.qtyping
  w = np.random.default_rng(1234).standard_normal((1024, 4096), dtype=np.float32) * np.float32(0.02)
  cfg = q.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=q.QuantGranularity.CHANNELWISE)
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  ref = O.octav_quant_params(w, 4, "CHANNELWISE")
  with m.ops.octav_mode("fast"):
    p = m.octav.get_tensor_quant_params(info, cfg, w)
  assert _rel(p.scale, ref["scale"]).max() <= CLIP_RTOL
  parity_rates.check("OCTAV one-read kernel int4 1024x4096 vs oracle", p.quantized_data, ref["quantized_data"], parity_rates.T2)
  exact = m.octav.get_tensor_quant_params(info, cfg, w)
  assert np.array_equal(exact.scale, ref["scale"]) and np.array_equal(exact.quantized_data, ref["quantized_data"])
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    href = O.hadamard_quant_params(w[:256], 4, "CHANNELWISE")
  with m.ops.octav_mode("fast"):
    hp = m.had.get_tensor_quant_params(info, cfg, w[:256])
  np.testing.assert_allclose(hp.scale, href["scale"], rtol=2e-6)
  parity_rates.check("Hadamard + OCTAV one-read kernel int4 256x4096 vs oracle", hp.quantized_data, href["quantized_data"],
                     parity_rates.T2)


def test_the_fast_kernel_runs_only_where_it_was_asked_for(m, monkeypatch):
  """Default mode, the environment switch, and the shapes the one-read kernel does not take (an odd unit length,
  TENSORWISE, units beyond 65536 elements) -- those stay on the exact kernels in either mode."""
  from mi355q import _ffi
  calls = []
  L = _ffi.lib()

  class Spy:
    def __getattr__(self, name):
      fn = getattr(L, name)
      if name in ("mi355q_octav_clip_fast_f32", "mi355q_octav_clip_f32"):
        return lambda *a: (calls.append(name), fn(*a))[1]
      return fn
  monkeypatch.setattr(_ffi, "lib", lambda: Spy())
  w = np.random.default_rng(2).standard_normal((8, 256)).astype(np.float32)
  ref = O.octav_clip(w, 4, (1,), 10, 3.0)

  def run(x, axis):
    del calls[:]
    return m.octav._guess_clipping_with_octav(x, 4, axis, 10, 3.0), list(calls)
  got, used = run(w, (1,))
  assert used == ["mi355q_octav_clip_f32"] and np.array_equal(got, ref)
  monkeypatch.setenv("MI355Q_OCTAV_FAST", "1")
  got, used = run(w, (1,))
  assert used == ["mi355q_octav_clip_fast_f32"] and _rel(got, ref).max() <= CLIP_RTOL
  with m.ops.octav_mode("exact"):                          # the block wins over the environment
    got, used = run(w, (1,))
    assert used == ["mi355q_octav_clip_f32"] and np.array_equal(got, ref)
  odd = np.random.default_rng(3).standard_normal((8, 250 + 3)).astype(np.float32)
  for x, axis in ((odd, (1,)), (w, None), (np.ones((2, 65540), np.float32), (1,))):
    got, used = run(x, axis)
    assert used == ["mi355q_octav_clip_f32"], (x.shape, axis)
    assert np.array_equal(got, O.octav_clip(x, 4, axis, 10, 3.0))
  with pytest.raises(ValueError, match="octav_mode"):
    with m.ops.octav_mode("quick"):
      pass
