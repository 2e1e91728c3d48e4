"""OCTAV on blockwise units of 32 / 64 / 128 / 256 elements: the lane-per-unit kernel (csrc/reduce_exact.hip,
octav_unit_lanes_kernel) against the oracle's NumPy iteration (ref octav.py:30-112) and against the kernel it replaced
(MI355Q_OCTAV_UNIT_LANES=0: octav_groups_kernel; the rows kernel for units of 256), bit for bit: clipping constants AND iteration counts.

The inputs aim at what distinguishes the two walks of the new kernel: runs of selected elements of exactly 7 / 8 / 9 / 15 /
16 / 17 / whole-unit length (NumPy's left-to-right loop below 8 elements, eight strided accumulators + tail from 8 on),
runs that touch a unit's first or last element, signed zeros, NaN / inf, units that select nothing at the first guess
(ordinary weights) and units that do (sigma = 1), unit counts that leave lanes of the last wave without a unit."""
import os

import numpy as np
import pytest

from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as ge
  ge.build()
  from mi355q import ops, runtime as rt
  return dict(torch=torch, ops=ops, rt=rt)


def _clip(g, w, unit_len, bits, early_stop=True, lanes=True):
  torch, ops = g["torch"], g["ops"]
  old = os.environ.get("MI355Q_OCTAV_UNIT_LANES")
  os.environ["MI355Q_OCTAV_UNIT_LANES"] = "1" if lanes else "0"
  try:
    xd = torch.from_numpy(np.ascontiguousarray(w, np.float32).reshape(-1)).cuda()
    clip, iters = ops.octav_clip(xd, w.size // unit_len, unit_len, bits, 10, 3.0, early_stop)
    torch.cuda.synchronize()
    return clip.cpu().numpy(), int(iters.cpu().item())
  finally:
    if old is None:
      del os.environ["MI355Q_OCTAV_UNIT_LANES"]
    else:
      os.environ["MI355Q_OCTAV_UNIT_LANES"] = old


def _same(a, b):
  return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)) or \
      np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _designed(rng, units, unit_len):
  """Units whose selected sets (at the guesses the iteration visits) are runs of chosen lengths: small magnitudes as
  background, stretches of large same-signed values on top."""
  w = (rng.standard_normal((units, unit_len)) * 0.01).astype(np.float32)
  lengths = [1, 2, 6, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, unit_len - 1, unit_len]
  for u in range(units):
    n = min(lengths[u % len(lengths)], unit_len)
    where = [0, unit_len - n, (unit_len - n) // 2][(u // len(lengths)) % 3]      # touching the start, the end, inside
    sign = -1.0 if (u // 7) % 2 else 1.0
    w[u, where:where + n] = sign * (1.0 + rng.random(n).astype(np.float32)) * np.float32(3.0 if u % 5 else 0.3)
    if u % 11 == 0 and where + n + 2 < unit_len:                                 # a second run right behind a one-element gap
      w[u, where + n + 1:where + n + 2 + (u % 9)] = sign * np.float32(2.5)
  return w


@pytest.mark.parametrize("unit_len", [32, 64, 128, 256])
@pytest.mark.parametrize("bits", [4, 8, 2])
def test_designed_runs_against_the_oracle(g, unit_len, bits):
  rng = np.random.default_rng(unit_len * 10 + bits)
  w = _designed(rng, 200, unit_len)          # 200 units: the last wave has lanes without a unit
  for early in (True, False):
    ref, ref_iters = O.octav_clip(w, bits, (1,), 10, 3.0, early_stop=early, return_iters=True)
    got, iters = _clip(g, w, unit_len, bits, early)
    assert _same(got, ref.reshape(-1)), (unit_len, bits, early)
    assert iters == ref_iters


@pytest.mark.parametrize("unit_len", [32, 64, 128, 256])
@pytest.mark.parametrize("kind", ["weights", "unit_normal", "same_sign", "zeros_and_specials", "constant"])
def test_random_layouts_against_the_oracle_and_the_groups_kernel(g, unit_len, kind):
  rng = np.random.default_rng(hash((unit_len, kind)) % (1 << 31))
  units = 4096 * 128 // unit_len // 8          # 512 x 128 elements' worth: a multiple of the groups kernel's 4096
  if kind == "weights":
    w = (rng.standard_normal((units, unit_len)) * 0.02).astype(np.float32)
  elif kind == "unit_normal":
    w = rng.standard_normal((units, unit_len)).astype(np.float32)
  elif kind == "same_sign":                    # every element in one mask: one run as long as the unit at guess 0
    w = np.abs(rng.standard_normal((units, unit_len))).astype(np.float32) * np.float32(0.05)
    w[1::2] *= -1
  elif kind == "zeros_and_specials":
    w = (rng.standard_normal((units, unit_len)) * 0.02).astype(np.float32)
    w[rng.random(w.shape) < 0.3] = 0.0
    w[rng.random(w.shape) < 0.05] = -0.0
    w[3, 5] = np.nan
    w[7, 0] = np.inf
    w[9, unit_len - 1] = -np.inf
    w[11, :] = 0.0
  else:
    w = np.full((units, unit_len), 0.37, np.float32)
    w[::3] = -0.11
  with np.errstate(all="ignore"):
    ref, ref_iters = O.octav_clip(w, 4, (1,), 10, 3.0, return_iters=True)
  got, iters = _clip(g, w, unit_len, 4)
  old, old_iters = _clip(g, w, unit_len, 4, lanes=False)
  assert _same(got, old) and iters == old_iters
  assert _same(got, ref.reshape(-1)) and iters == ref_iters


def test_layer_sized_blockwise_weight_against_the_groups_kernel(g):
  """4096 x 4096 in blocks of 128 / 32 (the bench's shape), sigma 0.02 and 1.0: 131 072 / 524 288 units, both kernels."""
  torch = g["torch"]
  gen = torch.Generator(device="cuda").manual_seed(99)
  for sigma in (0.02, 1.0):
    w = (torch.randn((4096, 4096), generator=gen, device="cuda") * sigma).cpu().numpy()
    for unit_len in (256, 128, 32):
      got, iters = _clip(g, w, unit_len, 4)
      old, old_iters = _clip(g, w, unit_len, 4, lanes=False)
      assert _same(got, old) and iters == old_iters, (sigma, unit_len)


def test_public_call_takes_the_new_kernel_and_matches_the_oracle(g):
  """octav.get_tensor_quant_params, BLOCKWISE_64 int4, against the oracle's parameters and integers."""
  from mi355q import qtyping
  from mi355q.algorithms.uniform_quantize import octav
  rng = np.random.default_rng(5)
  w = (rng.standard_normal((96, 256)) * 0.03).astype(np.float32)
  cfg = qtyping.TensorQuantizationConfig(num_bits=4, symmetric=True, granularity=qtyping.QuantGranularity.BLOCKWISE_64)
  info = qtyping.OpInfo(op=qtyping.OperatorT(), op_name=qtyping.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                        op_quant_config=qtyping.OpQuantizationConfig(weight_tensor_config=cfg))
  p = octav.get_tensor_quant_params(info, cfg, w)
  ref = O.octav_quant_params(w, 4, "BLOCKWISE_64")
  assert np.array_equal(np.asarray(p.scale), ref["scale"])
  assert np.array_equal(np.asarray(p.quantized_data), ref["quantized_data"])


@pytest.mark.parametrize("unit_len", [32, 64, 128, 256])
def test_a_million_mixed_units_against_the_groups_kernel(g, unit_len):
  """Both kernels on 2^24 elements of units drawn from very different laws -- Gaussian at several scales, heavy tails, a few
  huge outliers on a tiny background (iterates that overshoot and come back down), sparse units, two-valued units, long
  same-signed stretches: every clipping constant and the iteration count, bit for bit. What this sweeps that the designed
  cases cannot: the switch to the candidate lists (and back) at whatever iteration each wave takes it."""
  rng = np.random.default_rng(7 + unit_len)
  units = (1 << 24) // unit_len
  w = rng.standard_normal((units, unit_len)).astype(np.float32)
  kind = rng.integers(0, 8, size=units)
  scale = np.float32(10.0) ** rng.integers(-3, 2, size=units).astype(np.float32)
  w *= scale[:, None]
  heavy = kind == 1
  w[heavy] = (w[heavy] * np.exp(rng.standard_normal(w[heavy].shape) * 1.5)).astype(np.float32)
  outl = kind == 2
  w[outl] *= np.float32(1e-3)
  pick = rng.random(w.shape) < 0.03
  w[outl[:, None] & pick] *= np.float32(3e3)
  sparse = kind == 3
  w[sparse[:, None] & (rng.random(w.shape) < 0.9)] = 0.0
  two = kind == 4
  w[two] = np.where(rng.random(w[two].shape) < 0.1, np.float32(1.7), np.float32(-0.02))
  runs = kind == 5
  w[runs] = np.abs(w[runs])
  flip = rng.random(w.shape) < 0.08
  w[runs[:, None] & flip] *= -1
  got, iters = _clip(g, w, unit_len, 4)
  old, old_iters = _clip(g, w, unit_len, 4, lanes=False)
  assert iters == old_iters
  same = got.view(np.uint32) == old.view(np.uint32)
  assert same.all(), (int((~same).sum()), np.flatnonzero(~same)[:8], kind[np.flatnonzero(~same)[:8]])


@pytest.mark.parametrize("size", [32, 64, 128, 256])
@pytest.mark.parametrize("max_iter", [1, 3, 10, 25])
def test_one_unit_tensorwise_and_other_iteration_counts(g, size, max_iter):
  """A whole tensor of 32 .. 256 elements as ONE unit (TENSORWISE: the reference's count is a Python int there, s * N stays
  float32 -- ref octav.py:55-61) and iteration limits other than the recipe's ten."""
  torch, ops = g["torch"], g["ops"]
  rng = np.random.default_rng(size + max_iter)
  w = (rng.standard_normal(size) * 0.7).astype(np.float32)
  for early in (True, False):
    ref, ref_iters = O.octav_clip(w, 4, None, max_iter, 3.0, early_stop=early, return_iters=True)
    clip, iters = ops.octav_clip(torch.from_numpy(w).cuda(), 1, size, 4, max_iter, 3.0, early, False)
    torch.cuda.synchronize()
    assert np.array_equal(clip.cpu().numpy().view(np.uint32), np.asarray(ref, np.float32).reshape(-1).view(np.uint32))
    assert int(iters.cpu().item()) == ref_iters
