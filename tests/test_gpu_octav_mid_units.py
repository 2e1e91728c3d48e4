"""OCTAV on units of 129 .. 1023 elements (rows of small transformers' projections: 384, 512, 640, 768 ...): since round 6 the
rows kernel (a workgroup per unit) takes them; the one-wave-per-unit kernel (MI355Q_OCTAV_WAVE_KERNEL=1) took them before.
Both against the oracle's NumPy iteration (ref octav.py:30-112) and against each other, bit for bit."""
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
  from mi355q import ops
  return dict(torch=torch, ops=ops)


def _clip(g, w, bits, wave_kernel=False, early=True):
  torch, ops = g["torch"], g["ops"]
  if wave_kernel:
    os.environ["MI355Q_OCTAV_WAVE_KERNEL"] = "1"
  try:
    clip, iters = ops.octav_clip(torch.from_numpy(np.ascontiguousarray(w).reshape(-1)).cuda(), w.shape[0], w.shape[1], bits, 10, 3.0, early)
    torch.cuda.synchronize()
    return clip.cpu().numpy(), int(iters.cpu().item())
  finally:
    os.environ.pop("MI355Q_OCTAV_WAVE_KERNEL", None)


def _layout(rng, kind, units, n):
  if kind == "weights":
    return (rng.standard_normal((units, n)) * 0.02).astype(np.float32)
  if kind == "unit_normal":
    return rng.standard_normal((units, n)).astype(np.float32)
  if kind == "same_sign":                        # one run as long as the unit at guess 0: NumPy's pairwise recursion
    w = np.abs(rng.standard_normal((units, n))).astype(np.float32) * np.float32(0.05)
    w[1::2] *= -1
    return w
  if kind == "runs":                             # stretches of 7 / 8 / 9 / 127 / 128 / 129 large values on a small background
    w = (rng.standard_normal((units, n)) * 0.01).astype(np.float32)
    for u in range(units):
      k = min([7, 8, 9, 127, 128, 129, 200][u % 7], n)
      at = [0, n - k, (n - k) // 3][(u // 7) % 3]
      w[u, at:at + k] = (1.0 + rng.random(k)) * (-1.0 if u % 2 else 1.0)
    return w
  w = (rng.standard_normal((units, n)) * 0.02).astype(np.float32)       # zeros and specials
  w[rng.random(w.shape) < 0.3] = 0.0
  w[rng.random(w.shape) < 0.05] = -0.0
  w[3, 5] = np.nan
  w[7, 0] = np.inf
  w[9, n - 1] = -np.inf
  w[11, :] = 0.0
  return w


@pytest.mark.parametrize("n", [129, 130, 144, 200, 250, 257, 300, 384, 512, 640, 768, 896, 1000, 1023])
@pytest.mark.parametrize("kind", ["weights", "unit_normal", "same_sign", "runs", "zeros_and_specials"])
def test_mid_sized_units_against_the_oracle_and_the_wave_kernel(g, n, kind):
  rng = np.random.default_rng(n * 7 + len(kind))
  w = _layout(rng, kind, 150, n)
  with np.errstate(all="ignore"):
    ref, ref_iters = O.octav_clip(w, 4, (1,), 10, 3.0, return_iters=True)
  got, iters = _clip(g, w, 4)
  old, old_iters = _clip(g, w, 4, wave_kernel=True)
  assert np.array_equal(got.view(np.uint32), old.view(np.uint32)) and iters == old_iters
  assert np.array_equal(got, ref.reshape(-1), equal_nan=True) and iters == ref_iters


@pytest.mark.parametrize("n", [384, 768])
def test_layer_sized_projection_against_the_wave_kernel(g, n):
  """[3072, 768]-like projections: 2^22 elements, both kernels, every constant and the iteration count."""
  torch = g["torch"]
  w = (torch.randn(((1 << 22) // n, n), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda") * 0.02).cpu().numpy()
  got, iters = _clip(g, w, 4)
  old, old_iters = _clip(g, w, 4, wave_kernel=True)
  assert np.array_equal(got.view(np.uint32), old.view(np.uint32)) and iters == old_iters
