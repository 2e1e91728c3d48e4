"""The GPU OCTAV / MSE kernels claim bit-exactness by reproducing NumPy's float32
summation order. This pins that order model against NumPy itself (CPU only)."""
import numpy as np
import pytest

from numpy_sum_model import masked_sum, pairwise, plain_sum


@pytest.mark.parametrize("n", [5, 64, 300, 4096, 8192, 9000, 20000])
@pytest.mark.parametrize("thr", [0.0, 1.0, 2.5, -10.0])
def test_masked_row_sum_order(n, thr):
  rng = np.random.default_rng(n)
  x = rng.standard_normal((2, n)).astype(np.float32)
  mask = x >= np.float32(thr)
  got = np.sum(x, axis=1, where=mask, keepdims=True, dtype=np.float32)
  for r in range(2):
    assert got[r, 0] == masked_sum(x[r], mask[r])


def test_masked_sum_tensorwise_and_3d_units():
  rng = np.random.default_rng(7)
  x = rng.standard_normal((7, 3001)).astype(np.float32)
  mask = np.ones_like(x, dtype=bool)
  mask[0, 3] = False
  got = np.sum(x, axis=None, where=mask, keepdims=True, dtype=np.float32)
  assert got.reshape(-1)[0] == masked_sum(x.reshape(-1), mask.reshape(-1))
  y = rng.standard_normal((2, 30, 500)).astype(np.float32)
  m = y >= np.float32(-0.3)
  got = np.sum(y, axis=(1, 2), where=m, keepdims=True, dtype=np.float32)
  for r in range(2):
    assert got[r, 0, 0] == masked_sum(y[r].reshape(-1), m[r].reshape(-1))
  z = rng.standard_normal((5, 4, 128)).astype(np.float32)
  mz = z <= np.float32(-0.5)
  got = np.sum(z, axis=2, where=mz, keepdims=True)
  assert all(got[r, b, 0] == masked_sum(z[r, b], mz[r, b]) for r in range(5) for b in range(4))


@pytest.mark.parametrize("n", [5, 100, 300, 4096, 5000, 11008, 20000])
def test_plain_row_sum_and_mean_order(n):
  rng = np.random.default_rng(n + 1)
  x = rng.standard_normal((2, n)).astype(np.float32)
  sq = x**2
  assert np.array_equal(sq, x * x)
  s = np.sum(sq, axis=1, keepdims=True)
  m = np.mean(sq, axis=1, keepdims=True)
  for r in range(2):
    assert s[r, 0] == plain_sum(sq[r])
    assert m[r, 0] == np.float32(s[r, 0] / np.float32(n))
  if n <= 8192:
    assert s[0, 0] == pairwise(sq[0])
