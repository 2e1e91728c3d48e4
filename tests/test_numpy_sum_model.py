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


@pytest.mark.parametrize("shape", [(3, 4096), (2, 5, 7, 130), (5000, 300), (9, 7), (1, 3, 3, 64),
                                   (8193, 3), (20000, 2), (300, 2, 3), (64, 3, 3, 16), (17, 1)])
def test_channel_last_reductions_are_row_by_row(shape):
  """Reducing every axis but the last (DEPTHWISE_CONV_2D / BATCH_MATMUL weights)."""
  from numpy_sum_model import column_sums
  rng = np.random.default_rng(len(shape) * 1000 + shape[0])
  x = rng.standard_normal(shape).astype(np.float32) * np.float32(3.0)
  ax = tuple(range(len(shape) - 1))
  x2 = x.reshape(-1, shape[-1])
  for thr in (0.3, 1.5):
    for mask in (x >= np.float32(thr), x <= np.float32(-thr)):
      got = np.sum(x, axis=ax, where=mask, keepdims=True, dtype=np.float32).reshape(-1)
      if shape[-1] == 1:        # NumPy drops the size-1 axis: this is a contiguous unit again
        assert got[0] == masked_sum(x.reshape(-1), mask.reshape(-1))
      else:
        assert np.array_equal(got, column_sums(x2, mask.reshape(x2.shape)))
  if shape[-1] > 1:
    sq = x**2
    assert np.array_equal(np.sum(sq, axis=ax).reshape(-1), column_sums(sq.reshape(x2.shape)))
    assert np.array_equal(np.mean(sq, axis=ax).reshape(-1),
                          (column_sums(sq.reshape(x2.shape)) / np.float32(x2.shape[0])).astype(np.float32))


@pytest.mark.parametrize("shape", [(2, 16, 8), (3, 5, 100), (4, 3, 5000), (2, 2, 10000), (5, 7, 3),
                                   (3, 4, 129), (2, 3, 8192), (3, 2, 9000), (6, 200, 9), (2, 4, 2, 8)])
def test_mixed_reductions(shape):
  """A middle axis kept (BATCH_MATMUL right-hand side with adj_y): segments add up in order."""
  from numpy_sum_model import segment_sums
  rng = np.random.default_rng(sum(shape))
  x = rng.standard_normal(shape).astype(np.float32)
  ax = (0,) + tuple(range(2, len(shape)))
  x3 = x.reshape(shape[0], shape[1], -1)
  for thr in (0.2, 1.0):
    mask = x >= np.float32(thr)
    got = np.sum(x, axis=ax, where=mask, keepdims=True, dtype=np.float32).reshape(-1)
    assert np.array_equal(got, segment_sums(x3, mask.reshape(x3.shape)))
  sq = x**2
  assert np.array_equal(np.sum(sq, axis=ax).reshape(-1), segment_sums(sq.reshape(x3.shape)))
  n = np.float32(x3.shape[0] * x3.shape[2])
  assert np.array_equal(np.mean(sq, axis=ax).reshape(-1),
                        (segment_sums(sq.reshape(x3.shape)) / n).astype(np.float32))
