"""Pure-Python model of the order in which NumPy adds float32 values in
np.sum(x, axis, where=mask) / np.sum(x, axis) -- the order libmi355q's
reduce_exact.hip reproduces on the GPU. test_numpy_sum_model.py proves the
model against NumPy itself."""
import numpy as np

F = np.float32
CHUNK = 8192  # nditer buffer size


def pairwise(a):
  """NumPy's pairwise_sum for a contiguous float32 run."""
  n = len(a)
  if n < 8:
    res = F(0.0)
    for v in a:
      res = F(res + v)
    return res
  if n <= 128:
    r = [F(a[k]) for k in range(8)]
    i = 8
    while i + 8 <= n:
      for k in range(8):
        r[k] = F(r[k] + a[i + k])
      i += 8
    res = F(F(F(r[0] + r[1]) + F(r[2] + r[3])) + F(F(r[4] + r[5]) + F(r[6] + r[7])))
    while i < n:
      res = F(res + a[i])
      i += 1
    return res
  n2 = n // 2
  n2 -= n2 % 8
  return F(pairwise(a[:n2]) + pairwise(a[n2:]))


def masked_sum(x, mask):
  """sum of x[mask] for one reduction unit, in NumPy's order."""
  acc = F(0.0)
  n = len(x)
  i = 0
  while i < n:
    if not mask[i]:
      i += 1
      continue
    j = i
    while j < n and mask[j] and j // CHUNK == i // CHUNK:
      j += 1
    acc = F(acc + pairwise(x[i:j]))
    i = j
  return acc


def plain_sum(x):
  acc = None
  for i in range(0, len(x), CHUNK):
    p = pairwise(x[i:i + CHUNK])
    acc = p if acc is None else F(acc + p)
  return acc


def column_sums(x2, mask2=None):
  """np.sum(x, axis=<all leading axes>[, where=mask]) of a C-contiguous array viewed as
  [outer, channels]: NumPy walks it row by row, so each channel is a left-to-right float32
  sum over `outer` (masked-out elements skipped)."""
  acc = np.zeros(x2.shape[1], F)
  for o in range(x2.shape[0]):
    nxt = (acc + x2[o]).astype(F)
    acc = nxt if mask2 is None else np.where(mask2[o], nxt, acc)
  return acc


def segment_sums(x3, mask3=None):
  """np.sum(x, axis=(0, 2)[, where=mask]) of a C-contiguous [outer, channels, inner] array:
  one running total per channel; every segment x[o, c, :] is handed to the inner loop on its
  own (8192-element chunks and runs restart at the segment start) and added in order."""
  outer, channels, inner = x3.shape
  out = np.zeros(channels, F)
  for c in range(channels):
    acc = None
    for o in range(outer):
      seg = x3[o, c]
      if mask3 is None:
        for i in range(0, inner, CHUNK):
          p = pairwise(seg[i:i + CHUNK])
          acc = p if acc is None else F(acc + p)
      else:
        acc = F(0.0) if acc is None else acc
        msk = mask3[o, c]
        i = 0
        while i < inner:
          if not msk[i]:
            i += 1
            continue
          j = i
          while j < inner and msk[j] and j // CHUNK == i // CHUNK:
            j += 1
          acc = F(acc + pairwise(seg[i:j]))
          i = j
    out[c] = acc
  return out
