"""Stand-in for `ml_dtypes` limited to what the reference's hot path touches:
`x.astype(ml_dtypes.bfloat16).astype(np.float16)`.

bfloat16 conversion is IEEE round-to-nearest-even on the upper 16 bits of the
float32 pattern (ml_dtypes' documented behaviour; cross-checked against
torch.bfloat16 in tests/golden/gen/make_golden.py). Inputs that must take this
path are fed as `Bf16Aware` arrays so the reference source stays unpatched.
"""
import numpy as np


class _Bf16Token:
  def __repr__(self):
    return "bfloat16"


bfloat16 = _Bf16Token()


def round_to_bf16(x):
  """float32 -> nearest-even bfloat16, returned widened back to float32."""
  x = np.ascontiguousarray(np.asarray(x).view(np.ndarray), dtype=np.float32)
  bits = x.view(np.uint32)
  lsb = (bits >> np.uint32(16)) & np.uint32(1)
  out = ((bits + np.uint32(0x7FFF) + lsb) & np.uint32(0xFFFF0000)).view(np.float32).copy()
  out[np.isnan(x)] = np.nan
  return out


class Bf16Aware(np.ndarray):
  """ndarray whose .astype(bfloat16) applies round_to_bf16 (kept as float32)."""

  def astype(self, dtype, *a, **k):
    if dtype is bfloat16:
      return round_to_bf16(self).view(Bf16Aware)
    return np.ndarray.astype(self, dtype, *a, **k)
