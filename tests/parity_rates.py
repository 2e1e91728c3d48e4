"""T2 bookkeeping: every tolerance-class comparison states what it observed.

SURVEY section 7 (two-tier parity contract): where the reference itself runs sgemm / LAPACK, the
integers may differ by one step on a small fraction of the elements. A test that only asserts a
bound hides whether the observed rate is 1e-6 or 4e-3, so each such test goes through `check()`:
it prints the observed mismatch fraction, appends it to `gpurun_out/parity_rates.jsonl` (merged
back from the GPU box; the summary is committed as profiles/r02_parity_rates.txt) and asserts
the bound, which is kept at <= 10 x the recorded observation (never below the 1e-5 contract).
"""
import json
import os

import numpy as np

T2 = 1e-5     # at most 1e-5 of the integers may differ, by one step

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OUT = os.path.join(_ROOT, "gpurun_out", "parity_rates.jsonl")


def _append(rec: dict) -> None:
  print("PARITY_RATE " + json.dumps(rec))
  try:
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    with open(_OUT, "a") as f:
      f.write(json.dumps(rec) + "\n")
  except OSError:
    pass


def check(name: str, got, ref, bound: float, max_step: int = 1) -> float:
  """Integer buffers: fraction of differing elements <= bound, no difference beyond one step."""
  got, ref = np.asarray(got), np.asarray(ref)
  assert got.shape == ref.shape, (got.shape, ref.shape)
  diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
  frac = float((diff != 0).mean()) if diff.size else 0.0
  worst = int(diff.max()) if diff.size else 0
  _append({"test": name, "kind": "int_mismatch_fraction", "observed": frac, "max_step": worst,
           "bound": bound, "elements": int(diff.size)})
  assert worst <= max_step, f"{name}: a value differs by {worst} steps"
  assert frac <= bound, f"{name}: {frac:.3e} of the integers differ (bound {bound:.1e})"
  return frac


def check_rel(name: str, got, ref, bound: float) -> float:
  """Floating-point results: max |got - ref| / max |ref| <= bound."""
  got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
  err = float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-300)) if ref.size else 0.0
  _append({"test": name, "kind": "max_rel_error", "observed": err, "bound": bound,
           "elements": int(ref.size)})
  assert err <= bound, f"{name}: relative error {err:.3e} (bound {bound:.1e})"
  return err


def note(name: str, kind: str, observed: float, bound: float, **extra) -> None:
  """A rate measured by the test itself (e.g. on the device)."""
  _append({"test": name, "kind": kind, "observed": float(observed), "bound": bound, **extra})
  assert observed <= bound, f"{name}: {kind} {observed:.3e} (bound {bound:.1e})"
