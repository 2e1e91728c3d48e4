"""Helpers shared by the oracle and GPU parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a) -> str:
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_names(algo: str):
  with open(os.path.join(GOLDEN, "ref_cases.json")) as f:
    return [c["name"] for c in json.load(f)["cases"] if c["algo"] == algo]


def num(v, val=None):
  """JSON scalars: 'inf' / '-inf' / 'val' / '-val' placeholders."""
  if isinstance(v, str):
    table = {"inf": np.inf, "-inf": -np.inf}
    if val is not None:
      table.update({"val": val, "-val": -val})
    return table[v]
  return v


def gen_c2(seed=1234, shape=(4096, 4096)):
  return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def gen_c3(layer: int, shape=(4096, 11008)):
  return np.random.default_rng(1000 + layer).standard_normal(
      shape, dtype=np.float32) * np.float32(0.02)
