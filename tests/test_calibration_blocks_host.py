"""The host half of K-samples-per-launch calibration, without a GPU.

Calibrator.replay advances StepBlocks as array expressions; the equality with the event-by-event replay (the update
rules of ref utils/qsv_utils.py:43-122 applied in dataset order, ref calibrator.py:395-421) is a pure NumPy matter. The
gather / record side runs here too, with the launch replaced by the oracle's min / max of the very arrays the pointer
table names (`host_launch`, also used by the world-4 / world-8 gloo tests): what is covered is which sample lands in
which block, in which order, under which tag -- the kernel itself is covered by the `-m gpu` tests.
"""
import contextlib
import os
import sys

import numpy as np
import pytest

from oracle import aeq_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@contextlib.contextmanager
def host_launch():
  """Inside: the calibrator's block path reads host arrays (float32 ndarrays only), one `launch` per block."""
  import torch
  from mi355q import calibrator, ops, runtime as rt
  live = {}
  launches = []

  def describe(v):
    if not isinstance(v, np.ndarray) or v.dtype != np.float32 or not v.size:
      return None
    key = len(live) + 1
    live[key] = v
    return [v, torch.from_numpy(np.ascontiguousarray(v)), key, v.size, v.shape[0] if v.ndim else 1, v.ndim, None, 0]

  def entries(pointers, lengths, lo, hi):
    launches.append(len(pointers))
    out = np.empty((len(pointers), 2), np.float32)
    for i, (p, n) in enumerate(zip(pointers, lengths)):
      assert live[p].size == n
      r = O.activation_min_max(live[p], lo, hi)
      out[i] = np.ravel(r["min"])[0], np.ravel(r["max"])[0]
    return torch.from_numpy(out)

  class Event:
    def record(self): pass
    def synchronize(self): pass
  saved = (calibrator._describe, ops.act_minmax_entries, rt.require_gpu, calibrator.Calibrator._pinned_results,
           torch.cuda.Event)
  calibrator._describe, ops.act_minmax_entries, rt.require_gpu = describe, entries, lambda: None
  calibrator.Calibrator._pinned_results = lambda self, shape, dtype: torch.empty(shape, dtype=dtype)
  torch.cuda.Event = Event
  try:
    yield launches
  finally:
    (calibrator._describe, ops.act_minmax_entries, rt.require_gpu, calibrator.Calibrator._pinned_results,
     torch.cuda.Event) = saved


def _samples(n, tensors=5, seed=1):
  rng = np.random.default_rng(seed)
  return [{f"act{i}": rng.standard_normal((1 + (s + i) % 3, 4, 8)).astype(np.float32) * np.float32(1 + i)
           for i in range(tensors)} for s in range(n)]


def _rm(rcp):
  from mi355q import recipe_manager
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(rcp)
  return rm


def _reference_qsvs(samples):
  ref = {}
  for s in samples:
    for name, x in s.items():
      ref[name] = O.moving_average_update(ref.get(name), O.activation_min_max(x, -3e38, 3e38))
  return ref


@pytest.mark.parametrize("k", [2, 3, 17, 64])
def test_blocks_of_any_size_replay_to_the_oracles_chain(k):
  import c4_bench
  from mi355q import calibrator, recipe
  model = c4_bench.build_model(5, 8, 4)
  samples = _samples(17)
  with host_launch() as launches:
    cal = calibrator.Calibrator(model)
    cal.calibrate({"serving_default": samples}, _rm(recipe.static_wi8_ai8()), samples_per_launch=k)
  assert launches == [5 * min(k, 17 - i) for i in range(0, 17, k)]
  ref = _reference_qsvs(samples)
  got = cal.get_model_qsvs()
  assert set(got) == set(ref) and cal._metadata == {"num_samples_calibrated": 17}
  for name in ref:
    assert set(got[name]) == {"min", "max"}
    for key in ("min", "max"):
      assert got[name][key].dtype == np.float32 and got[name][key].shape == (1, 1, 1)
      assert got[name][key].tobytes() == np.asarray(ref[name][key], np.float32).tobytes()


def test_block_replay_equals_event_replay_under_every_rule():
  """Stock rules advance as arrays, anything else sample by sample: "ema" (min / max), "count" (+ num_samples beside
  a kept Hessian entry), a user's rule, and QSVs that are not calibration's own float32 pairs."""
  from mi355q import calibrator, qtyping
  from mi355q.utils import qsv_utils
  rng = np.random.default_rng(3)
  op = qtyping.TFLOperationName.FULLY_CONNECTED.value
  slots = (("a", "min_max_uniform_quantize", op), ("b", "GPTQ", op), ("c", "tagged", op), ("d", "min_max_uniform_quantize", op),
           ("e", "GPTQ", op))
  ndims = (3, 2, 1, 0, 2)

  def block(k, first):
    stats = rng.standard_normal((k, 5, 2)).astype(np.float32)
    stats[..., 0] = -np.abs(stats[..., 0])
    if k > 2:
      stats[1, 0, 1] = np.inf
      stats[2, 3, 0] = np.nan
    return calibrator.StepBlock(slots, stats, rng.integers(1, 5, (k, 5)), ndims, {2: 64}, first)
  blocks = [block(1, 0), block(7, 1), block(2, 8), block(1, 10)]

  def tagged(qsv, new):            # (what distributed._ema_and_count_update does)
    out = qsv_utils.moving_average_update(qsv, new)
    out["num_samples"] = qsv["num_samples"] + new["num_samples"]
    for key in ("hessian", "hessian_dim"):
      if key in qsv:
        out[key] = qsv[key]
    return out
  tagged_fast = lambda qsv, new: tagged(qsv, new)
  tagged_fast.block_mode = "count"

  def run(as_blocks, overrides, custom=None, loaded=None):
    cal = calibrator.Calibrator(_tiny_model(), **({"qsv_update_func": custom} if custom else {}))
    if loaded:
      cal.load_model_qsvs(loaded)
    if as_blocks:
      cal.replay(blocks, overrides)
    else:
      cal.replay((b.events(k) for b in blocks for k in range(len(b))), overrides)
    return cal
  loaded = {"a": {"min": np.array([[[-1.0]]], np.float32), "max": np.array([[[2.0]]], np.float32)},
            "b": {"min": np.array([[-3.0]]), "max": np.array([[0.5]]), "num_samples": 3, "hessian": np.eye(2)},   # float64
            "e": {"min": np.array([[-3.0]], np.float32), "max": np.array([[0.5]], np.float32), "num_samples": np.array(2),
                  "hessian": np.eye(2)}}
  for overrides, custom, pre in (({"tagged": tagged_fast}, None, None), ({"tagged": tagged}, None, None),
                                 ({"tagged": tagged_fast}, None, loaded),
                                 ({"tagged": tagged_fast}, qsv_utils.min_max_update, None)):
    a, b = run(True, overrides, custom, pre), run(False, overrides, custom, pre)
    assert a._metadata == b._metadata == {"num_samples_calibrated": 11}
    qa, qb = a.get_model_qsvs(), b.get_model_qsvs()
    assert set(qa) == set(qb) == set("abcde")
    for name in qb:
      assert set(qa[name]) == set(qb[name]), (name, sorted(qa[name]), sorted(qb[name]))
      for key, want in qb[name].items():
        got = qa[name][key]
        if key in ("min", "max"):
          assert np.asarray(got).dtype == np.asarray(want).dtype and np.asarray(got).shape == np.asarray(want).shape
          assert np.asarray(got).tobytes() == np.asarray(want).tobytes(), (name, key)
        elif key == "hessian":
          assert got is want or np.array_equal(got, want)
        else:
          assert int(got) == int(want), (name, key)


def _tiny_model():
  import c4_bench
  return c4_bench.build_model(2, 8, 4)


def test_misfits_and_failures_keep_the_per_sample_order():
  """A sample the block path does not cover (here: a float64 tensor) is walked on its own between two blocks; the
  walk's own calibration needs a GPU, so the fallback is observed, not executed."""
  import c4_bench
  from mi355q import calibrator, recipe
  model = c4_bench.build_model(3, 8, 4)
  rm = _rm(recipe.static_wi8_ai8())
  samples = _samples(9, 3, seed=4)
  samples[4]["act1"] = samples[4]["act1"].astype(np.float64)
  order = []
  with host_launch() as launches:
    cal = calibrator.Calibrator(model)
    with cal.plan_once():
      for first, item in cal.record_blocks("serving_default", iter(samples), rm, 3,
                                           fallback=lambda data: ("walked", data is samples[4]),
                                           taken_up=lambda n: order.append(n)):
        order.append((first, len(item) if isinstance(item, calibrator.StepBlock) else item))
  assert order == [3, (0, 3), 1, (3, 1), 1, (4, ("walked", True)), 3, (5, 3), 1, (8, 1)]
  assert launches == [9, 3, 9, 3]

  def broken():
    yield from samples[:4]
    raise RuntimeError("the dataset broke")
  got = []
  with host_launch():
    with pytest.raises(RuntimeError, match="the dataset broke"):
      for first, item in cal.record_blocks("serving_default", broken(), rm, 3):
        got.append((first, len(item)))
  assert got == [(0, 3), (3, 1)]          # the samples pulled before the failure were recorded first
