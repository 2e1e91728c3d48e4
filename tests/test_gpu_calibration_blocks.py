"""K calibration samples per launch (Calibrator.record_blocks / StepBlock) against the per-sample walk.

The per-sample walk mirrors ref calibrator.py:312-331, 500-587 op by op; the block path replaces K of its steps with
one launch and an array replay. Everything here is an equality: QSVs (min / max bits, key sets, sample counts),
GPTQ Hessians (the float64 array, bit for bit) and the state an exception leaves behind.
"""
import os
import sys

import numpy as np
import pytest

from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  import c4_bench
  import c5_model
  from mi355q import calibrator, distributed, recipe, recipe_manager
  from mi355q.algorithms.uniform_quantize import gptq
  return types.SimpleNamespace(torch=torch, cal=calibrator, dist=distributed, recipe=recipe, rm=recipe_manager,
                               c4=c4_bench, c5=c5_model, gptq=gptq)


def _rm(m, rcp):
  rm = m.rm.RecipeManager()
  rm.load_quantization_recipe(rcp)
  return rm


def _same_qsvs(got, want, hessians=True):
  assert set(got) == set(want)
  for name in want:
    assert set(got[name]) == set(want[name]), (name, sorted(got[name]), sorted(want[name]))
    for key in ("min", "max"):
      a, b = np.asarray(got[name][key]), np.asarray(want[name][key])
      assert a.dtype == b.dtype and a.shape == b.shape, (name, key)
      assert a.tobytes() == b.tobytes(), (name, key, a, b)
    if "num_samples" in want[name]:
      assert int(got[name]["num_samples"]) == int(want[name]["num_samples"]), name
    if hessians and "hessian" in want[name]:
      a, b = np.asarray(got[name]["hessian"]), np.asarray(want[name]["hessian"])
      assert a.dtype == b.dtype == np.float64 and a.tobytes() == b.tobytes(), name


def _c4_samples(n, tensors=8, seq=16, width=512, seed=44):
  rng = np.random.default_rng(seed)
  out = []
  for s in range(n):
    d = {}
    for i in range(tensors):
      x = rng.standard_normal((1, seq, width), dtype=np.float32) * np.float32(1 + i / 8)
      if (s * tensors + i) % 13 == 0:
        x.reshape(-1)[:4] = [np.inf, -np.inf, 3.39e38, -3.39e38]
      if (s * tensors + i) % 29 == 5:
        x.reshape(-1)[7] = np.nan
      d[f"act{i}"] = x
    out.append(d)
  return out


def _walk(m, model, rm, samples, k, **kw):
  cal = m.cal.Calibrator(model, **kw)
  cal.calibrate({"serving_default": samples}, rm, samples_per_launch=k)
  return cal


@pytest.mark.parametrize("k", [2, 5, 21, 64, None])
def test_any_block_size_gives_the_per_sample_walks_qsvs(m, k):
  model = m.c4.build_model(8, 512, 16)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(21)
  want = _walk(m, model, rm, samples, 1)
  calls = []
  real = m.cal.Calibrator._gather_block
  m.cal.Calibrator._gather_block = lambda self, *a, **kw: calls.append(1) or real(self, *a, **kw)
  try:
    got = _walk(m, model, rm, samples, k)
  finally:
    m.cal.Calibrator._gather_block = real
  assert calls, "the block path did not run"
  _same_qsvs(got.get_model_qsvs(), want.get_model_qsvs())
  assert got._metadata == want._metadata == {"num_samples_calibrated": 21}
  # ... and the oracle's replay of the oracle's statistics (ref common_quantize.py:1362-1413, qsv_utils.py:43-68)
  ref = {}
  for s in samples:
    for name, x in s.items():
      ref[name] = O.moving_average_update(ref.get(name), O.activation_min_max(x, -3e38, 3e38))
  for name in ref:
    for key in ("min", "max"):
      assert np.asarray(got.get_model_qsvs()[name][key]).tobytes() == np.asarray(ref[name][key], np.float32).tobytes()


def test_one_sample_and_blocks_on_top_of_a_loaded_result(m):
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(7, 4, 8, 256)
  # a single sample: the QSV is the event itself (num_samples stays beside min / max)
  _same_qsvs(_walk(m, model, rm, samples[:1], 8).get_model_qsvs(), _walk(m, model, rm, samples[:1], 1).get_model_qsvs())
  # resumed: the first three samples' result is loaded, the rest is calibrated on top of it
  first = _walk(m, model, rm, samples[:3], 1).get_model_qsvs()
  outs = []
  for k in (1, 4):
    cal = m.cal.Calibrator(model)
    cal.load_model_qsvs(first)
    cal.calibrate({"serving_default": samples[3:]}, rm, samples_per_launch=k)
    outs.append(cal.get_model_qsvs())
  _same_qsvs(outs[1], outs[0])
  _same_qsvs(outs[1], _walk(m, model, rm, samples, 1).get_model_qsvs())


def test_resident_shared_and_missing_entries(m):
  """Samples whose tensors live in HBM, one tensor object shared by all samples, and samples that leave a tensor out
  (the walk then reads what an earlier sample left in the content map, ref calibrator.py:533)."""
  torch = m.torch
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  host = _c4_samples(10, 4, 8, 256, seed=5)
  shared = torch.from_numpy(host[0]["act3"]).cuda()
  samples = []
  for s, d in enumerate(host):
    e = {k: torch.from_numpy(v).cuda() for k, v in d.items() if k != "act3"}
    e["act3"] = shared
    if s in (3, 4, 8):
      del e["act1"]                     # act1 of sample 2 (resp. 7) is seen again
    if s == 5:
      e["act2"] = e["act2"].to(torch.bfloat16)
    samples.append(e)
  want = _walk(m, model, rm, samples, 1)
  for k in (3, 10):
    _same_qsvs(_walk(m, model, rm, samples, k).get_model_qsvs(), want.get_model_qsvs())


def test_a_sample_the_block_path_does_not_cover_goes_through_the_walk(m):
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(12, 4, 8, 256, seed=6)
  samples[4]["act2"] = (samples[4]["act2"] * 100).astype(np.int32)        # integers: plain min / max (ref :1382-1384)
  samples[9]["act0"] = samples[9]["act0"].astype(np.float64).astype(np.float32).astype(np.float64)
  want = _walk(m, model, rm, samples, 1)
  got = _walk(m, model, rm, samples, 5)
  assert set(got.get_model_qsvs()) == set(want.get_model_qsvs())
  for name, qsv in want.get_model_qsvs().items():
    for key in ("min", "max"):
      assert np.asarray(got.get_model_qsvs()[name][key]).tobytes() == np.asarray(qsv[key]).tobytes(), (name, key)


def test_a_sample_that_raises_mid_block_leaves_the_per_sample_state(m):
  """The reference counts a sample when it is taken up and has merged every sample before it (calibrator.py:325-330)."""
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(12, 4, 8, 256, seed=7)
  samples[6]["act1"] = np.zeros((0, 8, 256), np.float32)
  states = []
  for k in (1, 4, 64):
    cal = m.cal.Calibrator(model)
    with pytest.raises(ValueError, match="zero-size array"):
      cal.calibrate({"serving_default": samples}, rm, samples_per_launch=k)
    states.append(cal)
  for cal in states[1:]:
    assert cal._metadata == states[0]._metadata == {"num_samples_calibrated": 7}
    _same_qsvs(cal.get_model_qsvs(), states[0].get_model_qsvs())

  def failing():
    for s in samples[:5]:
      yield s
    raise RuntimeError("the dataset broke")
  for k in (1, 3):
    cal = m.cal.Calibrator(model)
    with pytest.raises(RuntimeError, match="the dataset broke"):
      cal.calibrate({"serving_default": failing()}, rm, samples_per_launch=k)
    assert cal._metadata == {"num_samples_calibrated": 5}
    _same_qsvs(cal.get_model_qsvs(), _walk(m, model, rm, samples[:5], 1).get_model_qsvs())


def test_datasets_made_while_they_are_read_keep_the_per_sample_walk(m):
  """A generator (or a tensor_provider) may hand out one buffer again and again: K of its samples are only read after
  the K-th has been pulled, so blocks are the caller's decision there."""
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(6, 4, 8, 256, seed=8)
  cal = m.cal.Calibrator(model)
  assert cal.samples_per_launch("serving_default", samples, rm) > 1
  assert cal.samples_per_launch("serving_default", iter(samples), rm) == 1
  assert cal.samples_per_launch("serving_default", iter(samples), rm, 4) == 4
  assert cal.samples_per_launch("serving_default", samples, rm, 1) == 1
  provided = m.cal.Calibrator(model, tensor_provider=lambda sig, s: s)
  assert provided.samples_per_launch("serving_default", samples, rm) == 1
  buf = {k: np.empty_like(v) for k, v in samples[0].items()}

  def reusing():
    for s in samples:
      for k, v in s.items():
        buf[k][...] = v
      yield buf
  cal.calibrate({"serving_default": reusing()}, rm)
  _same_qsvs(cal.get_model_qsvs(), _walk(m, model, rm, samples, 1).get_model_qsvs())


# ---- GPTQ: Hessians ----------------------------------------------------------------------------------------------------
SHAPES = (256, 128, 512)


def _gptq_setup(m, layers=2, sequences=20, tokens=512, contiguous=True):
  torch = m.torch
  model = m.c5.build_model(layers, *SHAPES)
  samples = m.c5.calibration_set(torch, layers, sequences, tokens, *SHAPES, out_tokens=8)
  if not contiguous:        # every sample's tokens in an allocation of their own
    samples = [{k: (v.clone() if v.shape[1] == tokens else v) for k, v in s.items()} for s in samples]
  return model, samples


@pytest.mark.parametrize("contiguous", [True, False])
def test_gptq_hessians_are_the_per_sample_walks_bit_for_bit(m, contiguous, monkeypatch):
  """512-token samples, products of at most 4096 tokens: a product closes every 8 samples, inside blocks (K = 20),
  at their edges (K = 8) and across them (K = 3, 5). Back-to-back samples are multiplied where they lie."""
  monkeypatch.setattr(m.gptq.HessianAccumulator, "SLAB_TOKENS", 4096)
  model, samples = _gptq_setup(m, contiguous=contiguous)
  rm = _rm(m, m.c5.recipe("gptq"))
  want = _walk(m, model, rm, samples, 1).get_model_qsvs()
  assert sum("hessian" in q for q in want.values()) == 2 * 4
  in_place = []
  real = m.gptq._back_to_back
  monkeypatch.setattr(m.gptq, "_back_to_back", lambda *a: in_place.append(real(*a)) or in_place[-1])
  for k in (3, 5, 8, 20):
    del in_place[:]
    got = _walk(m, model, rm, samples, k).get_model_qsvs()
    _same_qsvs(got, want)
    if k in (8, 20):
      assert in_place and all((x is not None) == contiguous for x in in_place), (k, in_place)


def test_gptq_all_hessians_and_sharded_world_of_one(m, monkeypatch):
  monkeypatch.setattr(m.gptq.HessianAccumulator, "SLAB_TOKENS", 4096)
  model, samples = _gptq_setup(m, layers=1, sequences=11, tokens=160)
  rm = _rm(m, m.c5.recipe("gptq"))
  want = _walk(m, model, rm, samples, 1, hessians="all").get_model_qsvs()
  _same_qsvs(_walk(m, model, rm, samples, 4, hessians="all").get_model_qsvs(), want)
  assert sum("hessian" in q for q in want.values()) == 4 + 7
  # the multi-GPU layer with one rank: blocks recorded, Hessians set aside, replayed
  want = _walk(m, model, rm, samples, 1).get_model_qsvs()
  for k in (1, 4, None):
    got = m.dist.calibrate_sharded(model, m.c5.recipe("gptq"), {"serving_default": samples}, samples_per_launch=k)
    _same_qsvs(got, want)
  mixed = m.c5.recipe("mixed")
  want = _walk(m, model, _rm(m, mixed), samples, 1).get_model_qsvs()
  _same_qsvs(m.dist.calibrate_sharded(model, mixed, {"serving_default": samples}), want)


def test_step_blocks_pickle_as_plain_arrays(m):
  import pickle
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(6, 4, 8, 256, seed=9)
  cal = m.cal.Calibrator(model)
  with cal.plan_once():
    blocks = [b for _, b in cal.record_blocks("serving_default", samples, rm, 4)]
  cal.wait_for_statistics()
  assert [len(b) for b in blocks] == [4, 2] and [b.first for b in blocks] == [0, 4]
  again = pickle.loads(pickle.dumps(blocks))
  for a, b in zip(again, blocks):
    assert a.slots == b.slots and a.ndims == b.ndims and a.first == b.first
    assert np.array_equal(a.stats, b.stats, equal_nan=True) and np.array_equal(a.num_samples, b.num_samples)
  a, b = m.cal.Calibrator(model), m.cal.Calibrator(model)
  a.replay(again)
  b.replay(e for blk in blocks for e in (blk.events(k) for k in range(len(blk))))
  _same_qsvs(a.get_model_qsvs(), b.get_model_qsvs())
  assert a._metadata == b._metadata == {"num_samples_calibrated": 6}


# ---- what the block path must not lose ---------------------------------------------------------------------
def test_a_custom_update_rule_keeps_every_gptq_hessian(m, monkeypatch):
  """A rule without `block_mode` (a wrapper around the stock one, a partial, a user's own) is handed events by the block
  replay that carry no 'hessian' (the tokens went into an accumulator): such signatures take the per-sample walk, whose
  events do (ref calibrator.py:567-582, utils/qsv_utils.py:90-122)."""
  import functools
  from mi355q.utils import qsv_utils
  monkeypatch.setattr(m.gptq.HessianAccumulator, "SLAB_TOKENS", 4096)
  model, samples = _gptq_setup(m, layers=1, sequences=9, tokens=256)
  rm = _rm(m, m.c5.recipe("gptq"))
  want = _walk(m, model, rm, samples, 1).get_model_qsvs()
  assert sum("hessian" in q for q in want.values()) == 4
  rules = (lambda a, b: qsv_utils.gptq_and_moving_average_update(a, b),
           functools.partial(qsv_utils.gptq_and_moving_average_update))
  gathered = []
  real = m.cal.Calibrator._gather_block
  monkeypatch.setattr(m.cal.Calibrator, "_gather_block", lambda self, *a, **kw: gathered.append(1) or real(self, *a, **kw))
  for rule in rules:
    for k in (None, 4):
      got = _walk(m, model, rm, samples, k, qsv_update_func=rule)
      assert not gathered, "a rule that cannot advance over blocks was given blocks"
      _same_qsvs(got.get_model_qsvs(), want)
      assert got._metadata == {"num_samples_calibrated": 9}
  _same_qsvs(_walk(m, model, rm, samples, 4).get_model_qsvs(), want)      # the stock rule: blocks, same result
  assert gathered


def test_a_float64_sample_mid_dataset_under_gptq_joins_the_accumulator(m, monkeypatch):
  """Sample 5 brings float64 tokens for one GPTQ input: the per-sample walk takes it (FP64 GEMM, a plain float64 Hessian
  comes out of the merge) and the blocks after it must carry on -- the finished Hessian becomes the float64 share of an
  accumulator (ref utils/qsv_utils.py:71-88: the sample-weighted mean)."""
  monkeypatch.setattr(m.gptq.HessianAccumulator, "SLAB_TOKENS", 4096)
  model, samples = _gptq_setup(m, layers=1, sequences=12, tokens=256)
  samples = [dict(s) for s in samples]
  samples[5]["l0/attn_in"] = samples[5]["l0/attn_in"].cpu().numpy().astype(np.float64)
  rm = _rm(m, m.c5.recipe("gptq"))
  want = _walk(m, model, rm, samples, 1).get_model_qsvs()
  for k in (4, None):
    got = _walk(m, model, rm, samples, k).get_model_qsvs()
    _same_qsvs(got, want, hessians=False)
    for name in want:
      assert ("hessian" in got[name]) == ("hessian" in want[name]), name
      if "hessian" in want[name]:
        a, b = np.asarray(got[name]["hessian"]), np.asarray(want[name]["hessian"])
        assert a.dtype == b.dtype == np.float64
        # same samples, same weights; the float32 products are grouped differently around the float64 sample
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), name
    assert int(got["l0/attn_in"]["num_samples"]) == 12


def test_blocks_and_single_steps_share_one_content_map(m):
  """A per-sample step after a block-mode calibrate() may omit a tensor (it reuses what the blocks saw last), and a
  block-mode call after per-sample steps starts from what those left (ref calibrator.py:529: ONE map)."""
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  samples = _c4_samples(9, 4, 8, 256, seed=10)
  partial_a = {k: v for k, v in samples[6].items() if k != "act1"}
  partial_b = {k: v for k, v in samples[8].items() if k != "act2"}

  def run(k):
    cal = m.cal.Calibrator(model)
    cal.calibrate({"serving_default": samples[:6]}, rm, samples_per_launch=k)
    cal.calibrate({"serving_default": [partial_a]}, rm, samples_per_launch=1)       # reads act1 of sample 5
    cal.calibrate({"serving_default": [samples[7], partial_b]}, rm, samples_per_launch=k)   # reads act2 of sample 7
    return cal
  want, got = run(1), run(4)
  _same_qsvs(got.get_model_qsvs(), want.get_model_qsvs())
  assert got._metadata == want._metadata == {"num_samples_calibrated": 9}
  assert not got._raw_carry


def test_float32_copies_made_for_a_block_leave_with_it(m):
  """bfloat16 and strided device samples are widened / gathered into float32 copies: the block that launched them gives
  them back (kept to the end they would be a second copy of the whole dataset in HBM)."""
  torch = m.torch
  x = torch.randn((1, 8, 256), device="cuda")
  rec = m.cal._describe(x.to(torch.bfloat16))
  assert rec is not None and rec[7] == x.numel() * 4
  rec = m.cal._describe(torch.randn((1, 256, 8), device="cuda").transpose(1, 2))
  assert rec is not None and rec[7] == x.numel() * 4
  assert m.cal._describe(x)[7] == 0
  model = m.c4.build_model(4, 256, 8)
  rm = _rm(m, m.recipe.static_wi8_ai8())
  host = _c4_samples(8, 4, 8, 256, seed=11)
  samples = [{k: torch.from_numpy(v).cuda().to(torch.bfloat16) for k, v in s.items()} for s in host]
  cal = m.cal.Calibrator(model)
  left = []
  with cal.plan_once():
    for _, block in cal.record_blocks("serving_default", samples, rm, 4):
      left.append(len(cal._described))
  assert left == [0, 0], left
  cal.wait_for_statistics()
  _same_qsvs(_walk(m, model, rm, samples, 4).get_model_qsvs(), _walk(m, model, rm, samples, 1).get_model_qsvs())
