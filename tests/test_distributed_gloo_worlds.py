"""The multi-GPU layer at world sizes 2, 4 and 8 (gloo, CPU), with the shapes a large world makes awkward: fewer
samples than ranks (5 samples on 8 ranks: three ranks walk nothing), fewer ops than ranks, a signature whose only sample
lands on rank 0, Hessian owners that saw no sample of their tensor.

What runs is the product's control flow -- sample_shard, record_blocks / StepBlock, the object gather, the ordered
replay (ref calibrator.py:395-421, utils/qsv_utils.py:43-122), merge_hessians_across_ranks, plan_op_shards,
quantize_model_sharded -- with the per-tensor arithmetic supplied by the oracle where the product's needs a GPU (the
`-m gpu` tests run the same flows on the kernels with two ranks).
"""
import os
import sys

import numpy as np
import pytest

from test_distributed_gloo import (ROOT, _MODEL_CASES, _calibration_samples, _hessian_samples, _model_recipe,
                                   _register_oracle_algorithm, _register_oracle_calibration, _run, _setup, _tiny_fc)

WORLDS = [2, 4, 8]
TWO_SIGNATURES = os.path.join(ROOT, "tests", "golden", "models", "two_signatures.tflite")


def _signature_samples():
  """5 samples for "add", ONE for "multiply" (every rank but the first walks nothing of it)."""
  rng = np.random.default_rng(77)
  add = [{"add_x:0": rng.standard_normal(1).astype(np.float32) * (1 + s), "PartitionedCall:0": rng.standard_normal(1).astype(np.float32)}
         for s in range(5)]
  mul = [{"multiply_x:0": np.array([2.5], np.float32), "PartitionedCall_1:0": np.array([-7.0], np.float32)}]
  return {"add": add, "multiply": mul}


def _worker_block_calibration(rank, world, port, out):
  """calibrate_sharded on the product's K-samples-per-launch path (the launch stands in: host_launch)."""
  dist = _setup(rank, world, port)
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import c4_bench
  from test_calibration_blocks_host import host_launch
  from mi355q import calibrator, distributed as D, recipe
  from oracle import aeq_oracle as O
  model = c4_bench.build_model(3, 8, 4)
  rng = np.random.default_rng(5)
  samples = [{f"act{i}": rng.standard_normal((1, 4, 8)).astype(np.float32) * (1 + i + s) for i in range(3)} for s in range(5)]
  gathers = []
  real = dist.all_gather_object
  dist.all_gather_object = lambda parts, obj, group=None: (gathers.append(obj), real(parts, obj, group=group))[1]
  with host_launch() as launches:
    got = D.calibrate_sharded(model, recipe.static_wi8_ai8(), {"serving_default": samples})
    two = D.calibrate_sharded(TWO_SIGNATURES, recipe.static_wi8_ai8(), _signature_samples())
  blocks = [type(item).__name__ for g in gathers for _, _, item in g]
  ref = {}
  for s in samples:
    for name, x in s.items():
      ref[name] = O.moving_average_update(ref.get(name), O.activation_min_max(x, -3e38, 3e38))
  same = set(got) == set(ref) and all(
      np.asarray(got[n][k]).tobytes() == np.asarray(ref[n][k], np.float32).tobytes() for n in ref for k in ("min", "max"))
  ref2 = {}
  for sig in ("add", "multiply"):
    for s in _signature_samples()[sig]:
      for name, x in s.items():
        ref2[name] = O.moving_average_update(ref2.get(name), O.activation_min_max(x, -3e38, 3e38))
  same2 = all(np.asarray(two[n][k]).tobytes() == np.asarray(ref2[n][k], np.float32).tobytes() for n in ref2 for k in ("min", "max"))
  out.put((rank, same, same2, len(D.sample_shard(5, rank, world)), sum(launches), set(blocks) <= {"StepBlock"},
           {n: float(np.ravel(v["max"])[0]) for n, v in two.items()}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_block_calibration_with_fewer_samples_than_ranks(world):
  results = _run(_worker_block_calibration, world=world, timeout=240)
  assert len(results) == world
  assert sum(r[3] for r in results) == 5
  for rank, same, same2, n_mine, launched, only_blocks, two in results:
    assert same and same2 and only_blocks, rank
    # 3 tensors per sample of the first model, 2 per sample of whichever signature's samples this rank walks
    n_add = len(range(*_shard(5, rank, world)))
    assert launched == 3 * n_mine + 2 * n_add + (2 if rank == 0 else 0), (rank, launched)
    assert two == results[0][6]           # every rank ends with the same QSVs


def _shard(n, rank, world):
  base, extra = divmod(n, world)
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def _worker_owned_hessians(rank, world, port, out):
  """X2 with owners at any world size: 7 samples (rank 7 of 8 has none), the last rank owns "a", rank 0 owns "b"."""
  dist = _setup(rank, world, port)
  from mi355q import distributed as D
  from oracle import aeq_oracle as O
  xs = _hessian_samples()
  shard = D.sample_shard(len(xs), rank, world)
  local, totals = {}, {}
  for name, d in (("a", 16), ("b", 24)):
    q = None
    for s in shard:
      x = xs[s][..., :d]
      q = O.gptq_and_moving_average_update(q, {"min": np.float32(0), "max": np.float32(1), "hessian": O.gptq_hessian(x),
                                               "num_samples": x.shape[0]})
    if q is not None:
      local[name] = (q["hessian"], q["num_samples"])
    totals[name] = (d, sum(x.shape[0] for x in xs))
  merged = D.merge_hessians_across_ranks(local, totals, owners={"a": world - 1, "b": 0})
  everywhere = D.merge_hessians_across_ranks(local, totals)
  out.put((rank, {n: np.asarray(h) for n, h in merged.items()}, {n: np.asarray(h) for n, h in everywhere.items()}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_every_hessian_has_exactly_one_owner_at_any_world_size(world):
  from oracle import aeq_oracle as O
  results = _run(_worker_owned_hessians, world=world, timeout=240)
  held = {r[0]: set(r[1]) for r in results}
  assert held[world - 1] >= {"a"} and held[0] >= {"b"}
  assert sum("a" in h for h in held.values()) == 1 and sum("b" in h for h in held.values()) == 1
  xs = _hessian_samples()
  for name, d, owner in (("a", 16, world - 1), ("b", 24, 0)):
    q = None
    for x in xs:
      q = O.gptq_and_moving_average_update(q, {"min": np.float32(0), "max": np.float32(1),
                                               "hessian": O.gptq_hessian(x[..., :d]), "num_samples": x.shape[0]})
    got = dict((r[0], r[1]) for r in results)[owner][name]
    assert np.max(np.abs(got - q["hessian"])) / np.max(np.abs(q["hessian"])) <= 1e-14
    for r in results:             # ... and the all-reduce form gives every rank that same mean
      assert np.max(np.abs(r[2][name] - q["hessian"])) / np.max(np.abs(q["hessian"])) <= 1e-14


def _worker_calibrate_gptq_few_samples(rank, world, port, out):
  """calibrate_sharded with a GPTQ recipe and 5 samples: ranks without samples contribute zeros to every Hessian."""
  dist = _setup(rank, world, port)
  from mi355q import algorithm_manager as am, calibrator, distributed as D, recipe_manager
  from mi355q.algorithms.uniform_quantize import gptq
  from mi355q.utils import qsv_utils, tfl_flatbuffer_utils as fu
  from oracle import aeq_oracle as O
  base = _register_oracle_calibration()

  def calibrate(tfl_op, graph_info, tensor_content_map, inputs_to_ignore=None, outputs_to_ignore=None, valid_range=(-3e38, 3e38)):
    res = base(tfl_op, graph_info, tensor_content_map, inputs_to_ignore, outputs_to_ignore, valid_range)
    for name, qsv in res.items():
      qsv["hessian"] = O.gptq_hessian(tensor_content_map[name])
    return res
  for op in am.get_supported_ops(am.AlgorithmName.GPTQ.value):
    am.register_quantized_op(am.AlgorithmName.GPTQ.value, op, gptq.init_qsvs if hasattr(gptq, "init_qsvs") else None,
                             calibration_func=calibrate, materialize_func=lambda *a, **k: [],
                             update_qsv_func=qsv_utils.gptq_and_moving_average_update)
  calibrator.Calibrator._stage_sample = lambda *a, **k: None
  qsv_utils._gptq_merge_hessian = lambda a, b: (O.gptq_and_moving_average_update(
      {"min": 0.0, "max": 0.0, **a}, {"min": 0.0, "max": 0.0, **b})["hessian"], a["num_samples"] + b["num_samples"])
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_tiny_fc_gptq_w_{port}_{rank}.tflite")
  _tiny_fc(path)
  data = {"serving_default": _calibration_samples(5)}
  rcp = [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="GPTQ", op_config=dict(
      weight_tensor_config=dict(num_bits=4, symmetric=True, granularity="CHANNELWISE", dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]
  got = D.calibrate_sharded(path, rcp, data)
  owned = D.calibrate_sharded(path, rcp, data, hessian_owners={"x": world - 1})
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(rcp)
  single = calibrator.Calibrator(fu.read_model(path))
  single.calibrate(data, rm)
  want = single.get_model_qsvs()
  os.remove(path)
  rel = {n: float(np.max(np.abs(np.asarray(got[n]["hessian"]) - want[n]["hessian"])) / np.max(np.abs(want[n]["hessian"])))
         for n in want if "hessian" in want[n]}
  exact = all(np.array_equal(got[n][k], want[n][k]) for n in want for k in ("min", "max"))
  counts = all(int(got[n]["num_samples"]) == int(want[n]["num_samples"]) for n in want if "num_samples" in want[n])
  has_x = "hessian" in owned["x"]
  rel_owned = (float(np.max(np.abs(np.asarray(owned["x"]["hessian"]) - want["x"]["hessian"])) / np.max(np.abs(want["x"]["hessian"])))
               if has_x else None)
  out.put((rank, rel, exact and counts and set(got) == set(want), has_x, rel_owned))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gptq_calibration_with_idle_ranks(world):
  results = _run(_worker_calibrate_gptq_few_samples, world=world, timeout=240)
  for rank, rel, same, has_x, rel_owned in results:
    assert same and rel and all(e <= 1e-14 for e in rel.values()), (rank, rel)
    assert has_x == (rank == world - 1), rank          # reduced to its one owner (which walked no sample at world 8)
    assert rel_owned is None or rel_owned <= 1e-14


def _worker_model_files(rank, world, port, out):
  """quantize_model_sharded, bytes and FILE, on models with fewer quantized ops than ranks."""
  dist = _setup(rank, world, port)
  from mi355q import distributed as D, quantizer
  key = _register_oracle_algorithm()
  got = []
  for name, bits, gran in _MODEL_CASES:
    path = os.path.join(ROOT, "tests", "golden", "models", name)
    rcp = _model_recipe(key, bits, gran)
    _, _, plan, owner, _ = D.plan_model_shards(path, rcp, world)
    busy = len({o for it, o in zip(plan, owner) if str(getattr(it[4], "value", it[4])) != "no_quantize"})
    sharded = D.quantize_model_sharded(path, rcp)
    dst = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_worlds_{port}_{name}.q")
    D.quantize_model_sharded(path, rcp, serialize_to_path=dst)
    dist.barrier()
    if rank == 0:
      single = bytes(quantizer.Quantizer(path, rcp).quantize().quantized_model)
      with open(dst, "rb") as fh:
        got.append((bytes(sharded) == single, fh.read() == single, busy))
      os.remove(dst)
    else:
      got.append((sharded is None, True, busy))
  out.put((rank, got))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_model_files_equal_the_single_process_file(world):
  results = _run(_worker_model_files, world=world, timeout=300)
  for rank, got in results:
    assert len(got) == len(_MODEL_CASES)
    for same_bytes, same_file, busy in got:
      assert same_bytes and same_file, rank
      assert 1 <= busy <= world
