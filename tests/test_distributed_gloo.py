"""world_size=2 gloo tests of the multi-GPU layer's exchange + replay logic (CPU).

Per-sample statistics are produced with the oracle here (the product path needs
a GPU); what is under test is that sharding + all-gather + host replay reproduce
the single-process reference sequence bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _setup(rank, world, port):
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                    WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group("gloo", rank=rank, world_size=world)
  return dist


def _make_samples(n_samples=11, n_tensors=3):
  rng = np.random.default_rng(42)
  samples = []
  for s in range(n_samples):
    d = {}
    for t in range(n_tensors):
      x = rng.standard_normal((1, 8, 16), dtype=np.float32) * (1 + t)
      if (s + t) % 4 == 0:
        x.reshape(-1)[:3] = [np.inf, -np.inf, 3.39e38]
      d[f"act{t}"] = x
    samples.append(d)
  return samples


def _worker_calibration(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D
  from mi355q.utils import qsv_utils
  from oracle import aeq_oracle as O
  samples = _make_samples()
  names = sorted(samples[0])
  shard = D.sample_shard(len(samples), rank, world)
  local = np.empty((len(shard), len(names), 2), np.float32)
  for i, s in enumerate(shard):
    for t, n in enumerate(names):
      q = O.activation_min_max(samples[s][n], -3e38, 3e38)
      local[i, t] = (q["min"].item(), q["max"].item())
  stats = D.gather_sample_stats(local)
  shapes = [samples[0][n].shape for n in names]
  qsvs = D.replay_qsv_updates(stats, names, shapes)
  fast = D.allreduce_min_max(local)
  mm = D.replay_qsv_updates(stats, names, shapes, update_fn=qsv_utils.min_max_update)
  out.put((rank, stats, {n: (qsvs[n]["min"], qsvs[n]["max"]) for n in names}, fast,
           {n: (mm[n]["min"], mm[n]["max"]) for n in names}))
  dist.barrier()
  dist.destroy_process_group()


def _worker_hessian(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D
  rng = np.random.default_rng(5)
  xs = [rng.standard_normal((2 + i % 3, 6, 16)).astype(np.float32) for i in range(7)]
  shard = D.sample_shard(len(xs), rank, world)
  acc, n = np.zeros((16, 16), np.float64), 0
  for s in shard:
    x2 = xs[s].reshape(-1, 16)
    ns = xs[s].shape[0]
    acc += ns * ((2.0 / np.array(ns)) * x2.T.dot(x2))
    n += ns
  h, total = D.allreduce_hessian(acc, n)
  out.put((rank, h, total))
  dist.barrier()
  dist.destroy_process_group()


def _worker_shard_quantize(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D
  from oracle import aeq_oracle as O
  rng = np.random.default_rng(9)
  tensors = {f"w{i}": rng.standard_normal((8 + 4 * i, 32)).astype(np.float32) for i in range(5)}
  seen = []

  def fn(name, arr):
    seen.append(name)
    return O.min_max_quant_params(arr, 8, True, "CHANNELWISE")["quantized_data"]
  res = D.quantize_sharded(tensors, fn)
  out.put((rank, sorted(seen), None if res is None else {k: v for k, v in res.items()}))
  dist.barrier()
  dist.destroy_process_group()


def _register_oracle_algorithm():
  """Re-registers the min/max key, in this test process only, with the CPU oracle as
  get_tensor_quant_params, so that the model-level sharding logic can run without a GPU (the
  product algorithms need one)."""
  import functools
  from mi355q import algorithm_manager as am, default_policy, qtyping
  from mi355q.algorithms.uniform_quantize import common_quantize, naive_min_max_quantize
  from oracle import aeq_oracle as O
  key = am.AlgorithmName.MIN_MAX_UNIFORM_QUANT.value

  def get_tensor_quant_params(op_info, cfg, tensor_content=None, tensor_qsv=None):
    g = cfg.granularity
    r = O.min_max_quant_params(tensor_content, cfg.num_bits, cfg.symmetric,
                               getattr(g, "name", str(g)), op=op_info.op_name.name, qsv=tensor_qsv)
    return qtyping.UniformQuantParams(**r)
  for op, fn in am._MATERIALIZERS.items():
    am.register_quantized_op(key, op, naive_min_max_quantize.init_qsvs,
                             calibration_func=naive_min_max_quantize.min_max_calibrate,
                             materialize_func=functools.partial(fn, get_tensor_quant_params),
                             update_qsv_func=None)
  del default_policy, common_quantize
  return key


_MODEL_CASES = [("conv_fc_mnist.tflite", 8, "CHANNELWISE"), ("toy_model_with_kv_cache_multi_signature.tflite", 8, "CHANNELWISE"),
                ("weight_sharing_fcs.tflite", 8, "CHANNELWISE")]


def _model_recipe(key, bits, gran):
  return [dict(regex=".*", operation="*", algorithm_key=key, op_config=dict(
      weight_tensor_config=dict(num_bits=bits, symmetric=True, granularity=gran, dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]


def _worker_model(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D, quantizer
  key = _register_oracle_algorithm()
  got = []
  for name, bits, gran in _MODEL_CASES:
    path = os.path.join(ROOT, "tests", "golden", "models", name)
    recipe = _model_recipe(key, bits, gran)
    sharded = D.quantize_model_sharded(path, recipe)
    single = bytes(quantizer.Quantizer(path, recipe).quantize().quantized_model) if rank == 0 else None
    got.append((None if sharded is None else bytes(sharded), single))
  out.put((rank, got))
  dist.barrier()
  dist.destroy_process_group()


def _register_oracle_calibration():
  """min/max calibration through the oracle (the product function needs a GPU)."""
  from mi355q import algorithm_manager as am
  from mi355q.algorithms.uniform_quantize import common_quantize, naive_min_max_quantize
  from mi355q.utils import tfl_flatbuffer_utils as fu
  from oracle import aeq_oracle as O

  def calibrate(tfl_op, graph_info, tensor_content_map, inputs_to_ignore=None, outputs_to_ignore=None,
                valid_range=(-3e38, 3e38)):
    out = {}
    for tid in common_quantize.get_tensor_indices_requiring_calibration(tfl_op, graph_info, inputs_to_ignore,
                                                                        outputs_to_ignore):
      tensor = graph_info.subgraph_tensors[tid]
      if fu.get_tensor_data(tensor, graph_info.buffers) is not None:
        continue
      name = fu.get_tensor_name(tensor)
      x = tensor_content_map[name]
      qsv = O.activation_min_max(x, *valid_range)
      qsv["num_samples"] = np.array(x.shape[0] if x.ndim > 0 else 1)
      out[name] = qsv
    return out
  key = am.AlgorithmName.MIN_MAX_UNIFORM_QUANT.value
  for op in am.get_supported_ops(key):
    am.register_quantized_op(key, op, naive_min_max_quantize.init_qsvs, calibration_func=calibrate,
                             materialize_func=lambda *a, **k: [])
  return calibrate


def _calibration_samples(n=9):
  rng = np.random.default_rng(123)
  return [{"x": (rng.standard_normal((1, 8)) * (1 + s)).astype(np.float32),
           "y": (rng.standard_normal((1, 4)) * (2 + s)).astype(np.float32)} for s in range(n)]


def _tiny_fc(path):
  from mi355q import qtyping as q
  from mi355q.utils import tflite_flatbuffer as fb
  w = np.arange(32, dtype=np.float32).reshape(4, 8) / 7
  model = q.ModelT(version=3, buffers=[q.BufferT(), q.BufferT(data=w.reshape(-1).view(np.uint8))],
                   operatorCodes=[q.OperatorCodeT(builtinCode=9, deprecatedBuiltinCode=9)])
  sg = q.SubGraphT(name=b"main", inputs=[0], outputs=[2],
                   tensors=[q.TensorT(name=b"x", shape=[1, 8], buffer=0), q.TensorT(name=b"w", shape=[4, 8], buffer=1),
                            q.TensorT(name=b"y", shape=[1, 4], buffer=0)],
                   operators=[q.OperatorT(inputs=[0, 1, -1], outputs=[2], builtinOptionsType=8,
                                          builtinOptions=q.FullyConnectedOptionsT())])
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0,
                                         inputs=[q.TensorMapT(name=b"x", tensorIndex=0)],
                                         outputs=[q.TensorMapT(name=b"y", tensorIndex=2)])]
  open(path, "wb").write(fb.write_model(model))


def _worker_calibrate_model(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import calibrator, distributed as D, recipe, recipe_manager
  from mi355q.utils import tfl_flatbuffer_utils as fu
  _register_oracle_calibration()
  calibrator.Calibrator._stage_sample = lambda *a, **k: None      # HBM staging needs a GPU; test process only
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_tiny_fc_{port}_{rank}.tflite")
  _tiny_fc(path)
  data = {"serving_default": _calibration_samples()}
  rcp = recipe.static_wi8_ai8()
  got = D.calibrate_sharded(path, rcp, data)
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(rcp)
  single = calibrator.Calibrator(fu.read_model(path))
  single.calibrate(data, rm)
  want = single.get_model_qsvs()
  os.remove(path)
  same = set(got) == set(want) and all(
      np.array_equal(got[n][k], want[n][k]) for n in want for k in ("min", "max") if k in want[n])
  out.put((rank, same, sorted(got), float(np.ravel(got["x"]["max"])[0])))
  dist.barrier()
  dist.destroy_process_group()


def _worker_x2_exchange(rank, world, port, out):
  """X2 at the exchange level: every rank holds the running mean of its own samples."""
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
  # "c" was seen by rank 1 only (a rank without samples of a tensor contributes zeros)
  totals["c"] = (8, 5)
  if rank == 1:
    local["c"] = (np.arange(64, dtype=np.float64).reshape(8, 8), 5)
  before = {n: np.array(h, copy=True) for n, (h, _) in local.items()}
  merged = D.merge_hessians_across_ranks(local, totals)
  untouched = all(np.array_equal(before[n], local[n][0]) for n in local)     # the QSVs' own arrays are not written
  out.put((rank, {n: np.asarray(h) for n, h in merged.items()}, untouched))
  dist.barrier()
  dist.destroy_process_group()


def _worker_x2_owners(rank, world, port, out):
  """X2 with owners: every Hessian is reduced to the one rank that will read it."""
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
    local[name] = (q["hessian"], q["num_samples"])
    totals[name] = (d, sum(x.shape[0] for x in xs))
  merged = D.merge_hessians_across_ranks(local, totals, owners={"a": 1, "b": 0})
  out.put((rank, {n: np.asarray(h) for n, h in merged.items()}))
  dist.barrier()
  dist.destroy_process_group()


def _hessian_samples():
  rng = np.random.default_rng(15)
  return [rng.standard_normal((1 + i % 3, 6, 24)).astype(np.float32) * (1 + i) for i in range(7)]


def _worker_calibrate_gptq(rank, world, port, out):
  """calibrate_sharded with a GPTQ recipe on CPU: the Hessian statistics come from the oracle (the
  product functions need a GPU; test process only), the sharded control flow is the product's."""
  dist = _setup(rank, world, port)
  import pickle
  from mi355q import algorithm_manager as am, calibrator, distributed as D, recipe_manager
  from mi355q.algorithms.uniform_quantize import gptq
  from mi355q.utils import qsv_utils, tfl_flatbuffer_utils as fu
  from oracle import aeq_oracle as O
  base = _register_oracle_calibration()

  def calibrate(tfl_op, graph_info, tensor_content_map, inputs_to_ignore=None, outputs_to_ignore=None,
                valid_range=(-3e38, 3e38)):
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
  sizes = []
  real_gather = dist.all_gather_object

  def counting_gather(parts, obj, group=None):
    sizes.append(len(pickle.dumps(obj)))
    return real_gather(parts, obj, group=group)
  dist.all_gather_object = counting_gather
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_tiny_fc_gptq_{port}_{rank}.tflite")
  _tiny_fc(path)
  data = {"serving_default": _calibration_samples()}
  rcp = [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="GPTQ", op_config=dict(
      weight_tensor_config=dict(num_bits=4, symmetric=True, granularity="CHANNELWISE", dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]
  got = D.calibrate_sharded(path, rcp, data)
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
  clean = all("hessian_dim" not in q for q in got.values())
  out.put((rank, rel, exact and counts and clean and set(got) == set(want), max(sizes),
           max(want[n]["hessian"].nbytes for n in rel)))
  dist.barrier()
  dist.destroy_process_group()


def _run(worker, world=2, timeout=300):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  results = []
  try:
    import queue as _queue
    import time
    deadline = time.time() + timeout
    while len(results) < world:
      try:
        results.append(q.get(timeout=1.0))
      except _queue.Empty:
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        assert not dead, f"worker exited with {dead}"
        assert time.time() < deadline, "workers timed out"
    for p in procs:
      p.join(120)
      assert p.exitcode == 0
  finally:
    for p in procs:
      if p.is_alive():
        p.kill()
  return sorted(results, key=lambda r: r[0])


def test_sample_shard_and_plan():
  from mi355q import distributed as D
  assert [list(D.sample_shard(11, r, 2)) for r in range(2)] == [list(range(6)), list(range(6, 11))]
  assert sum(len(D.sample_shard(512, r, 8)) for r in range(8)) == 512
  owner = D.plan_tensor_shards([10, 9, 8, 7, 1, 1], 2)
  loads = [sum(b for b, o in zip([10, 9, 8, 7, 1, 1], owner) if o == r) for r in range(2)]
  assert abs(loads[0] - loads[1]) <= 2 and owner == D.plan_tensor_shards([10, 9, 8, 7, 1, 1], 2)
  # C3: 32 equal layers over 8 ranks -> 4 each
  owner = D.plan_tensor_shards([180355072] * 32, 8)
  assert sorted(owner.count(r) for r in range(8)) == [4] * 8


def test_gathered_ema_replay_matches_sequential_reference():
  from oracle import aeq_oracle as O
  results = _run(_worker_calibration)
  samples = _make_samples()
  names = sorted(samples[0])
  # single-process reference sequence: moving_average_update per sample, in order
  ref, ref_mm = {}, {}
  for s in samples:
    for n in names:
      q = O.activation_min_max(s[n], -3e38, 3e38)
      ref[n] = O.moving_average_update(ref.get(n), {"min": q["min"], "max": q["max"]})
      ref_mm[n] = O.min_max_update(ref_mm.get(n), {"min": q["min"], "max": q["max"]})
  for rank, stats, qsvs, fast, mm in results:
    assert stats.shape == (len(samples), len(names), 2)
    for t, n in enumerate(names):
      assert np.array_equal(qsvs[n][0], ref[n]["min"]) and np.array_equal(qsvs[n][1], ref[n]["max"])
      assert qsvs[n][0].shape == (1, 1, 1)
      assert np.array_equal(mm[n][0], ref_mm[n]["min"]) and np.array_equal(mm[n][1], ref_mm[n]["max"])
      assert fast[t, 0] == ref_mm[n]["min"].item() and fast[t, 1] == ref_mm[n]["max"].item()
  assert np.array_equal(results[0][1], results[1][1])


def test_hessian_allreduce_matches_sequential_merge():
  from oracle import aeq_oracle as O
  results = _run(_worker_hessian)
  rng = np.random.default_rng(5)
  xs = [rng.standard_normal((2 + i % 3, 6, 16)).astype(np.float32) for i in range(7)]
  q = None
  for x in xs:
    q = O.gptq_and_moving_average_update(
        q, {"min": np.float32(0), "max": np.float32(1), "hessian": O.gptq_hessian(x),
            "num_samples": x.shape[0]})
  for rank, h, total in results:
    assert total == q["num_samples"]
    np.testing.assert_allclose(h, q["hessian"], rtol=1e-12, atol=1e-12)


def test_x2_hessian_exchange_equals_sequential_merge_chain():
  """merge_hessians_across_ranks (one weighted all-reduce per distinct Hessian) against the chain
  of _gptq_merge_hessian over all samples in order (ref utils/qsv_utils.py:71-102)."""
  from oracle import aeq_oracle as O
  results = _run(_worker_x2_exchange)
  xs = _hessian_samples()
  for name, d in (("a", 16), ("b", 24)):
    q = None
    for x in xs:
      q = O.gptq_and_moving_average_update(q, {"min": np.float32(0), "max": np.float32(1),
                                               "hessian": O.gptq_hessian(x[..., :d]), "num_samples": x.shape[0]})
    for rank, merged, untouched in results:
      assert untouched
      err = np.max(np.abs(merged[name] - q["hessian"])) / np.max(np.abs(q["hessian"]))
      assert err <= 1e-14, (name, rank, err)
  for rank, merged, _ in results:
    assert np.array_equal(merged["c"], np.arange(64, dtype=np.float64).reshape(8, 8))   # weight 5/5 from one rank
  assert all(np.array_equal(results[0][1][n], results[1][1][n]) for n in ("a", "b", "c"))


def test_x2_reduce_to_owner_keeps_each_hessian_on_the_rank_that_reads_it():
  from oracle import aeq_oracle as O
  results = dict(_run(_worker_x2_owners))
  assert set(results[0]) == {"b"} and set(results[1]) == {"a"}
  xs = _hessian_samples()
  for name, d, owner in (("a", 16, 1), ("b", 24, 0)):
    q = None
    for x in xs:
      q = O.gptq_and_moving_average_update(q, {"min": np.float32(0), "max": np.float32(1),
                                               "hessian": O.gptq_hessian(x[..., :d]), "num_samples": x.shape[0]})
    got = results[owner][name]
    assert np.max(np.abs(got - q["hessian"])) / np.max(np.abs(q["hessian"])) <= 1e-14


def _c5_plan(world, variant="gptq", layers=18):
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  import c5_model as C
  from mi355q import distributed as D
  model = C.build_model(layers, weights="virtual")       # shapes and byte counts only
  _, _, plan, owner, costs = D.plan_model_shards(model, C.recipe(variant), world)
  return D, plan, owner, costs


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_c5_plan_keeps_a_hessians_ops_together_and_balances_cost(world):
  """The sharding plan of the 18-layer Gemma-2B-shaped model under GPTQ (ref
  gptq.py:243-300 for what is shared): ops that read one activation sit on one rank, so every
  inverse is computed once in the whole job; the load is balanced by modelled seconds, not bytes."""
  D, plan, owner, costs = _c5_plan(world)
  by_key = {}
  for (_, key, _), r in zip(costs, owner):
    if key is not None:
      by_key.setdefault(key, set()).add(r)
  assert len(by_key) == 18 * 4 and all(len(r) == 1 for r in by_key.values())
  owners = D.hessian_owners(plan, owner, costs)
  assert len(owners) == 18 * 4
  loads = D.plan_loads(costs, owner, world)
  mean = sum(loads) / world
  # 18 indivisible d = 16384 units (one inverse + one apply each, ~64 ms) dominate: the best any plan
  # can do is ceil(18 / world) of them on the busiest rank
  units = {}
  for w, key, shared in costs:
    if key is not None:
      units[key] = units.get(key, shared) + w
  heavy = max(units.values())
  bound = -(-18 // world) * heavy
  assert max(loads) <= max(1.15 * mean, bound * 1.02), (loads, mean, bound)
  # by bytes the two kinds of 128 MiB weights would be interchangeable; by cost they are not
  down = [c for c, it in zip(costs, plan) if it[3] is not None and c[2] > 0.01]
  assert len(down) == 18


def test_c5_plan_is_deterministic_and_mixed_recipe_has_no_heavy_units():
  a = _c5_plan(8)[2]
  b = _c5_plan(8)[2]
  assert a == b
  D, plan, owner, costs = _c5_plan(8, "mixed")
  loads = D.plan_loads(costs, owner, 8)
  assert max(loads) <= 1.15 * (sum(loads) / 8)


def test_c5_plan_prices_the_hessian_exchange():
  """X2 in the plan: one ring reduce of a packed float32 triangle per distinct Hessian, to its owner;
  the time is the same for every rank, and what stays exposed once the reduces run beside the inverses is about one reduce."""
  tri = lambda d: d * (d + 1) // 2 * 4
  D, plan, owner, costs = _c5_plan(1)
  x = D.x2_reduce_plan(plan, owner, costs, 1)
  assert x["hessians"] == 72 and x["bytes"] == 0 and x["seconds"] == 0.0           # one rank: nothing travels
  for world in (2, 8):
    D, plan, owner, costs = _c5_plan(world)
    x = D.x2_reduce_plan(plan, owner, costs, world)
    assert x["hessians"] == 72 and x["bytes"] == 18 * tri(16384) + 54 * tri(2048)
    assert sum(x["bytes_to_owner"]) == x["bytes"] and len(x["bytes_to_owner"]) == world
    owners = D.hessian_owners(plan, owner, costs)
    assert all(b > 0 for b in x["bytes_to_owner"]) and len(set(owners.values())) == world
    link = D.COST_MODEL["xgmi_link_bytes_per_s"]
    want = 72 * D.COST_MODEL["collective_s"] + x["bytes"] * (world - 1) / world / link
    assert abs(x["seconds"] - want) < 1e-4
    one_big = D.COST_MODEL["collective_s"] + tri(16384) * (world - 1) / world / link
    assert one_big - 1e-5 <= x["seconds_exposed"] < 0.1 * x["seconds"] + one_big          # the first reduce of an owner
  D, plan, owner, costs = _c5_plan(8, "mixed")
  x = D.x2_reduce_plan(plan, owner, costs, 8)
  assert x["hessians"] == 54 and x["bytes"] == 54 * tri(2048)                             # no d = 16384 Hessian in the mixed recipe


def test_sample_sharded_gptq_calibration_reduces_hessians_instead_of_gathering_them():
  """calibrate_sharded with a GPTQ recipe: min / max / num_samples equal the single-process result
  exactly, the Hessian within FP64 rounding, and no d x d array enters the object gather."""
  results = _run(_worker_calibrate_gptq)
  for rank, rel, same, gathered_bytes, hessian_bytes in results:
    assert same and rel and all(e <= 1e-14 for e in rel.values()), (rank, rel)
    assert gathered_bytes < hessian_bytes * 9       # nine samples' Hessians would be at least this
    assert gathered_bytes < 8192, gathered_bytes


def test_tensor_sharded_quantize_gathers_everything_on_rank0():
  from oracle import aeq_oracle as O
  results = _run(_worker_shard_quantize)
  rng = np.random.default_rng(9)
  tensors = {f"w{i}": rng.standard_normal((8 + 4 * i, 32)).astype(np.float32) for i in range(5)}
  (r0, seen0, res0), (r1, seen1, res1) = results
  assert res1 is None and sorted(seen0 + seen1) == sorted(tensors)
  assert seen0 and seen1  # both ranks did work
  for n, w in tensors.items():
    assert np.array_equal(res0[n], O.min_max_quant_params(w, 8, True, "CHANNELWISE")["quantized_data"])


def test_model_level_sharded_quantize_equals_single_process():
  (r0, got0), (r1, got1) = _run(_worker_model)
  for (sharded, single), (other, _) in zip(got0, got1):
    assert other is None and sharded is not None and sharded == single


def test_model_level_sample_sharded_calibration_equals_single_process():
  (r0, same0, names0, x0), (r1, same1, names1, x1) = _run(_worker_calibrate_model)
  assert same0 and same1 and names0 == names1 and x0 == x1
  assert "x" in names0 and "y" in names0
