"""Model-level tensor sharding on the GPU: two ranks (gloo rendezvous, both on cuda:0) quantize a
model file with the product algorithms and rank 0's file must equal the single-process file."""
import os
import sys

import pytest

import parity_rates
from test_distributed_gloo import ROOT, _run
from test_distributed_gloo import _setup as _setup_gloo

pytestmark = pytest.mark.gpu

# Every worker below runs twice: as two gloo ranks sharing cuda:0, and as two ranks over RCCL ("nccl" backend), where the data
# collectives are libmi355q's own RCCL entry points (mi355q_allgather_minmax, mi355q_reduce_product_f32, ...) and every rank's
# payloads leave from HBM. With two GPUs visible that is one rank per GPU over xGMI; on a one-GPU box (the builder's lease) the
# two ranks name different hosts (distributed.one_gpu_ranks_env: NCCL_HOSTID) and meet over RCCL's socket transport, both on
# cuda:0 -- RCCL with a real peer either way.
_BACKEND_ENV = "MI355Q_TEST_DIST_BACKEND"


def _gpus() -> int:
  try:
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0
  except Exception:  # pylint: disable=broad-exception-caught
    return 0


two_ranks_over_rccl = pytest.mark.skipif(_gpus() < 1, reason="RCCL with N > 1: one rank per GPU, or ranks as separate hosts on one GPU")


def _setup(rank, world, port):
  """The workers' rendezvous: gloo (ranks share cuda:0) unless the test asked for RCCL (one rank per device)."""
  if os.environ.get(_BACKEND_ENV) != "nccl":
    return _setup_gloo(rank, world, port)
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  import torch
  import torch.distributed as dist
  device = rank
  if torch.cuda.device_count() < world:      # fewer GPUs than ranks: the ranks are separate "hosts" sharing cuda:0
    from mi355q import distributed as D
    os.environ.update(D.one_gpu_ranks_env(rank))
    device = 0
  torch.cuda.set_device(device)
  dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
  return dist


def _rccl_ranks():
  """(ranks, my rank) as libmi355q's communicator reports them (mi355q_comm_info), or None on a host transport."""
  import ctypes
  from mi355q import _ffi, distributed as D
  comm = D.rccl_comm()
  if comm is None:
    return None
  nr, rk = ctypes.c_int32(-1), ctypes.c_int32(-1)
  _ffi.check(_ffi.lib().mi355q_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
  return (nr.value, rk.value)


def _over_rccl(monkeypatch):
  monkeypatch.setenv(_BACKEND_ENV, "nccl")      # (spawned workers inherit the environment)

_CASES = [
    ("conv_fc_mnist.tflite", "min_max_uniform_quantize", 8, "CHANNELWISE"),
    ("conv_fc_mnist.tflite", "OCTAV", 4, "CHANNELWISE"),
    ("toy_model_with_kv_cache_multi_signature.tflite", "min_max_uniform_quantize", 4, "BLOCKWISE_32"),
    ("toy_model_with_kv_cache_multi_signature.tflite", "MSE", 4, "CHANNELWISE"),
    ("weight_sharing_fcs.tflite", "min_max_uniform_quantize", 8, "CHANNELWISE"),
    ("constant_tensor_and_buffer_only_sharing_weight_fcs.tflite", "min_max_uniform_quantize", 4, "CHANNELWISE"),
    ("bmm_constant_input.tflite", "min_max_uniform_quantize", 8, "CHANNELWISE"),
]


def _recipe(key, bits, gran):
  return [dict(regex=".*", operation="*", algorithm_key=key, op_config=dict(
      weight_tensor_config=dict(num_bits=bits, symmetric=True, granularity=gran, dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]


def _big_fc_model(path):
  """Two FULLY_CONNECTED layers of 8 MiB each: their quantized weights stay in HBM on the rank
  that made them and cross to rank 0 as host data."""
  import numpy as np
  from mi355q import qtyping as q
  from mi355q.utils import tflite_flatbuffer as fb
  rng = np.random.default_rng(5)
  model = q.ModelT(version=3)
  model.buffers = [q.BufferT()]
  sg = q.SubGraphT(name=b"main", inputs=[0], outputs=[4], tensors=[q.TensorT(name=b"x", shape=[1, 1024], buffer=0)], operators=[])
  for i in range(2):
    w = rng.standard_normal((2048 if i == 0 else 1024, 1024 if i == 0 else 2048)).astype(np.float32)
    model.buffers.append(q.BufferT(data=w.reshape(-1).view(np.uint8)))
    sg.tensors.append(q.TensorT(name=f"w{i}".encode(), shape=list(w.shape), buffer=len(model.buffers) - 1))
    sg.tensors.append(q.TensorT(name=f"y{i}".encode(), shape=[1, w.shape[0]], buffer=0))
    sg.operators.append(q.OperatorT(inputs=[2 * i, 2 * i + 1, -1], outputs=[2 * i + 2], builtinOptionsType=8,
                                    builtinOptions=q.FullyConnectedOptionsT()))
  model.operatorCodes = [q.OperatorCodeT(builtinCode=9, deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  open(path, "wb").write(fb.write_model(model))


def _shared_weight_model(path):
  """x -> FC(w) -> y0 -> FC(w) -> y1 -> FC(w2) -> y2: the first two ops read ONE weight tensor (tied weights) and w2 is
  another tensor over the SAME buffer; 1 MiB of int8 each, so the payloads stay in their rank's HBM when a file is written."""
  import numpy as np
  from mi355q import qtyping as q
  from mi355q.utils import tflite_flatbuffer as fb
  w = np.random.default_rng(11).standard_normal((1024, 1024)).astype(np.float32)
  model = q.ModelT(version=3, buffers=[q.BufferT(), q.BufferT(data=w.reshape(-1).view(np.uint8))],
                   operatorCodes=[q.OperatorCodeT(builtinCode=9, deprecatedBuiltinCode=9)])
  t = [q.TensorT(name=b"x", shape=[1, 1024], buffer=0), q.TensorT(name=b"w", shape=[1024, 1024], buffer=1),
       q.TensorT(name=b"y0", shape=[1, 1024], buffer=0), q.TensorT(name=b"y1", shape=[1, 1024], buffer=0),
       q.TensorT(name=b"w2", shape=[1024, 1024], buffer=1), q.TensorT(name=b"y2", shape=[1, 1024], buffer=0)]
  fc = lambda i, w_, o: q.OperatorT(inputs=[i, w_, -1], outputs=[o], builtinOptionsType=8, builtinOptions=q.FullyConnectedOptionsT())
  model.subgraphs = [q.SubGraphT(name=b"main", inputs=[0], outputs=[5], tensors=t, operators=[fc(0, 1, 2), fc(2, 1, 3), fc(3, 4, 5)])]
  open(path, "wb").write(fb.write_model(model))


def _worker_shared_weights(rank, world, port, out):
  """A weight two ops read, written to a FILE by a sharded run: the ops land on one rank (distributed.shared_constant_links),
  their results name one payload, and the sharing checks compare records instead of bytes -- through the ranks' own writes
  and through the by-host route (MI355Q_REMOTE_WRITES_BY_HOST: what a rank does when it cannot open the output file)."""
  dist = _setup(rank, world, port)
  from mi355q import distributed as D, quantizer
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_shared_w_{port}.tflite")
  if rank == 0:
    _shared_weight_model(path)
  dist.barrier()
  got = []
  for key, bits, gran in (("min_max_uniform_quantize", 8, "CHANNELWISE"), ("min_max_uniform_quantize", 4, "BLOCKWISE_128")):
    rcp = _recipe(key, bits, gran)
    _, _, plan, owner, _ = D.plan_model_shards(path, rcp, world)
    one_rank = len({o for it, o in zip(plan, owner) if str(getattr(it[4], "value", it[4])) != "no_quantize"}) == 1
    single = bytes(quantizer.Quantizer(path, rcp).quantize().quantized_model) if rank == 0 else None
    for by_host in (False, True):
      if by_host:
        os.environ["MI355Q_REMOTE_WRITES_BY_HOST"] = "1"
      try:
        dst = path + f".{bits}.{int(by_host)}.q"
        D.quantize_model_sharded(path, rcp, serialize_to_path=dst)
      finally:
        os.environ.pop("MI355Q_REMOTE_WRITES_BY_HOST", None)
      dist.barrier()
      if rank == 0:
        with open(dst, "rb") as fh:
          got.append((fh.read() == single, one_rank))
        os.remove(dst)
  dist.barrier()
  if rank == 0:
    os.remove(path)
  out.put((rank, (got, _rccl_ranks())))
  dist.barrier()
  dist.destroy_process_group()


def _check_shared_weights(results, rccl: bool):
  results = dict(results)
  got, _ = results[0]
  assert len(got) == 4
  for same_file, one_rank in got:
    assert same_file and one_rank
  for rank, (_, comm) in results.items():
    assert comm == ((2, rank) if rccl else None), (rank, comm)


def test_two_ranks_write_a_model_whose_ops_share_a_weight():
  _check_shared_weights(_run(_worker_shared_weights, timeout=600), rccl=False)


@two_ranks_over_rccl
def test_two_ranks_write_a_model_whose_ops_share_a_weight_over_rccl(monkeypatch):
  """One rank per GPU: the payload of the rank that is not the writer crosses from ITS device into the file."""
  _over_rccl(monkeypatch)
  _check_shared_weights(_run(_worker_shared_weights, timeout=600), rccl=True)


def _worker_layout_failure(rank, world, port, out):
  """The rank that lays the file out fails (an unwritable directory): every rank raises, none waits for the others."""
  dist = _setup(rank, world, port)
  from mi355q import distributed as D
  big = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_big_fc_fail_{port}.tflite")
  if rank == 0:
    _big_fc_model(big)
  dist.barrier()
  try:
    D.quantize_model_sharded(big, _recipe("min_max_uniform_quantize", 8, "CHANNELWISE"),
                             serialize_to_path="/nonexistent-directory/out.tflite")
    raised = None
  except Exception as e:  # pylint: disable=broad-exception-caught
    raised = type(e).__name__ + ": " + str(e)[:80]
  dist.barrier()
  if rank == 0:
    os.remove(big)
  out.put((rank, raised))
  dist.barrier()
  dist.destroy_process_group()


def test_a_failure_of_the_writing_rank_reaches_every_rank():
  results = dict(_run(_worker_layout_failure, timeout=300))
  assert results[0] is not None and results[1] is not None, results
  assert "lays the output file out failed" in results[1], results


def _worker(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D, quantizer
  got = []
  big = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_big_fc_{port}.tflite")
  if rank == 0:
    _big_fc_model(big)
  dist.barrier()
  for key, bits, gran in (("min_max_uniform_quantize", 4, "BLOCKWISE_128"), ("min_max_uniform_quantize", 8, "CHANNELWISE")):
    sharded = D.quantize_model_sharded(big, _recipe(key, bits, gran))
    single = bytes(quantizer.Quantizer(big, _recipe(key, bits, gran)).quantize().quantized_model) if rank == 0 else None
    got.append((None if sharded is None else bytes(sharded), single))
    # ... and written to a FILE: the quantized payloads then stay in their ranks' HBM and every rank writes its own into
    # the file rank 0 laid out (runtime.RemoteBuffer); blockwise scales (64 KiB of float32 here) still travel as values
    out_path = big + f".{bits}.q.tflite"
    ret = D.quantize_model_sharded(big, _recipe(key, bits, gran), serialize_to_path=out_path)
    dist.barrier()
    if rank == 0:
      with open(out_path, "rb") as fh:
        on_disk = fh.read()
      got.append((on_disk, single))
      assert ret is not None
      os.remove(out_path)
    else:
      got.append((None, None))
      assert ret is None
  dist.barrier()
  if rank == 0:
    os.remove(big)
  for name, key, bits, gran in _CASES:
    path = os.path.join(ROOT, "tests", "golden", "models", name)
    sharded = D.quantize_model_sharded(path, _recipe(key, bits, gran))
    single = bytes(quantizer.Quantizer(path, _recipe(key, bits, gran)).quantize().quantized_model) if rank == 0 else None
    got.append((None if sharded is None else bytes(sharded), single))
  out.put((rank, got))
  dist.barrier()
  dist.destroy_process_group()


def _check_model_files(results):
  (r0, got0), (r1, got1) = results
  assert len(got0) == len(_CASES) + 4
  for case, (sharded, single), (other, _) in zip([("big", 4), ("big file", 4), ("big", 8), ("big file", 8)] + _CASES, got0, got1):
    assert other is None and sharded is not None, case
    assert sharded == single, case


def test_two_ranks_quantize_model_files_like_one():
  _check_model_files(_run(_worker, timeout=600))


@two_ranks_over_rccl
def test_two_ranks_quantize_model_files_like_one_over_rccl(monkeypatch):
  _over_rccl(monkeypatch)
  _check_model_files(_run(_worker, timeout=600))


def _worker_rccl_single(rank, world, port, out):
  """World of one over RCCL ("nccl" backend): the collectives take the HBM code path."""
  import numpy as np
  import torch
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
  import torch.distributed as dist
  torch.cuda.set_device(0)
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
  from mi355q import distributed as D, runtime as rt
  assert D._comm_device().type == "cuda"
  rng = np.random.default_rng(3)
  h = rng.standard_normal((64, 64))
  got, total = D.allreduce_hessian(rt.HbmArray(torch.from_numpy(h * 5).cuda()), 5)
  ok = bool(np.allclose(got.cpu().numpy(), h) and total == 5)
  stats = rng.standard_normal((7, 3, 2)).astype(np.float32)
  ok &= bool(np.array_equal(D.gather_sample_stats(stats), stats))
  mm = D.allreduce_min_max(stats)
  ok &= bool(np.array_equal(mm[:, 0], stats[..., 0].min(0)) and np.array_equal(mm[:, 1], stats[..., 1].max(0)))
  # ---- the C-ABI RCCL entry points themselves, on the communicator distributed.py creates
  import ctypes
  from mi355q import _ffi
  L = _ffi.lib()
  comm = D.rccl_comm()
  ok &= comm is not None and D.rccl_comm() is comm
  nr, rk = ctypes.c_int32(-1), ctypes.c_int32(-1)
  _ffi.check(L.mi355q_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
  ok &= (nr.value, rk.value) == (1, 0)
  st = rt.stream_ptr()
  a = torch.from_numpy(stats.reshape(-1).copy()).cuda()
  b = torch.empty_like(a)
  _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(a), a.numel(), rt.ptr(b), st))
  ok &= bool(torch.equal(a, b))
  mn, mx = a.clone(), a.clone()
  _ffi.check(L.mi355q_allreduce_minmax_f32(comm, rt.ptr(mn), rt.ptr(mx), a.numel(), st))
  ok &= bool(torch.equal(mn, a) and torch.equal(mx, a))
  f32 = a.clone()
  _ffi.check(L.mi355q_allreduce_sum_f32(comm, rt.ptr(f32), f32.numel(), st))
  f64 = torch.from_numpy(h).cuda()
  _ffi.check(L.mi355q_allreduce_sum_f64(comm, rt.ptr(f64), f64.numel(), st))
  ok &= bool(torch.equal(f32, a) and np.array_equal(f64.cpu().numpy(), h))
  hh = torch.from_numpy(h).cuda()
  _ffi.check(L.mi355q_allreduce_hessian_f64(comm, rt.ptr(hh), 64, 0.25, st))
  ok &= bool(np.array_equal(hh.cpu().numpy(), h * 0.25))
  ok &= L.mi355q_allreduce_hessian_f64(comm, rt.ptr(hh), 64, 1.5, st) == -1          # weight outside [0, 1]
  # the packed-triangle form: all-reduce (root -1) and reduce to a root; 67 is not a multiple of the tile
  sym = rng.standard_normal((67, 67)); sym = sym + sym.T
  need = L.mi355q_hessian_exchange_workspace_bytes(67)
  ok &= need == 67 * 68 // 2 * 8
  ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
  for root in (-1, 0):
    hs = torch.from_numpy(sym).cuda()
    _ffi.check(L.mi355q_reduce_hessian_f64(comm, rt.ptr(hs), 67, 0.5, root, rt.ptr(ws), need, st))
    ok &= bool(np.array_equal(hs.cpu().numpy(), sym * 0.5))
  ok &= L.mi355q_reduce_hessian_f64(comm, rt.ptr(hs), 67, 0.5, 1, rt.ptr(ws), need, st) == -1      # no rank 1
  ok &= L.mi355q_reduce_hessian_f64(comm, rt.ptr(hs), 67, 0.5, 0, rt.ptr(ws), need - 8, st) == -1  # workspace too small
  # the float32 product form (round 4): only the lower triangle travels and only it is written back
  prod = torch.from_numpy(rng.standard_normal((67, 67)).astype(np.float32)).cuda()
  need32 = L.mi355q_product_exchange_workspace_bytes(67)
  ok &= need32 == 67 * 68 // 2 * 4
  ws32 = torch.empty((need32,), dtype=torch.uint8, device="cuda")
  for root in (-1, 0):
    pp = prod.clone()
    _ffi.check(L.mi355q_reduce_product_f32(comm, rt.ptr(pp), 67, root, rt.ptr(ws32), need32, st))
    ok &= bool(torch.equal(pp, prod))                      # a world of one: the sum is the rank's own product
  ok &= L.mi355q_reduce_product_f32(comm, None, 67, 0, rt.ptr(ws32), need32, st) == -1            # the receiving rank has no buffer
  ok &= L.mi355q_reduce_product_f32(comm, rt.ptr(pp), 67, 1, rt.ptr(ws32), need32, st) == -1      # no rank 1
  ok &= L.mi355q_reduce_product_f32(comm, rt.ptr(pp), 67, 0, rt.ptr(ws32), need32 - 4, st) == -1  # workspace too small
  ok &= L.mi355q_allreduce_sum_f32(None, rt.ptr(f32), 4, st) == -1                    # null communicator
  # a second communicator from an explicit unique-id hand-over (what a non-torch rendezvous would do)
  other = D.new_rccl_comm(0, 1, lambda uid: uid)
  _ffi.check(L.mi355q_allreduce_sum_f32(other, rt.ptr(f32), f32.numel(), st))
  torch.cuda.synchronize()
  ok &= bool(torch.equal(f32, a))
  _ffi.check(L.mi355q_comm_destroy(other))
  merged = D.merge_hessians_across_ranks({"t": (rt.HbmArray(torch.from_numpy(h).cuda()), 3)}, {"t": (64, 3)})
  ok &= bool(np.array_equal(np.asarray(merged["t"]), h))
  # ---- the product reduces on the communication stream (what N ranks run for X2): issued back to back in the order
  # given, one event per Hessian, the reader of a product waits for ITS event only
  from mi355q.algorithms.uniform_quantize import gptq
  del D.ISSUED[:]
  big = torch.from_numpy(rng.standard_normal((1024, 1024)).astype(np.float32)).cuda()
  small = torch.from_numpy(rng.standard_normal((67, 67)).astype(np.float32)).cuda()
  want_big, want_small = big.clone(), small.clone()
  busy = torch.ones((1 << 26,), device="cuda")
  for _ in range(20):
    busy = busy * 1.0000001                                  # the compute stream has work queued in front of the reduces
  events = D.reduce_products_beside_compute(D.rccl_comm(), [("a/big", big, 1024, 0), ("b/small", small, 67, -1)])
  ok &= D.ISSUED == [("a/big", 0), ("b/small", -1)] and list(events) == ["a/big", "b/small"]
  acc = gptq.HessianAccumulator(1024)
  acc._prod, acc._n_prod, acc.ready = big, 4.0, events["a/big"]        # pylint: disable=protected-access
  prod, alpha = acc.product_form()                                       # waits for the event on the reading stream
  ok &= acc.ready is None and alpha == 0.5
  ok &= bool(torch.equal(torch.tril(prod), torch.tril(want_big)))        # a world of one: the sum is the rank's own triangle
  events["b/small"].synchronize()
  ok &= bool(torch.equal(torch.tril(small), torch.tril(want_small)))
  ok &= D.comm_stream() is D.comm_stream() and D.comm_stream() != torch.cuda.current_stream()
  D.destroy_rccl_comms()
  out.put((0, bool(ok)))
  dist.barrier()
  dist.destroy_process_group()


def test_collectives_over_rccl_world_of_one():
  """torch.distributed is only the rendezvous: the data collectives are libmi355q's own RCCL entry
  points (include/mi355q.h), exercised here on a world of one (the box has one GPU)."""
  (_, ok), = _run(_worker_rccl_single, world=1, timeout=600)
  assert ok


def _worker_rccl_pair(rank, world, port, out):
  """Two ranks over RCCL: every collective entry point of include/mi355q.h with a real peer and rank-dependent data
  (sums, extrema, gathers and root-directed reduces that a world of one cannot tell from a copy)."""
  import ctypes
  import numpy as np
  os.environ[_BACKEND_ENV] = "nccl"
  dist = _setup(rank, world, port)
  import torch
  from mi355q import _ffi, distributed as D, runtime as rt
  L = _ffi.lib()
  comm = D.rccl_comm()
  ok = comm is not None and D._comm_device().type == "cuda"     # pylint: disable=protected-access
  nr, rk = ctypes.c_int32(-1), ctypes.c_int32(-1)
  _ffi.check(L.mi355q_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
  ok &= (nr.value, rk.value) == (world, rank)
  st = rt.stream_ptr()
  rngs = [np.random.default_rng(100 + r) for r in range(world)]      # every rank can rebuild every rank's data
  stats = [g.standard_normal((7, 3, 2)).astype(np.float32) for g in rngs]
  hs = [g.standard_normal((67, 67)) for g in rngs]
  hs = [h + h.T for h in hs]
  prods = [g.standard_normal((67, 67)).astype(np.float32) for g in rngs]
  weights = [0.25, 0.75] if world == 2 else [1.0 / world] * world
  # X1: all-gather of the per-sample pairs, and the min / max all-reduce pair (one RCCL group)
  a = torch.from_numpy(stats[rank].reshape(-1).copy()).cuda()
  g_out = torch.empty((world * a.numel(),), device="cuda")
  _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(a), a.numel(), rt.ptr(g_out), st))
  ok &= bool(np.array_equal(g_out.cpu().numpy(), np.concatenate([s.reshape(-1) for s in stats])))
  mn, mx = a.clone(), a.clone()
  _ffi.check(L.mi355q_allreduce_minmax_f32(comm, rt.ptr(mn), rt.ptr(mx), a.numel(), st))
  flat = np.stack([s.reshape(-1) for s in stats])
  ok &= bool(np.array_equal(mn.cpu().numpy(), flat.min(0)) and np.array_equal(mx.cpu().numpy(), flat.max(0)))
  # the same through the host layer (what calibrate_sharded calls)
  ok &= bool(np.array_equal(D.gather_sample_stats(stats[rank]), np.concatenate(stats)))
  both = D.allreduce_min_max(stats[rank])
  allst = np.concatenate(stats)
  ok &= bool(np.array_equal(both[:, 0], allst[..., 0].min(0)) and np.array_equal(both[:, 1], allst[..., 1].max(0)))
  # sums (two ranks: one addition per element, the same in any order)
  f32 = a.clone()
  _ffi.check(L.mi355q_allreduce_sum_f32(comm, rt.ptr(f32), f32.numel(), st))
  f64 = torch.from_numpy(hs[rank]).cuda()
  _ffi.check(L.mi355q_allreduce_sum_f64(comm, rt.ptr(f64), f64.numel(), st))
  if world == 2:
    ok &= bool(np.array_equal(f32.cpu().numpy(), flat[0] + flat[1]) and np.array_equal(f64.cpu().numpy(), hs[0] + hs[1]))
  # X2, float64: the weighted all-reduce, then the packed triangle to every rank / to each root in turn
  hh = torch.from_numpy(hs[rank]).cuda()
  _ffi.check(L.mi355q_allreduce_hessian_f64(comm, rt.ptr(hh), 67, weights[rank], st))
  want = sum(w * h for w, h in zip(weights, hs)) if world != 2 else weights[0] * hs[0] + weights[1] * hs[1]
  ok &= bool(np.array_equal(hh.cpu().numpy(), want))
  need = L.mi355q_hessian_exchange_workspace_bytes(67)
  ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
  for root in [-1] + list(range(world)):
    h1 = torch.from_numpy(hs[rank]).cuda()
    _ffi.check(L.mi355q_reduce_hessian_f64(comm, rt.ptr(h1), 67, weights[rank], root, rt.ptr(ws), need, st))
    got = h1.cpu().numpy()
    ok &= bool(np.array_equal(got, want if root in (-1, rank) else hs[rank]))      # the other ranks keep what they had
  ok &= L.mi355q_reduce_hessian_f64(comm, rt.ptr(h1), 67, 0.5, world, rt.ptr(ws), need, st) == -1     # no such rank (no peer is waited for)
  # X2, the float32 product form: lower triangles summed, one rank may have seen no sample (NULL product)
  need32 = L.mi355q_product_exchange_workspace_bytes(67)
  ws32 = torch.empty((need32,), dtype=torch.uint8, device="cuda")
  tril = np.tril(np.ones((67, 67), bool))
  for root in [-1] + list(range(world)):
    pp = torch.from_numpy(prods[rank]).cuda()
    _ffi.check(L.mi355q_reduce_product_f32(comm, rt.ptr(pp), 67, root, rt.ptr(ws32), need32, st))
    got = pp.cpu().numpy()
    if root in (-1, rank):
      wsum = prods[0] + prods[1] if world == 2 else sum(prods)
      ok &= bool(np.array_equal(got[tril], wsum[tril]))
    else:
      ok &= bool(np.array_equal(got, prods[rank]))
  if world == 2:       # rank 1 saw no sample: it contributes zeros, rank 0 ends with its own triangle
    pp = torch.from_numpy(prods[rank]).cuda()
    _ffi.check(L.mi355q_reduce_product_f32(comm, rt.ptr(pp) if rank == 0 else None, 67, 0, rt.ptr(ws32), need32, st))
    if rank == 0:
      ok &= bool(np.array_equal(pp.cpu().numpy()[tril], prods[0][tril]))
  # the host layer over the same communicator: the weighted Hessian mean and the merge of per-rank Hessians
  got_h, total = D.allreduce_hessian(rt.HbmArray(torch.from_numpy(hs[rank] * (rank + 2)).cuda()), rank + 2)
  mean = sum(hs[r] * (r + 2) for r in range(world)) / sum(r + 2 for r in range(world))
  ok &= total == sum(r + 2 for r in range(world))
  ok &= bool(np.allclose(got_h.cpu().numpy(), mean, rtol=1e-13, atol=1e-13))
  # the product reduces on the communication stream, owner-directed, behind queued compute: each reader waits for ITS event
  from mi355q.algorithms.uniform_quantize import gptq
  del D.ISSUED[:]
  big = [torch.from_numpy(np.random.default_rng(300 + r).standard_normal((1024, 1024)).astype(np.float32)) for r in range(world)]
  mine_big, mine_small = big[rank].cuda(), torch.from_numpy(prods[rank]).cuda()
  busy = torch.ones((1 << 26,), device="cuda")
  for _ in range(20):
    busy = busy * 1.0000001
  owner_big = world - 1
  events = D.reduce_products_beside_compute(comm, [("a/big", mine_big, 1024, owner_big), ("b/small", mine_small, 67, -1)])
  ok &= D.ISSUED == [("a/big", owner_big), ("b/small", -1)]
  acc = gptq.HessianAccumulator(1024)
  acc._prod, acc._n_prod, acc.ready = mine_big, 4.0, events["a/big"]      # pylint: disable=protected-access
  prod, _ = acc.product_form()
  t1024 = np.tril(np.ones((1024, 1024), bool))
  if rank == owner_big:
    wsum = (big[0] + big[1]).numpy() if world == 2 else sum(b.numpy() for b in big)
    ok &= bool(np.array_equal(prod.cpu().numpy()[t1024], wsum[t1024]))
  else:
    ok &= bool(np.array_equal(prod.cpu().numpy(), big[rank].numpy()))
  events["b/small"].synchronize()
  wsmall = prods[0] + prods[1] if world == 2 else sum(prods)
  ok &= bool(np.array_equal(mine_small.cpu().numpy()[tril], wsmall[tril]))
  torch.cuda.synchronize()
  dist.barrier()
  D.destroy_rccl_comms()
  out.put((rank, bool(ok)))
  dist.barrier()
  dist.destroy_process_group()


@two_ranks_over_rccl
def test_collectives_over_rccl_two_ranks():
  """Every RCCL entry point of the C ABI and the host layer's "nccl" branches with a real peer (one rank per GPU where two are
  visible, else two "hosts" on cuda:0: distributed.one_gpu_ranks_env)."""
  results = dict(_run(_worker_rccl_pair, world=2, timeout=600))
  assert results == {0: True, 1: True}


def _gptq_recipe(bits=4):
  return [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="GPTQ", op_config=dict(
      weight_tensor_config=dict(num_bits=bits, symmetric=True, granularity="CHANNELWISE", dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]


def _fc_model(path, rows, d, seed=3):
  import numpy as np
  from mi355q import qtyping as q
  from mi355q.utils import tflite_flatbuffer as fb
  w = np.random.default_rng(seed).standard_normal((rows, d)).astype(np.float32) * np.float32(0.05)
  model = q.ModelT(version=3, buffers=[q.BufferT(), q.BufferT(data=w.reshape(-1).view(np.uint8))],
                   operatorCodes=[q.OperatorCodeT(builtinCode=9, deprecatedBuiltinCode=9)])
  sg = q.SubGraphT(name=b"main", inputs=[0], outputs=[2],
                   tensors=[q.TensorT(name=b"x", shape=[1, d], buffer=0), q.TensorT(name=b"w", shape=[rows, d], buffer=1),
                            q.TensorT(name=b"y", shape=[1, rows], buffer=0)],
                   operators=[q.OperatorT(inputs=[0, 1, -1], outputs=[2], builtinOptionsType=8,
                                          builtinOptions=q.FullyConnectedOptionsT())])
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0,
                                         inputs=[q.TensorMapT(name=b"x", tensorIndex=0)],
                                         outputs=[q.TensorMapT(name=b"y", tensorIndex=2)])]
  open(path, "wb").write(fb.write_model(model))


def _gptq_samples(d, rows, n=9):
  import numpy as np
  rng = np.random.default_rng(77)
  return [{"x": (rng.standard_normal((2 + s % 3, 16, d)) * (1 + 0.1 * s)).astype(np.float32),
           "y": rng.standard_normal((2 + s % 3, 16, rows)).astype(np.float32)} for s in range(n)]


def _worker_calibrate_gptq(rank, world, port, out):
  """X2 on the GPU (two gloo ranks sharing cuda:0): Hessians stay in HBM on the rank that made
  them and are combined by one weighted all-reduce; nothing d x d is pickled."""
  dist = _setup(rank, world, port)
  import pickle
  import numpy as np
  from mi355q import calibrator, distributed as D, recipe_manager, runtime as rt
  from mi355q.utils import tfl_flatbuffer_utils as fu
  d, rows = 256, 64
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_fc_gptq_{port}_{rank}.tflite")
  _fc_model(path, rows, d)
  data = {"serving_default": _gptq_samples(d, rows)}
  sizes = []
  real_gather = dist.all_gather_object

  def counting_gather(parts, obj, group=None):
    sizes.append(len(pickle.dumps(obj)))
    return real_gather(parts, obj, group=group)
  dist.all_gather_object = counting_gather
  got = D.calibrate_sharded(path, _gptq_recipe(), data)
  # the exchange above summed the ranks' float32 PRODUCTS (packed triangle); the float64 exchange of round 3
  # (MI355Q_X2_F64=1: every rank's product scaled into a float64 Hessian first) must give the same mean
  os.environ["MI355Q_X2_F64"] = "1"
  got64 = D.calibrate_sharded(path, _gptq_recipe(), data)
  del os.environ["MI355Q_X2_F64"]
  h32, h64 = np.asarray(got["x"]["hessian"]), np.asarray(got64["x"]["hessian"])
  rel_forms = float(np.max(np.abs(h32 - h64)) / np.max(np.abs(h64)))
  kept_as_product = hasattr(got["x"]["hessian"], "product_form") and got["x"]["hessian"].product_form() is not None
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(_gptq_recipe())
  single = calibrator.Calibrator(fu.read_model(path))
  single.calibrate(data, rm)
  want = single.get_model_qsvs()
  os.remove(path)
  ok = set(got) == set(want) and isinstance(got["x"]["hessian"], rt.HbmArray)
  hw, hg = np.asarray(want["x"]["hessian"]), np.asarray(got["x"]["hessian"])
  rel = float(np.max(np.abs(hg - hw)) / np.max(np.abs(hw)))
  ok &= all(np.array_equal(np.asarray(got[n][k]), np.asarray(want[n][k])) for n in want for k in ("min", "max"))
  ok &= int(got["x"]["num_samples"]) == int(want["x"]["num_samples"]) == sum(s["x"].shape[0] for s in data["serving_default"])
  ok &= "hessian_dim" not in got["x"]
  out.put((rank, bool(ok), rel, max(sizes), hw.nbytes, rel_forms, kept_as_product, _rccl_ranks()))
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_reduce_gptq_hessians_in_hbm():
  _check_reduced_hessians(_run(_worker_calibrate_gptq, timeout=600), rccl=False)


@two_ranks_over_rccl
def test_two_ranks_reduce_gptq_hessians_in_hbm_over_rccl(monkeypatch):
  """X2 over xGMI: mi355q_reduce_product_f32 (packed float32 triangles, ncclAllReduce / ncclReduce) between two devices
  -- what replaces the reference's sample-ordered merge chain (ref utils/qsv_utils.py:71-102, calibrator.py:395-421)."""
  _over_rccl(monkeypatch)
  _check_reduced_hessians(_run(_worker_calibrate_gptq, timeout=600), rccl=True)


def _check_reduced_hessians(results, rccl: bool):
  for rank, ok, rel, gathered_bytes, hessian_bytes, rel_forms, kept_as_product, comm in results:
    assert comm == ((2, rank) if rccl else None), (rank, comm)
    assert ok, rank
    # X2 as packed float32 product triangles against X2 as float64 Hessians: the same mean within float32 summation
    assert rel_forms <= 1e-7, rel_forms
    assert kept_as_product        # the receiving rank keeps product form: the damped inverse reads it as it is
    # vs the one-process calibration: every process multiplies the tokens of its own samples in one
    # float32-accumulated product (gptq.HessianAccumulator), so two ranks add two such products in FP64
    # where one process forms a single one -- float32 accumulation noise, far inside T2's 2e-6 (the
    # FP64 all-reduce itself equals the sequential merge chain to 1e-14: tests/test_distributed_gloo.py)
    assert rel <= 1e-6, rel
    assert gathered_bytes < hessian_bytes // 8    # 9 samples' statistics, not one d x d array


def _worker_c5_fused(rank, world, port, out):
  """calibrate_and_quantize_sharded on a small Gemma-shaped model (two gloo ranks on cuda:0)."""
  dist = _setup(rank, world, port)
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import torch
  import c5_model as C
  from mi355q import distributed as D, model_modifier, ops, quantizer
  shapes = (256, 128, 512)
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_c5_small_{port}.tflite")
  if rank == 0:
    model_modifier.serialize_model(C.build_model(2, *shapes), path)
  dist.barrier()
  samples = C.calibration_set(torch, 2, 8, 160, *shapes)          # 8 samples of [1, 160, d] per input
  data = {"serving_default": samples}
  rcp = C.recipe("gptq")
  calls = {"hinv": 0}
  real, real_batched, real_product = ops.gptq_hinv, ops.gptq_hinv_batched, ops.gptq_hinv_from_product

  def counting(*a, **k):
    calls["hinv"] += 1
    return real(*a, **k)

  def counting_batched(hs, *a, **k):
    calls["hinv"] += len(hs)
    return real_batched(hs, *a, **k)

  def counting_product(*a, **k):          # (a Hessian that arrived as the ranks' summed float32 product is inverted from it)
    calls["hinv"] += 1
    return real_product(*a, **k)
  ops.gptq_hinv, ops.gptq_hinv_batched, ops.gptq_hinv_from_product = counting, counting_batched, counting_product
  del D.ISSUED[:]
  sharded = D.calibrate_and_quantize_sharded(path, rcp, data)
  mine = calls["hinv"]
  issued = list(D.ISSUED)
  ops.gptq_hinv, ops.gptq_hinv_batched, ops.gptq_hinv_from_product = real, real_batched, real_product
  single = None
  if rank == 0:
    qz = quantizer.Quantizer(path, rcp)
    single = bytes(qz.quantize(calibration_result=qz.calibrate(data)).quantized_model)
  dist.barrier()
  if rank == 0:
    os.remove(path)
  out.put((rank, mine, None if sharded is None else bytes(sharded), single, _rccl_ranks(), issued))
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_calibrate_and_quantize_in_one_call_one_inverse_per_hessian():
  """BASELINE config 5's call on two ranks: every distinct Hessian (2 layers x 4 inputs) is inverted
  exactly once in the whole job, on the rank that owns the ops reading it, and the file rank 0
  writes is the single-process file but for T2 (the Hessian of two ranks' token products; observed
  identical bytes)."""
  _check_c5_fused(sorted(_run(_worker_c5_fused, timeout=600)), "two ranks", rccl=False)


@two_ranks_over_rccl
def test_two_ranks_calibrate_and_quantize_in_one_call_over_rccl(monkeypatch):
  """The same call with one rank per GPU: statistics all-gathered, every Hessian reduced over xGMI to the rank that owns
  its readers (ncclReduce on the packed float32 triangle), results gathered to rank 0."""
  _over_rccl(monkeypatch)
  _check_c5_fused(sorted(_run(_worker_c5_fused, timeout=600)), "two GPUs over RCCL", rccl=True)


def _check_c5_fused(results, label: str, rccl: bool):
  import numpy as np
  (r0, n0, sharded, single, comm0, issued0), (r1, n1, other, _, comm1, issued1) = results
  assert (comm0, comm1) == (((2, 0), (2, 1)) if rccl else (None, None))
  # X2: every distinct Hessian is ONE reduce, issued in the same order on both ranks (RCCL matches collectives by order
  # of issue: a rank that issued them in another order would deadlock or add the wrong triangles), each to the rank
  # that owns its readers
  assert issued0 == issued1 and len(issued0) == 8 and [n for n, _ in issued0] == sorted(n for n, _ in issued0)
  assert {root for _, root in issued0} == {0, 1}
  assert other is None and sharded is not None
  assert n0 + n1 == 8 and n0 > 0 and n1 > 0, (n0, n1)
  assert len(sharded) == len(single)
  diff = np.frombuffer(sharded, np.uint8) != np.frombuffer(single, np.uint8)
  parity_rates.note(f"C5 small model, {label} vs one: differing bytes of the written file", "byte_mismatch_fraction",
                    float(diff.mean()), 1e-4)


def test_quantize_sharded_tool_with_a_gptq_recipe(tmp_path):
  """tools/quantize_sharded.py, two ranks (gloo transport, both on cuda:0), GPTQ recipe with
  calibration samples: the file equals the one-rank file but for T2 (observed: identical)."""
  import json
  import subprocess
  import numpy as np
  from test_distributed_gloo import _free_port
  d, rows = 256, 64
  model = str(tmp_path / "fc.tflite")
  for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
    if p not in sys.path:
      sys.path.insert(0, p)
  _fc_model(model, rows, d)
  rcp = str(tmp_path / "gptq.json")
  json.dump(_gptq_recipe(), open(rcp, "w"))
  samples = _gptq_samples(d, rows)
  np.savez(str(tmp_path / "calib.npz"), **{f"{i}/{k}": v for i, s in enumerate(samples) for k, v in s.items()})
  outs = []
  for n in (1, 2):
    dst = str(tmp_path / f"q{n}.tflite")
    env = dict(os.environ, MI355Q_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "quantize_sharded.py"), model, rcp, dst,
           str(tmp_path / "calib.npz")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    outs.append(open(dst, "rb").read())
  assert len(outs[0]) == len(outs[1])
  import parity_rates
  a, b = np.frombuffer(outs[0], np.uint8), np.frombuffer(outs[1], np.uint8)
  assert (a != b).mean() <= 1e-4                  # everything but (possibly) a few int4 nibbles
  from mi355q.utils import tfl_flatbuffer_utils as fu
  qa, qb = (np.asarray(fu.read_model(o).buffers[1].data) for o in outs)
  def nib(v):
    u = np.stack([v & 15, v >> 4], 1).reshape(-1).astype(np.int16)
    return np.where(u > 7, u - 16, u)
  parity_rates.check("quantize_sharded.py GPTQ int4: 2 ranks vs 1 rank", nib(qa), nib(qb), parity_rates.T2)


def _worker_calibrate(rank, world, port, out):
  dist = _setup(rank, world, port)
  import numpy as np
  from mi355q import calibrator, distributed as D, recipe, recipe_manager
  from mi355q.utils import tfl_flatbuffer_utils as fu
  from test_distributed_gloo import _tiny_fc
  path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mi355q_tiny_fc_{port}_{rank}.tflite")
  _tiny_fc(path)
  rng = np.random.default_rng(31)
  data = {"serving_default": [{"x": (rng.standard_normal((1 + s % 3, 8)) * (1 + s)).astype(np.float32),
                               "y": (rng.standard_normal((1 + s % 3, 4)) * 3).astype(np.float32)} for s in range(11)]}
  oscar_static = [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="OSCAR", op_config=dict(
      activation_tensor_config=dict(num_bits=8, symmetric=False, granularity="TENSORWISE", dtype="INT"),
      weight_tensor_config=dict(num_bits=8, symmetric=True, granularity="CHANNELWISE", dtype="INT"),
      compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]
  ok = True
  for rcp, keys in ((recipe.static_wi8_ai8(), ("min", "max")), (oscar_static, ("min", "max", "mu2", "num_samples"))):
    got = D.calibrate_sharded(path, rcp, data)
    rm = recipe_manager.RecipeManager()
    rm.load_quantization_recipe(rcp)
    single = calibrator.Calibrator(fu.read_model(path))
    single.calibrate(data, rm)
    want = single.get_model_qsvs()
    ok &= set(got) == set(want) and all(np.array_equal(np.asarray(got[n][k]), np.asarray(want[n][k]))
                                       for n in want for k in keys if k in want[n])
    ok &= all(k in got["x"] for k in keys)
  os.remove(path)
  out.put((rank, bool(ok)))
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_calibrate_like_one():
  """Samples sharded over two ranks, events gathered and replayed: the QSVs (moving-average
  min/max; OSCAR's sample-weighted mu2) equal the single-process ones bit for bit."""
  (r0, ok0), (r1, ok1) = _run(_worker_calibrate, timeout=600)
  assert ok0 and ok1


@two_ranks_over_rccl
def test_two_ranks_calibrate_like_one_over_rccl(monkeypatch):
  """BASELINE config 4's exchange between two devices: the per-sample (min, max) travel through
  mi355q_allgather_minmax and are replayed in dataset order (ref utils/qsv_utils.py:43-68)."""
  _over_rccl(monkeypatch)
  (r0, ok0), (r1, ok1) = _run(_worker_calibrate, timeout=600)
  assert ok0 and ok1
