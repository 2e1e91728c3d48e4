"""Model-level tensor sharding on the GPU: two ranks (gloo rendezvous, both on cuda:0) quantize a
model file with the product algorithms and rank 0's file must equal the single-process file."""
import os
import sys

import pytest

from test_distributed_gloo import ROOT, _run, _setup

pytestmark = pytest.mark.gpu

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


def test_two_ranks_quantize_model_files_like_one():
  (r0, got0), (r1, got1) = _run(_worker, timeout=600)
  assert len(got0) == len(_CASES) + 2
  for case, (sharded, single), (other, _) in zip([("big", 4), ("big", 8)] + _CASES, got0, got1):
    assert other is None and sharded is not None, case
    assert sharded == single, case


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
  out.put((0, ok))
  dist.barrier()
  dist.destroy_process_group()


def test_collectives_over_rccl_world_of_one():
  (_, ok), = _run(_worker_rccl_single, world=1, timeout=600)
  assert ok


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
