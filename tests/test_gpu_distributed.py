"""Model-level tensor sharding on the GPU: two ranks (gloo rendezvous, both on cuda:0) quantize a
model file with the product algorithms and rank 0's file must equal the single-process file."""
import os

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


def _worker(rank, world, port, out):
  dist = _setup(rank, world, port)
  from mi355q import distributed as D, quantizer
  got = []
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
  assert len(got0) == len(_CASES)
  for case, (sharded, single), (other, _) in zip(_CASES, got0, got1):
    assert other is None and sharded is not None, case
    assert sharded == single, case
