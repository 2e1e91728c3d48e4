"""dequantized_weight_recovery on the GPU against outputs of the real reference
(tests/golden/ref_dwr_cases.*, ref_dwr_model_cases.json) and the oracle on seeded problems."""
import json
import os

import numpy as np
import pytest

from golden_util import describe_model
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = os.path.join(HERE, "golden", "models")
with open(os.path.join(HERE, "golden", "ref_dwr_cases.json")) as _f:
  CASES = {c["name"]: c for c in json.load(_f)["cases"]}
with open(os.path.join(HERE, "golden", "ref_dwr_model_cases.json")) as _f:
  MODEL_CASES = json.load(_f)["cases"]


def _call(w, bits, gran, op="FULLY_CONNECTED", skip_checks=False):
  from mi355q import qtyping as q
  from mi355q.algorithms.uniform_quantize import dequantized_weight_recovery as dwr
  cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName[op], subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg, skip_checks=skip_checks))
  return dwr.get_tensor_quant_params(info, cfg, w)


@pytest.mark.parametrize("name", sorted(CASES))
def test_quant_params_match_reference(name):
  z = np.load(os.path.join(HERE, "golden", "ref_dwr_cases.npz"))
  c = CASES[name]
  w = z[f"{name}/w"]
  if "error" in c:
    with pytest.raises(RuntimeError) as err:
      _call(w, c["num_bits"], c["granularity"], c["op"])
    assert str(err.value)[:170] == c["message"][:170]          # up to the digits of the max diff
    assert "unique values" in str(err.value)
    res = _call(w, c["num_bits"], c["granularity"], c["op"], skip_checks=True)   # skip_checks: no validation
    assert res.quantized_data.shape == w.shape
    return
  res = _call(w, c["num_bits"], c["granularity"], c["op"])
  assert res.scale.dtype == z[f"{name}/scale"].dtype and np.array_equal(res.scale, z[f"{name}/scale"])
  assert res.scale.shape == z[f"{name}/scale"].shape
  assert np.array_equal(res.quantized_data, z[f"{name}/q"]) and res.quantized_data.dtype == np.int8
  assert np.array_equal(res.zero_point, z[f"{name}/zero_point"])
  assert res.quantized_dimension == c["quantized_dimension"] and res.block_size == c["block_size"]


@pytest.mark.parametrize("seed", range(24))
def test_seeded_fake_quantized_weights_match_oracle(seed):
  rng = np.random.default_rng(3000 + seed)
  gran = ["CHANNELWISE", "BLOCKWISE_32", "BLOCKWISE_64", "TENSORWISE"][seed % 4]
  bits = [4, 8][seed % 2]
  rows = int(rng.integers(1, 200))
  cols = int(gran.split("_")[1]) * int(rng.integers(1, 6)) if "BLOCK" in gran else int(rng.integers(1, 1500))
  qmax = 2 ** (bits - 1) - 1
  q = rng.integers(-qmax, qmax + 1, size=(rows, cols)).astype(np.int8)
  if seed % 3 == 0:
    q[rng.random(q.shape) < 0.6] = 0
  if gran == "TENSORWISE":
    w = (q.astype(np.float32) * np.float32(rng.uniform(1e-3, 0.1))).astype(np.float32)
  elif "BLOCK" in gran:
    b = int(gran.split("_")[1])
    sc = rng.uniform(1e-3, 0.1, (rows, cols // b)).astype(np.float32)
    w = (q.reshape(rows, -1, b).astype(np.float32) * sc[:, :, None]).reshape(rows, cols)
  else:
    w = q.astype(np.float32) * rng.uniform(1e-3, 0.1, (rows, 1)).astype(np.float32)
  try:
    ref = O.dwr_quant_params(w, bits, gran)
  except RuntimeError:        # rounding of q * scale to float32 left a step the check rejects
    with pytest.raises(RuntimeError, match="Failed to recover weights"):
      _call(w, bits, gran)
    ref = O.dwr_quant_params(w, bits, gran, check=False)
    res = _call(w, bits, gran, skip_checks=True)
  else:
    res = _call(w, bits, gran)
  assert res.scale.dtype == ref["scale"].dtype and np.array_equal(res.scale, ref["scale"])
  assert np.array_equal(res.quantized_data, ref["quantized_data"])


@pytest.mark.parametrize("rows,cols", [(2, 9000), (3, 16384), (2, 9001), (1, 20000), (2, 40001)])
def test_long_rows_take_the_run_merge_path(rows, cols):
  rng = np.random.default_rng(rows * cols)
  q = rng.integers(-7, 8, size=(rows, cols)).astype(np.int8)
  w = q.astype(np.float32) * rng.uniform(1e-3, 0.1, (rows, 1)).astype(np.float32)
  res, ref = _call(w, 4, "CHANNELWISE"), O.dwr_quant_params(w, 4, "CHANNELWISE")
  assert np.array_equal(res.scale, ref["scale"]) and np.array_equal(res.quantized_data, ref["quantized_data"])
  if rows * cols > 50000:      # ... and the whole tensor as one segment of several runs
    w1 = q.astype(np.float32) * np.float32(0.03)
    res, ref = _call(w1, 4, "TENSORWISE"), O.dwr_quant_params(w1, 4, "TENSORWISE")
    assert np.array_equal(res.scale, ref["scale"]) and np.array_equal(res.quantized_data, ref["quantized_data"])


def test_argument_errors():
  from mi355q.algorithms.uniform_quantize import dequantized_weight_recovery as dwr
  w = np.ones((4, 8), np.float32)
  with pytest.raises(ValueError, match="quantized_dimension must be 0, 1, or None"):
    dwr.get_zp_scale_from_dequantized_symmetric_weights(w, quantized_dimension=2)
  from mi355q import qtyping as q
  cfg = q.TensorQuantizationConfig(num_bits=4, symmetric=False, granularity=q.QuantGranularity.CHANNELWISE)
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  with pytest.raises(ValueError, match="Only symmetric weights"):
    dwr.get_tensor_quant_params(info, cfg, w)


@pytest.mark.parametrize("key", sorted(MODEL_CASES))
def test_fake_quantized_models_match_reference(key):
  from mi355q import quantizer
  from mi355q.utils import tfl_flatbuffer_utils
  case = MODEL_CASES[key]
  qz = quantizer.Quantizer(os.path.join(MODELS, case["model"] + ".tflite"), case["recipe"])
  if "error" in case:
    with pytest.raises(ValueError) as err:
      qz.quantize()
    assert str(err.value)[:150] == case["message"][:150]
    return
  res = qz.quantize()
  got = describe_model(tfl_flatbuffer_utils.read_model(bytes(res.quantized_model)))
  want = case["result"]
  assert got["buffers"] == want["buffers"]
  assert got["subgraphs"] == want["subgraphs"]
  assert got["signatures"] == want["signatures"]
