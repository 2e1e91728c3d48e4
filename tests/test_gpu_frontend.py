"""BASELINE config 1 end to end through the drop-in front end (Quantizer ->
RecipeManager -> ParamsGenerator -> registry materializers -> quantize_tensor) on an
in-memory FC model, against what the REAL reference's orchestration produced for
the same model (tests/golden/gen/make_golden.py::c1_cases), plus calibration flows."""
import numpy as np
import pytest

import parity_rates

from golden_util import case_names, sha
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import algorithm_manager, qtyping, quantizer, recipe, recipe_manager
  from mi355q.algorithms.uniform_quantize import common_quantize, gptq, naive_min_max_quantize
  from mi355q.algorithms.utils import common_utils
  from mi355q.utils import qsv_utils
  return types.SimpleNamespace(am=algorithm_manager, q=qtyping, quantizer=quantizer, recipe=recipe,
                               rm=recipe_manager, cq=common_quantize, gptq=gptq,
                               mm=naive_min_max_quantize, cu=common_utils, qsv=qsv_utils)


def build_fc_model(m, w, bias=None):
  q = m.q

  def tensor(name, shape, buf, typ=q.TensorType.FLOAT32):
    return q.TensorT(name=name.encode(), shape=list(shape), buffer=buf, type=typ)
  model = q.ModelT(version=3)
  model.buffers = [q.BufferT(), q.BufferT(data=w.view(np.uint8).reshape(-1)), q.BufferT()]
  sg = q.SubGraphT()
  sg.tensors = [tensor("x", (1, w.shape[1]), 0), tensor("w", w.shape, 1), tensor("y", (1, w.shape[0]), 2)]
  inputs = [0, 1, -1]
  if bias is not None:
    model.buffers.append(q.BufferT(data=bias.view(np.uint8).reshape(-1)))
    sg.tensors.append(tensor("b", bias.shape, 3))
    inputs = [0, 1, 3]
  sg.operators = [q.OperatorT(inputs=inputs, outputs=[2], opcodeIndex=0)]
  sg.inputs, sg.outputs = [0], [2]
  model.operatorCodes = [q.OperatorCodeT(builtinCode=q.BuiltinOperator.FULLY_CONNECTED,
                                         deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  return model, sg


@pytest.mark.parametrize("name", case_names("c1"))
def test_c1_quantizer_matches_reference_orchestration(m, ref_cases, ref_digests, name):
  arrays, cases = ref_cases
  c = cases[name]
  w = np.random.default_rng(1234).standard_normal((256, 256), dtype=np.float32)
  model, sg = build_fc_model(m, w)
  qz = m.quantizer.Quantizer(model, c["recipe"])
  assert qz.get_quantization_recipe()[0]["algorithm_key"] == c["recipe"][0]["algorithm_key"]
  res = qz.quantize()
  # the float model is untouched; the result is the serialized .tflite, re-read here
  assert int(sg.tensors[1].type) == int(m.q.TensorType.FLOAT32) and sg.tensors[1].quantization is None
  assert np.array_equal(np.asarray(model.buffers[1].data), w.view(np.uint8).reshape(-1))
  from mi355q.utils import tfl_flatbuffer_utils
  model = tfl_flatbuffer_utils.read_model(bytes(res.quantized_model))
  sg = model.subgraphs[0]
  wt = sg.tensors[1]
  assert int(wt.type) == c["tensor_type"]
  buf = np.asarray(model.buffers[1].data).view(np.uint8)
  assert sha(buf) == ref_digests[name]["buffer"]
  assert np.array_equal(buf, arrays[f"{name}/buffer"])
  qp = wt.quantization
  assert qp.quantizedDimension == c["quantized_dimension"]
  assert len(sg.tensors) == c["n_tensors"]
  if "block_size" in c:
    assert qp.detailsType == m.q.QuantizationDetails.BlockwiseQuantization
    assert qp.details.blockSize == c["block_size"] and qp.details.zeroPoints == c["zero_points"]
    st = sg.tensors[qp.details.scales]
    assert st.name.decode() == c["scales_tensor_name"] and int(st.type) == c["scales_tensor_type"]
    assert list(st.shape) == c["scales_tensor_shape"]
    got = np.frombuffer(bytes(np.asarray(model.buffers[st.buffer].data)), dtype=np.float16)
    assert np.array_equal(got, arrays[f"{name}/scales_f16"])
  else:
    assert qp.scale.dtype == np.float32 and np.array_equal(qp.scale, arrays[f"{name}/scale"])
    assert qp.zeroPoint.dtype == np.int64 and np.array_equal(qp.zeroPoint, arrays[f"{name}/zero_point"])
  # the activations stay float
  assert int(sg.tensors[0].type) == int(m.q.TensorType.FLOAT32) and sg.tensors[0].quantization is None


def test_registry_materialize_like_params_generator(m):
  """What ref params_generator.py:162-172 does for one FC op, through our registry."""
  q = m.q
  w = np.random.default_rng(5).standard_normal((32, 64), dtype=np.float32)
  model, sg = build_fc_model(m, w)
  rm = m.rm.RecipeManager()
  rm.load_quantization_recipe(m.recipe.dynamic_wi8_afp32())
  alg, cfg = rm.get_quantization_configs(q.TFLOperationName.FULLY_CONNECTED, "y;")
  assert alg == "min_max_uniform_quantize" and not rm.need_calibration()
  fn = m.am.get_quantization_func(alg, q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.MATERIALIZE)
  cache = m.cu.TensorQuantParamsCache()
  info = q.OpInfo(sg.operators[0], q.TFLOperationName.FULLY_CONNECTED, 0, cfg)
  out = fn(op_info=info, graph_info=q.GraphInfo(sg.tensors, model.buffers), tensor_name_to_qsv={},
           tensor_quant_params_cache=cache)
  assert [p.tensor_name for p in out] == ["x", "w", "y"]
  assert out[0].consumers[0].transformations == [q.QuantTransformation.NO_QUANTIZE]
  assert out[1].consumers[0].transformations == [q.QuantTransformation.QUANTIZE_TENSOR]
  ref = O.min_max_quant_params(w, 8, True, "CHANNELWISE")
  p = out[1].consumers[0].parameters
  assert np.array_equal(p.quantized_data, ref["quantized_data"]) and np.array_equal(p.scale, ref["scale"])
  assert cache.lookup(1, cfg.weight_tensor_config) is p   # cached by (buffer id, config)
  again = fn(op_info=info, graph_info=q.GraphInfo(sg.tensors, model.buffers), tensor_name_to_qsv={},
             tensor_quant_params_cache=cache)
  assert again[1].consumers[0].parameters is p
  with pytest.raises(ValueError, match="Unsupported operation"):
    m.am.get_quantization_func("GPTQ", q.TFLOperationName.SOFTMAX, q.QuantizeMode.MATERIALIZE)
  # activation-only ops are registered for min/max (static recipes quantize them)
  assert callable(m.am.get_quantization_func(alg, q.TFLOperationName.SOFTMAX, q.QuantizeMode.MATERIALIZE))


def test_static_recipe_needs_calibration_and_quantizes_bias(m):
  """SRQ: int8 activations from QSVs, int8 weights, int32 bias with scale s_in * s_w."""
  q = m.q
  rng = np.random.default_rng(6)
  w = rng.standard_normal((16, 32), dtype=np.float32)
  bias = rng.standard_normal(16, dtype=np.float32)
  model, sg = build_fc_model(m, w, bias)
  rm = m.rm.RecipeManager()
  rm.load_quantization_recipe(m.recipe.static_wi8_ai8())
  assert rm.need_calibration()
  from mi355q import params_generator
  with pytest.raises(RuntimeError, match="QSVs"):
    params_generator.ParamsGenerator(model).generate_quantization_parameters(rm)
  qsvs = {"x": {"min": np.array([[-2.0]], np.float32), "max": np.array([[3.0]], np.float32)},
          "y": {"min": np.array([[-7.0]], np.float32), "max": np.array([[9.0]], np.float32)}}
  params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm, qsvs)
  x_p = [c for c in params["x"].consumers if c.subgraph_op_id == 0][0]
  assert x_p.transformations == [q.QuantTransformation.ADD_QUANTIZE]
  zp, sc = O.zp_scale_from_min_max(qsvs["x"]["min"], qsvs["x"]["max"], 8, False, "TENSORWISE")
  assert np.array_equal(x_p.parameters.scale, sc) and np.array_equal(x_p.parameters.zero_point, zp)
  w_p = params["w"].consumers[0].parameters
  wref = O.min_max_quant_params(w, 8, True, "CHANNELWISE")
  assert np.array_equal(w_p.quantized_data, wref["quantized_data"])
  b_link = params["b"].consumers[0]
  assert b_link.transformations == [q.QuantTransformation.QUANTIZE_TENSOR]
  bq, bscale, _, bits, _ = O.quantize_bias(bias, sc, wref["scale"], 8)
  assert b_link.parameters.num_bits == bits == 32
  assert np.array_equal(b_link.parameters.quantized_data, bq)
  assert np.array_equal(b_link.parameters.scale, bscale)
  assert params["y"].producer.transformations == [q.QuantTransformation.ADD_DEQUANTIZE]


def test_min_max_calibrate_and_ema_update(m):
  """calibrator.py:500-587 per-sample flow: calibrate func -> update func."""
  q = m.q
  w = np.zeros((4, 8), np.float32)
  model, sg = build_fc_model(m, w)
  gi = q.GraphInfo(sg.tensors, model.buffers)
  rng = np.random.default_rng(7)
  upd = m.am.get_update_qsv_func("min_max_uniform_quantize", q.TFLOperationName.FULLY_CONNECTED)
  cal = m.am.get_quantization_func("min_max_uniform_quantize", q.TFLOperationName.FULLY_CONNECTED,
                                   q.QuantizeMode.CALIBRATE)
  model_qsvs, ref = {}, {}
  for s in range(5):
    content = {"x": rng.standard_normal((1, 8), dtype=np.float32) * (s + 1),
               "y": rng.standard_normal((1, 4), dtype=np.float32)}
    if s == 2:
      content["x"][0, 0] = np.inf
    op_qsvs = cal(sg.operators[0], gi, content)
    assert set(op_qsvs) == {"x", "y"}  # the constant weight is skipped
    for name, new in op_qsvs.items():
      assert new["min"].shape == (1, 1) and new["num_samples"] == 1
      r = O.activation_qsv(content[name])
      assert new["min"] == r["min"] and new["max"] == r["max"]
      model_qsvs[name] = upd(model_qsvs.get(name), new)
      ref[name] = O.moving_average_update(ref.get(name), r)
  for name in ref:
    assert np.array_equal(model_qsvs[name]["min"], ref[name]["min"])
    assert np.array_equal(model_qsvs[name]["max"], ref[name]["max"])
  qi = m.cq.get_activation_min_max(np.array([1, 2, -10, 10], np.int32))
  assert qi["min"].item() == -10 and qi["max"].item() == 10 and qi["min"].dtype == np.int32


def test_gptq_calibrate_merge_and_quantize(m):
  """GPTQ end to end through the registry: Hessian per sample, weighted merge, OBS apply."""
  q = m.q
  rng = np.random.default_rng(8)
  w = (rng.standard_normal((24, 64)) * 0.05).astype(np.float32)
  model, sg = build_fc_model(m, w)
  gi = q.GraphInfo(sg.tensors, model.buffers)
  alg = "GPTQ"
  cal = m.am.get_quantization_func(alg, q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.CALIBRATE)
  upd = m.am.get_update_qsv_func(alg, q.TFLOperationName.FULLY_CONNECTED)
  assert upd is m.qsv.gptq_and_moving_average_update
  qsvs, ref = {}, None
  for s in range(3):
    x = rng.standard_normal((2 + s, 10, 64), dtype=np.float32)
    y = rng.standard_normal((2 + s, 10, 24), dtype=np.float32)
    out = cal(sg.operators[0], gi, {"x": x, "y": y}, inputs_to_ignore=[1, 2])
    assert out["x"]["hessian"].dtype == np.float64 and out["x"]["num_samples"] == 2 + s
    assert np.max(np.abs(out["x"]["hessian"] - O.gptq_hessian(x))) <= 2e-6 * np.abs(O.gptq_hessian(x)).max()
    r = O.activation_qsv(x)
    r["hessian"] = np.array(out["x"]["hessian"])      # host copy of this sample's own Hessian (merging collects tokens in place)
    for k, v in out.items():
      qsvs[k] = upd(qsvs.get(k), v)
    ref = O.gptq_and_moving_average_update(ref, r)
  assert qsvs["x"]["num_samples"] == ref["num_samples"] == 9
  # the merged statistic is (2/N) X^T X over all 90 tokens in one float32-accumulated product; the oracle's
  # chain merges three products in FP64 (ref utils/qsv_utils.py:71-102): equal but for float32 rounding
  assert np.max(np.abs(np.asarray(qsvs["x"]["hessian"]) - ref["hessian"])) <= 1e-6 * np.abs(ref["hessian"]).max()
  rm = m.rm.RecipeManager()
  rm.add_dynamic_config(".*", q.TFLOperationName.FULLY_CONNECTED, 4, algorithm_key="GPTQ")
  assert rm.need_calibration()
  from mi355q import params_generator
  params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm, qsvs)
  p = params["w"].consumers[0].parameters
  oref = O.gptq_quant_params(w, 4, True, "CHANNELWISE", {"activation_tensor_qsv": ref})
  assert np.array_equal(p.scale, oref["scale"])
  parity_rates.check("gptq through ParamsGenerator (frontend)", p.quantized_data, oref["quantized_data"], parity_rates.T2)


def test_hadamard_recipe_materializes_rotation_instruction(m):
  q = m.q
  w = np.random.default_rng(9).standard_normal((16, 64), dtype=np.float32)
  bias = np.zeros(16, np.float32)
  model, sg = build_fc_model(m, w, bias)
  rm = m.rm.RecipeManager()
  rm.load_quantization_recipe(m.recipe.dynamic_wi8c_hr_afp32(operation_name=q.TFLOperationName.FULLY_CONNECTED))
  from mi355q import params_generator
  params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm)
  assert params["x"].consumers[0].transformations == [q.QuantTransformation.INSERT_DECOMPOSED_HADAMARD_ROTATION]
  wp = params["w"].consumers[0]
  assert wp.transformations == [q.QuantTransformation.QUANTIZE_TENSOR]
  assert wp.parameters.hadamard.hadamard_size == 64
  n_ops = len(sg.operators)
  m.quantizer.apply_quantize_tensor_transformations(model, params)
  assert len(sg.operators) == n_ops + 3              # RESHAPE -> FC(H / sqrt(h)) -> RESHAPE in front of the FC
  codes = [model.operatorCodes[op.opcodeIndex].builtinCode for op in sg.operators]
  B = q.BuiltinOperator
  assert codes == [B.RESHAPE, B.FULLY_CONNECTED, B.RESHAPE, B.FULLY_CONNECTED]
  hm = sg.tensors[sg.operators[1].inputs[1]]
  H = np.asarray(model.buffers[hm.buffer].data).view(np.float32).reshape(64, 64)
  assert np.allclose(H @ H.T, np.eye(64), atol=1e-6) and sg.operators[3].inputs[0] == sg.operators[2].outputs[0]
  # the custom-op form: ONE "aeq.hadamard_rotation" op in front of the FC, its options a FlexBuffer map
  # (ref transformations/insert_hadamard_rotation.py; tests/test_hadamard_custom_op.py has the details)
  from mi355q.utils import flexbuffer
  rm2 = m.rm.RecipeManager()
  rm2.load_quantization_recipe(m.recipe.dynamic_wi8_afp32(algorithm_key="HADAMARD_ROTATION"))
  model2, sg2 = build_fc_model(m, w, bias)
  params2 = params_generator.ParamsGenerator(model2).generate_quantization_parameters(rm2)
  assert params2["x"].consumers[0].transformations == [q.QuantTransformation.INSERT_HADAMARD_ROTATION]
  n_ops2 = len(sg2.operators)
  m.quantizer.apply_quantize_tensor_transformations(model2, params2)
  assert len(sg2.operators) == n_ops2 + 1
  codes2 = [model2.operatorCodes[op.opcodeIndex].builtinCode for op in sg2.operators]
  assert codes2 == [B.CUSTOM, B.FULLY_CONNECTED] and sg2.operators[1].inputs[0] == sg2.operators[0].outputs[0]
  opts = flexbuffer.decode(bytes(np.asarray(sg2.operators[0].customOptions, np.uint8)))
  hp = params2["w"].consumers[0].parameters.hadamard
  assert opts == {"hadamard_size": int(hp.hadamard_size), "random_binary_vector": np.asarray(hp.random_binary_vector).tolist()}



@pytest.mark.parametrize("alg", ["min_max_uniform_quantize", "GPTQ", "OSCAR"])
def test_calibration_samples_resident_in_hbm_give_the_same_qsvs(m, alg):
  """Samples handed over as device tensors (activations produced on this GPU) are calibrated where
  they are: same QSVs, bit for bit, as the same samples handed over as host arrays."""
  import torch
  from mi355q import calibrator, runtime as rt
  q = m.q
  rng = np.random.default_rng(61)
  w = (rng.standard_normal((24, 64)) * 0.05).astype(np.float32)
  model, _ = build_fc_model(m, w)
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0)]
  rm = m.rm.RecipeManager()
  if alg == "min_max_uniform_quantize":
    rm.load_quantization_recipe(m.recipe.static_wi8_ai8())
  else:
    rm.add_dynamic_config(".*", q.TFLOperationName.FULLY_CONNECTED, 4, algorithm_key=alg)
  host = [{"x": rng.standard_normal((3, 7, 64), dtype=np.float32) * np.float32(1 + s),
           "y": rng.standard_normal((3, 7, 24), dtype=np.float32)} for s in range(4)]
  host[2]["x"][0, 0, 0] = np.inf
  resident = [{k: torch.from_numpy(v).cuda() for k, v in s.items()} for s in host]
  resident[1]["y"] = rt.HbmArray(resident[1]["y"])        # both spellings of "already in HBM"
  resident[3]["x"] = torch.from_numpy(host[3]["x"])        # a host torch tensor is accepted as well
  out = []
  for data in (host, resident):
    cal = calibrator.Calibrator(model)
    cal.calibrate({"serving_default": data}, rm)
    out.append(cal.get_model_qsvs())
  assert set(out[0]) == set(out[1]) and out[0]
  for name, qsv in out[0].items():
    assert set(qsv) == set(out[1][name])
    for key, val in qsv.items():
      assert np.array_equal(np.asarray(val), np.asarray(out[1][name][key])), (name, key)
