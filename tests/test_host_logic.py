"""CPU-only tests of the host side of the drop-in layer: value types, registry,
recipe resolution, the O(#scales) parameter math and argument validation that
happens before any kernel is launched."""
import os

import numpy as np
import pytest

from golden_util import case_names
from oracle import aeq_oracle as O

from mi355q import algorithm_manager as am
from mi355q import algorithm_manager_api
from mi355q import qtyping as q
from mi355q import recipe, recipe_manager
from mi355q.algorithms.uniform_quantize import (hadamard_rotation, mse, naive_min_max_quantize,
                                                octav, uniform_quantize_tensor as uqt)
from mi355q.algorithms.utils import common_utils
from mi355q.transformations import quantize_tensor
from mi355q.utils import qsv_utils, tfl_flatbuffer_utils


def test_uniform_quant_params_value_equality():
  a = q.UniformQuantParams(8, 0, np.array([1.0, 2.0], np.float32), np.zeros(2, np.int8),
                           quantized_data=np.arange(4, dtype=np.int8))
  b = q.UniformQuantParams(8, 0, np.array([1.0, 2.0], np.float32), np.zeros(2, np.int8),
                           quantized_data=np.arange(4, dtype=np.int8))
  assert a == b and a is not b
  assert a != q.UniformQuantParams(8, 0, np.array([1.0, 2.5], np.float32), np.zeros(2, np.int8))
  h = q.UniformQuantParams.HadamardRotationParams
  assert h(np.ones(4, np.int8), 4) == h(np.ones(4, np.int8), 4)
  assert h(np.ones(4, np.int8), 4) != h(np.ones(4, np.int8), 8)
  with pytest.raises(TypeError):
    hash(a)
  with pytest.raises(Exception):
    a.num_bits = 4  # frozen


def test_tensor_quantization_config_dict_round_trip_and_legacy_block_size():
  c = q.TensorQuantizationConfig.from_dict({"num_bits": 4, "block_size": 64, "max_hadamard_size": 128})
  assert c.granularity == q.QuantGranularity.BLOCKWISE_64
  assert c.algorithm_params == {"max_hadamard_size": 128} and hash(c) is not None
  assert q.TensorQuantizationConfig.from_dict(c.to_dict()) == c
  with pytest.raises(ValueError, match="Unsupported block size"):
    q.TensorQuantizationConfig.from_dict({"num_bits": 4, "block_size": 48})
  with pytest.raises(ValueError, match="integer activation but float weights"):
    q.OpQuantizationConfig(activation_tensor_config=q.TensorQuantizationConfig(8),
                           weight_tensor_config=q.TensorQuantizationConfig(16, dtype=q.TensorDataType.FLOAT))
  with pytest.raises(ValueError, match="must be SRQ"):
    q.OpQuantizationConfig(activation_tensor_config=q.TensorQuantizationConfig(8),
                           weight_tensor_config=q.TensorQuantizationConfig(8))


def test_registry_api_surface():
  api = algorithm_manager_api.AlgorithmManagerApi()
  fns = dict(init_qsv_func=lambda *a, **k: {}, calibration_func=lambda *a, **k: {"c": 1},
             materialize_func=lambda *a, **k: ["m"])
  api.register_quantized_op("alg", q.TFLOperationName.FULLY_CONNECTED, **fns)
  assert api.is_algorithm_registered("alg") and api.is_op_registered("alg", q.TFLOperationName.FULLY_CONNECTED)
  assert api.get_supported_ops("alg") == [q.TFLOperationName.FULLY_CONNECTED]
  assert api.get_quantization_func("alg", q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.MATERIALIZE)() == ["m"]
  assert api.get_quantization_func("alg", q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.CALIBRATE)() == {"c": 1}
  assert api.get_update_qsv_func("alg", q.TFLOperationName.FULLY_CONNECTED) is qsv_utils.moving_average_update
  with pytest.raises(ValueError, match="Unsupported operation"):
    api.get_quantization_func("alg", q.TFLOperationName.CONV_2D, q.QuantizeMode.MATERIALIZE)
  with pytest.raises(ValueError, match="Unregistered algorithm"):
    api.get_supported_ops("nope")
  with pytest.raises(ValueError, match="Config checking function"):
    api.check_op_quantization_config("alg", q.TFLOperationName.FULLY_CONNECTED, q.OpQuantizationConfig())
  api.check_op_quantization_config("alg", q.TFLOperationName.CONV_2D, q.OpQuantizationConfig(skip_checks=True))


def test_module_registry_has_every_hot_path_algorithm():
  for name in ("min_max_uniform_quantize", "OCTAV", "MSE", "GPTQ", "HADAMARD_ROTATION",
               "DECOMPOSED_HADAMARD_ROTATION"):
    assert am.is_algorithm_registered(name)
    assert am.is_op_registered(name, q.TFLOperationName.FULLY_CONNECTED)
  assert am.AlgorithmName("OCTAV") is am.AlgorithmName.OCTAV
  assert am.get_update_qsv_func("GPTQ", q.TFLOperationName.FULLY_CONNECTED) is \
      qsv_utils.gptq_and_moving_average_update
  fn = am.get_quantization_func("OCTAV", q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.MATERIALIZE)
  assert fn.args == (octav.get_tensor_quant_params,)  # functools.partial(materialize, get_tensor_quant_params)


def test_recipe_resolution_last_valid_match_wins():
  rm = recipe_manager.RecipeManager()
  rm.add_dynamic_config(".*", q.TFLOperationName.ALL_SUPPORTED, 8)
  rm.add_dynamic_config("attn", q.TFLOperationName.FULLY_CONNECTED, 4,
                        granularity=q.QuantGranularity.BLOCKWISE_32, algorithm_key="OCTAV")
  rm.add_quantization_config("skip_me", q.TFLOperationName.FULLY_CONNECTED, algorithm_key="no_quantize")
  FC = q.TFLOperationName.FULLY_CONNECTED
  assert rm.get_quantization_configs(FC, "mlp/out;")[0] == "min_max_uniform_quantize"
  alg, cfg = rm.get_quantization_configs(FC, "layer0/attn/q;")
  assert alg == "OCTAV" and cfg.weight_tensor_config.granularity == q.QuantGranularity.BLOCKWISE_32
  assert rm.get_quantization_configs(FC, "attn/skip_me;")[0] == "no_quantize"
  # the policy has no DRQ entry for the virtual INPUT op (ref default_policy.py:278-289)
  assert rm.get_quantization_configs(q.TFLOperationName.INPUT, "attn;")[0] == "no_quantize"
  assert rm.get_quantization_configs(q.TFLOperationName.SOFTMAX, "x;")[0] == "no_quantize"
  rt = recipe_manager.RecipeManager()
  rt.load_quantization_recipe(rm.get_quantization_recipe())
  assert rt.get_quantization_recipe() == rm.get_quantization_recipe()
  assert not rm.need_calibration()
  rs = recipe_manager.RecipeManager()
  rs.load_quantization_recipe(recipe.static_wi8_ai8())
  assert rs.need_calibration()
  with pytest.raises(ValueError, match="Unsupported algorithm key"):
    rm.add_quantization_config(".*", FC, algorithm_key="bogus")
  assert recipe.dynamic_wi4b32_afp32()[0]["op_config"]["weight_tensor_config"]["granularity"] == \
      q.QuantGranularity.BLOCKWISE_32


@pytest.mark.parametrize("name", case_names("min_max"))
def test_host_zp_scale_math_matches_reference(ref_cases, name):
  """tensor_zp_scale_from_min_max (host NumPy, O(#scales)) against the reference's outputs."""
  arrays, cases = ref_cases
  c = cases[name]
  w = arrays[f"{name}/w"]
  mmv = O.init_tensor_min_max(w, c["granularity"], c["quantized_dimension"])
  import warnings
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    zp, scale = uqt.tensor_zp_scale_from_min_max(mmv["min"], mmv["max"], c["num_bits"], c["symmetric"],
                                                 q.QuantGranularity[c["granularity"]])
  assert np.array_equal(scale, arrays[f"{name}/scale"], equal_nan=True)
  assert np.array_equal(zp, arrays[f"{name}/zero_point"]) and zp.dtype == arrays[f"{name}/zero_point"].dtype


def test_host_helpers_and_validation_errors():
  assert uqt.get_quantized_range(uqt.IntType(4, True)) == (-8.0, 7.0)
  assert uqt.extract_block_size_from_granularity(q.QuantGranularity.BLOCKWISE_128) == 128
  assert uqt.extract_block_size_from_granularity(q.QuantGranularity.CHANNELWISE) == 0
  assert uqt._channel_view((8, 3, 3, 16), (8, 1, 1, 1)) == (1, 8, 144)
  assert uqt._channel_view((1, 3, 3, 24), (1, 1, 1, 24)) == (9, 24, 1)
  assert uqt._channel_view((4, 5), (1, 1)) == (1, 1, 20)
  with pytest.raises(ValueError, match="not adjacent"):
    uqt._channel_view((4, 5, 6), (4, 1, 6))
  # ... which is why such parameters are repeated over the dimensions in between first (NumPy broadcasting, ref :273-409)
  sc = np.arange(24, dtype=np.float32).reshape(4, 1, 6) + 1
  zpt = np.arange(24, dtype=np.int8).reshape(4, 1, 6)
  s2, z2 = uqt._adjacent_params((4, 5, 6), sc, zpt)
  assert s2.shape == z2.shape == (4, 5, 6) and np.array_equal(s2, np.broadcast_to(sc, (4, 5, 6)))
  assert np.array_equal(z2, np.broadcast_to(zpt, (4, 5, 6))) and uqt._channel_view((4, 5, 6), s2.shape) == (1, 120, 1)
  s3, z3 = uqt._adjacent_params((2, 4, 5, 6), sc.reshape(1, 4, 1, 6), np.zeros((1, 1, 1, 1), np.int8))
  assert s3.shape == (1, 4, 5, 6) and z3.shape == (1, 1, 1, 1) and uqt._channel_view((2, 4, 5, 6), s3.shape) == (2, 120, 1)
  for keep in (sc.reshape(4, 6, 1), sc.reshape(1, 4, 6)[:, :, :1]):
    assert uqt._adjacent_params((4, 6, 7), keep, None)[0] is keep
  x = np.array([-3.0, 1.3, 2.4, 16.0])
  p = q.UniformQuantParams(4, 0, np.array([[[1.2666667]]]), np.array([[-6]]))
  with pytest.raises(ValueError, match=r"Ranks of scales \(3\) and zps \(2\)"):
    uqt.uniform_quantize(x, p)
  with pytest.raises(ValueError, match="zero_points need to be"):
    uqt.uniform_quantize(x, q.UniformQuantParams(8, 0, np.array([1.0]), np.array([0.5])))
  with pytest.raises(ValueError, match="single element for scalar tensor"):
    uqt.fix_quantization_params_rank(np.array(6.66), q.UniformQuantParams(8, 0, np.ones(2), np.zeros(2, np.int8)))
  with pytest.raises(TypeError, match="not exactly representable"):
    uqt._as_f32_exact(np.array([0.1], np.float64))
  assert uqt._as_f32_exact(np.array([0.5, 3.0], np.float64)).dtype == np.float32
  assert np.array_equal(uqt.round_to_bf16(np.array([1.0, 1.00390625, 1.01171875], np.float32)),
                        O.round_to_bf16(np.array([1.0, 1.00390625, 1.01171875], np.float32)))


def test_algorithm_argument_errors_raise_before_any_gpu_work():
  cfg = q.TensorQuantizationConfig(num_bits=4, symmetric=False, granularity=q.QuantGranularity.CHANNELWISE)
  info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=3,
                  op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
  w = np.ones((4, 4), np.float32)
  with pytest.raises(ValueError, match="Unsupported symmetry"):
    octav.get_tensor_quant_params(info, cfg, w)
  with pytest.raises(ValueError, match="Unsupported symmetry"):
    mse.get_tensor_quant_params(info, cfg, w)
  bcfg = q.TensorQuantizationConfig(num_bits=4, granularity=q.QuantGranularity.BLOCKWISE_32)
  with pytest.raises(ValueError, match="Blockwise quantization is not supported for MSE"):
    mse.get_tensor_quant_params(info, bcfg, w)
  with pytest.raises(ValueError, match="only supported for weight tensors"):
    hadamard_rotation.get_tensor_quant_params(info, cfg, None)
  with pytest.raises(ValueError, match="static quantization"):
    hadamard_rotation.get_tensor_quant_params(info, cfg, w, {})
  with pytest.raises(ValueError, match="rank >= 2"):
    hadamard_rotation.get_tensor_quant_params(info, cfg, np.ones(4, np.float32))
  with pytest.raises(ValueError, match=r"FULLY_CONNECTED\(index: 3\) not found in tensor_name_to_qsv"):
    naive_min_max_quantize.get_tensor_quant_params(info, cfg, None, None)
  with pytest.raises(ValueError, match="power of 2"):
    hadamard_rotation._make_hadamard_matrix(12)
  assert hadamard_rotation.hadamard_size_for(11008) == 256
  assert hadamard_rotation.hadamard_size_for(4096, 100) == 64
  assert np.array_equal(hadamard_rotation._make_hadamard_matrix(8), O.hadamard_matrix(8))


def test_activation_params_from_qsv_need_no_gpu():
  """tensor_content=None (activations): pure host math, same as the reference."""
  for bits, sym in ((8, False), (8, True), (16, True)):
    cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=sym)
    info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                    op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
    qsv = {"min": np.array([[-3.25]], np.float32), "max": np.array([[7.5]], np.float32)}
    p = naive_min_max_quantize.get_tensor_quant_params(info, cfg, None, qsv)
    zp, scale = O.zp_scale_from_min_max(qsv["min"], qsv["max"], bits, sym, "TENSORWISE")
    assert p.quantized_data is None and np.array_equal(p.scale, scale) and np.array_equal(p.zero_point, zp)


def test_transformation_table_and_dim_helpers():
  T = q.QuantTransformation
  w8 = q.TensorQuantizationConfig(8, granularity=q.QuantGranularity.CHANNELWISE)
  drq = q.OpQuantizationConfig(weight_tensor_config=w8, compute_precision=q.ComputePrecision.INTEGER)
  wo = q.OpQuantizationConfig(weight_tensor_config=w8, explicit_dequantize=True)
  srq = q.OpQuantizationConfig(activation_tensor_config=q.TensorQuantizationConfig(8, symmetric=False),
                               weight_tensor_config=w8, compute_precision=q.ComputePrecision.INTEGER)
  g = common_utils.get_tensor_transformations
  assert g(drq, True, True) == [T.QUANTIZE_TENSOR] and g(drq, True, False) == [T.NO_QUANTIZE]
  assert g(wo, True, True) == [T.ADD_DEQUANTIZE] and g(wo, False, False) == [T.NO_QUANTIZE]
  assert g(srq, True, False) == [T.ADD_QUANTIZE] and g(srq, True, True) == [T.QUANTIZE_TENSOR]
  assert g(srq, False, False) == [T.ADD_DEQUANTIZE]
  with pytest.raises(ValueError, match="Unsupported compute precision"):
    g(q.OpQuantizationConfig(weight_tensor_config=w8), True, True)
  assert common_utils.get_reduce_dims(0, (4, 5, 6)) == (1, 2) and common_utils.get_reduce_dims(None, (4,)) is None
  info = q.OpInfo(q.OperatorT(), q.TFLOperationName.DEPTHWISE_CONV_2D, 0, drq)
  assert common_utils.get_weight_quantized_dim(info, np.zeros((1, 3, 3, 8)), q.QuantGranularity.CHANNELWISE) == 3
  assert tfl_flatbuffer_utils.TFL_OP_TO_BLOCKWISE_WEIGHT_QUANTIZED_DIM[q.TFLOperationName.EMBEDDING_LOOKUP] == 1
  with pytest.raises(ValueError, match="Unsupported op for blockwise"):
    common_utils.check_subchannel_config(
        q.TFLOperationName.CONV_2D,
        q.OpQuantizationConfig(weight_tensor_config=q.TensorQuantizationConfig(
            4, granularity=q.QuantGranularity.BLOCKWISE_32)))
  assert quantize_tensor.quant_params_to_tflite_type(4) == q.TensorType.INT4
  assert quantize_tensor.quant_params_to_tflite_type(2) == q.TensorType.INT2
  assert quantize_tensor.quant_params_to_tflite_type(8) == q.TensorType.INT8
  assert quantize_tensor.quant_params_to_tflite_type(32) == q.TensorType.INT32
  with pytest.raises(ValueError, match="Unsupported bitwidth"):
    quantize_tensor.quant_params_to_tflite_type(128)


def test_op_scope_and_tensor_data_views():
  w = np.arange(12, dtype=np.float32).reshape(3, 4)
  t = [q.TensorT(name=b"in", shape=[1, 4], buffer=0), q.TensorT(name=b"w", shape=[3, 4], buffer=1),
       q.TensorT(name=b"out/a", shape=[1, 3], buffer=0), q.TensorT(name=b"out/b", shape=[1, 3], buffer=0)]
  bufs = [q.BufferT(), q.BufferT(data=w.view(np.uint8).reshape(-1))]
  op = q.OperatorT(inputs=[0, 1, -1], outputs=[2, 3])
  assert tfl_flatbuffer_utils.get_op_scope(op, t) == "out/a;out/b;"
  assert tfl_flatbuffer_utils.get_op_scope(q.OperatorT(inputs=[0], outputs=[]), t) == "in;"
  assert tfl_flatbuffer_utils.get_tensor_data(t[0], bufs) is None
  view = tfl_flatbuffer_utils.get_tensor_data(t[1], bufs)
  assert np.array_equal(view, w) and np.shares_memory(view, bufs[1].data)  # zero copy


def test_config_check_policy_equals_reference_acceptance_set():
  """default_policy.py's rule form accepts exactly the (op, config) pairs the reference's
  unrolled JSON policy holds (tests/golden/ref_policy.json, recorded from the real reference),
  and nothing else over the whole config grid."""
  import itertools
  import json as _json
  import os as _os
  from mi355q import default_policy
  path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "ref_policy.json")
  ref = {op: {tuple(r) for r in rows} for op, rows in _json.load(open(path))["policy"].items()}
  pol = default_policy.DEFAULT_CONFIG_CHECK_POLICY
  assert {o.value for o in pol.keys()} == set(ref)
  G, P = q.QuantGranularity, q.ComputePrecision
  grans = [G.TENSORWISE, G.CHANNELWISE, G.BLOCKWISE_32, G.BLOCKWISE_64, G.BLOCKWISE_128, G.BLOCKWISE_256]
  n_ok = 0
  for op in q.TFLOperationName:
    for act in [None] + [(b, s, g) for b in (8, 16) for s in (True, False) for g in (G.TENSORWISE, G.CHANNELWISE)]:
      for wb, ws, wg, prec, dq in itertools.product((2, 4, 8), (True, False), grans, (P.INTEGER, P.FLOAT), (True, False)):
        if act is not None and prec != P.INTEGER:
          continue                       # rejected by OpQuantizationConfig itself
        cfg = q.OpQuantizationConfig(
            activation_tensor_config=None if act is None else q.TensorQuantizationConfig(act[0], act[1], act[2]),
            weight_tensor_config=q.TensorQuantizationConfig(wb, ws, wg), compute_precision=prec,
            explicit_dequantize=dq, min_weight_elements=7)
        row = (None if act is None else act[0], None if act is None else act[1],
               None if act is None else act[2].value, wb, ws, wg.value, prec.value, dq)
        want = row in ref.get(op.value, ())
        assert pol.accepts(op, cfg) == want, (op, row)
        n_ok += want
  assert n_ok == sum(len(v) for v in ref.values())
  with pytest.raises(ValueError, match="Unsupported op for"):
    default_policy.check_if_valid_op_config(
        q.TFLOperationName.BATCH_MATMUL,
        q.OpQuantizationConfig(weight_tensor_config=q.TensorQuantizationConfig(4, True, G.CHANNELWISE),
                               compute_precision=P.INTEGER), pol)
  with pytest.raises(ValueError, match="No policy was specified at all"):
    default_policy.check_if_valid_op_config(q.TFLOperationName.FULLY_CONNECTED, q.OpQuantizationConfig(), None)


def test_transformation_instructions_match_reference_known_answers():
  """256 synthetic (producer, consumers) parameter combinations for one tensor of
  branching_conv_fc: horizontal grouping, DQ.Q cancellation, requantization, DQ kept for float
  consumers, validity errors - all as the real reference's generator answers
  (tests/golden/ref_instruction_cases.json)."""
  import json as _json
  import os as _os
  from mi355q import transformation_instruction_generator as tig
  here = _os.path.dirname(_os.path.abspath(__file__))
  ref = _json.load(open(_os.path.join(here, "golden", "ref_instruction_cases.json")))
  model = tfl_flatbuffer_utils.read_model(_os.path.join(here, "golden", "models", ref["model"] + ".tflite"))
  T = q.QuantTransformation

  def qp(k):
    return q.UniformQuantParams(num_bits=8, quantized_dimension=None, scale=np.array([0.1 * (k + 1)], np.float32),
                                zero_point=np.array([k], np.int64), symmetric=False)
  P = [qp(0), qp(1)]

  def link(op_id, spec):
    names, k = spec
    return q.OpToTensorParams(subgraph_op_id=op_id, transformations=[T[n] for n in names],
                              parameters=None if k is None else P[k])
  n_err = 0
  for case in ref["cases"]:
    gen = tig.TransformationInstructionsGenerator(model)       # fresh graph info (lists are consumed)
    param = q.TensorTransformationParams(
        tensor_name=ref["tensor"],
        producer=None if case["producer"] is None else link(ref["producer_op"], case["producer"]),
        consumers=[link(op, spec) for op, spec in zip(ref["op_ids"], case["consumers"])])
    if "error" in case:
      n_err += 1
      with pytest.raises(ValueError):
        gen.tensor_instructions(param)
      continue
    got = [[i.transformation.name, i.tensor_id, i.producer, list(i.consumers),
            None if i.parameters is None else int(np.asarray(i.parameters.zero_point).ravel()[0])]
           for i in gen.tensor_instructions(param).instructions]
    assert got == case["instructions"], case
  assert 0 < n_err < len(ref["cases"])


def test_calibrator_bookkeeping_without_gpu(tmp_path):
  """Signature lookup, sample typing, QSV save / load - the parts of the calibrator that do
  not touch the GPU."""
  import os as _os
  from mi355q import calibrator
  here = _os.path.dirname(_os.path.abspath(__file__))
  c = calibrator.Calibrator(_os.path.join(here, "golden", "models", "two_signatures.tflite"))
  assert sorted(c.get_signature_list()) == ["add", "multiply"]
  with pytest.raises(ValueError, match="signature_key is required"):
    c._main_subgraph(None)
  with pytest.raises(ValueError, match="not found"):
    c._main_subgraph("nope")
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(recipe.static_wi8_ai8())
  with pytest.raises(TypeError, match="tensor name"):
    c.calibrate({"add": [np.zeros(3, np.float32)]}, rm)
  c.load_model_qsvs({"t": {"min": np.array([[-1.0]], np.float32), "max": np.array([[2.0]], np.float32)}})
  p = str(tmp_path / "qsv.json")
  c.save_calibration_result(p, {"note": "x"})
  d = calibrator.Calibrator(_os.path.join(here, "golden", "models", "two_signatures.tflite"))
  d.load_model_qsvs(p)
  assert d.get_model_qsvs()["t"]["max"].dtype == np.float32 and d.get_model_qsvs()["t"]["max"][0, 0] == 2.0
  assert d._metadata["note"] == "x" and d._metadata["num_samples_calibrated"] == 1   # counted before the step ran, as in the reference
  d.reset_model_qsvs()
  assert d.get_model_qsvs() == {}


def _fc_model_for_multiply():
  """x[1,8] -> FC(w0) -> y0 and x -> FC(w1) -> y1 (two consumers of one activation)."""
  tensors = [q.TensorT(name=b"x", shape=[1, 8], type=0, buffer=0),
             q.TensorT(name=b"w0", shape=[4, 8], type=0, buffer=1),
             q.TensorT(name=b"w1", shape=[4, 8], type=0, buffer=2),
             q.TensorT(name=b"y0", shape=[1, 4], type=0, buffer=0),
             q.TensorT(name=b"y1", shape=[1, 4], type=0, buffer=0)]
  ops = [q.OperatorT(opcodeIndex=0, inputs=[0, 1, -1], outputs=[3]),
         q.OperatorT(opcodeIndex=0, inputs=[0, 2, -1], outputs=[4])]
  sg = q.SubGraphT(tensors=tensors, operators=ops, inputs=[0], outputs=[3, 4])
  w = np.arange(32, dtype=np.float32)
  return q.ModelT(version=3, subgraphs=[sg], operatorCodes=[q.OperatorCodeT(builtinCode=int(q.BuiltinOperator.FULLY_CONNECTED))],
                  buffers=[q.BufferT(), q.BufferT(data=w.view(np.uint8)), q.BufferT(data=(w + 1).view(np.uint8))])


def test_insert_multiply_edits_the_graph_like_the_reference():
  """ref: transformations/insert_multiply_test.py (graph shape, sharing, error behaviour)."""
  from mi355q.transformations import graph_edits, transformation_utils
  mult = np.linspace(0.5, 2.0, 8).astype(np.float32)
  params = q.UniformQuantParams(num_bits=4, quantized_dimension=0, scale=np.ones((4, 1), np.float32),
                                zero_point=np.zeros((4, 1), np.int8), custom_algorithm_param={"multiplier": mult})
  model = _fc_model_for_multiply()
  sg = model.subgraphs[0]

  def ti(consumers, tensor_id=0, p=params):
    return transformation_utils.TransformationInput(tensor_id=tensor_id, model=model, subgraph=sg, producer=-1,
                                                    consumers=consumers, quant_params=p)
  info = graph_edits.insert_multiply(ti([0]))
  assert (info.op_id, info.num_ops_added) == (0, 1) and len(sg.tensors) == 7 and len(sg.operators) == 3
  mul = sg.operators[0]
  assert model.operatorCodes[mul.opcodeIndex].builtinCode == q.BuiltinOperator.MUL
  assert mul.inputs[0] == 0 and mul.outputs == [info.output_tensor_id]
  assert mul.builtinOptionsType == q.BuiltinOptions.MulOptions and mul.builtinOptions.fusedActivationFunction == 0
  mt = sg.tensors[mul.inputs[1]]
  assert mt.type == q.TensorType.FLOAT32 and mt.name == b"x_multiplier" and list(mt.shape) == [8]
  assert np.array_equal(tfl_flatbuffer_utils.get_tensor_data(mt, model.buffers), mult)
  assert sg.tensors[info.output_tensor_id].name == b"x_scaled" and list(sg.tensors[info.output_tensor_id].shape) == [1, 8]
  assert sg.operators[1].inputs[0] == info.output_tensor_id       # the FC consumer was re-pointed
  assert sg.operators[2].inputs[0] == 0                            # the other one was not asked
  # a second insertion with the same vector shares the constant tensor
  info2 = graph_edits.insert_multiply(ti([2]))
  assert sg.operators[info2.op_id].inputs[1] == mul.inputs[1]
  # the edited model serializes and re-reads (typed MulOptions table)
  from mi355q.utils import tflite_flatbuffer as fb
  again = fb.read_model(bytes(fb.write_model(model)))
  assert [o.builtinOptionsType for o in again.subgraphs[0].operators].count(int(q.BuiltinOptions.MulOptions)) == 2
  # errors
  with pytest.raises(ValueError, match="uniform quantization only"):
    graph_edits.insert_multiply(ti([0], p=q.NonLinearQuantParams(num_bits=16, quantized_data=None)))
  with pytest.raises(ValueError, match='"multiplier" is not set'):
    graph_edits.insert_multiply(ti([0], p=q.UniformQuantParams(num_bits=4, quantized_dimension=0, scale=np.ones(1, np.float32),
                                                                zero_point=np.zeros(1, np.int8))))
  sg.tensors[0].type = int(q.TensorType.INT8)
  with pytest.raises(ValueError, match="float32 tensors only"):
    graph_edits.insert_multiply(ti([0]))
  sg.tensors[0].type = 0
  model.operatorCodes[0].builtinCode = int(q.BuiltinOperator.ADD)
  with pytest.raises(ValueError, match="fully connected consumers only"):
    graph_edits.insert_multiply(ti([1]))


def test_oscar_registration_and_qsv_merge():
  from mi355q.algorithms.uniform_quantize import oscar
  assert am.AlgorithmName.OSCAR == "OSCAR" and am.get_supported_ops("OSCAR") == [q.TFLOperationName.FULLY_CONNECTED]
  assert am.get_update_qsv_func("OSCAR", q.TFLOperationName.FULLY_CONNECTED) is qsv_utils.oscar_and_moving_average_update
  assert am.get_quantization_func("OSCAR", q.TFLOperationName.FULLY_CONNECTED, q.QuantizeMode.CALIBRATE) is oscar.calibrate
  a = {"min": np.float32(-1), "max": np.float32(2), "mu2": np.array([1.0, 3.0]), "num_samples": 2}
  b = {"min": np.float32(-3), "max": np.float32(1), "mu2": np.array([5.0, 1.0]), "num_samples": 6}
  got, want = qsv_utils.oscar_and_moving_average_update(a, b), O.oscar_and_moving_average_update(a, b)
  assert np.array_equal(got["mu2"], want["mu2"]) and got["num_samples"] == 8
  assert got["min"] == want["min"] and got["max"] == want["max"]
  assert qsv_utils.oscar_and_moving_average_update(None, b) is b


def test_quantizer_recipe_editing_api_and_policy_file(tmp_path):
  """Quantizer's recipe-editing methods (ref quantizer.py:207-352) and a user policy .json."""
  import json as _json
  from mi355q import default_policy, quantizer
  model_path = str(tmp_path / "m.tflite")
  from mi355q.utils import tflite_flatbuffer as fb
  open(model_path, "wb").write(fb.write_model(_fc_model_for_multiply()))
  qz = quantizer.Quantizer(model_path)
  assert not qz.need_calibration and not qz.need_calibration()         # property (reference) and call
  qz.add_dynamic_config(".*", q.TFLOperationName.FULLY_CONNECTED, num_bits=4)
  qz.add_weight_only_config("w1", q.TFLOperationName.FULLY_CONNECTED, num_bits=8)
  assert [r["op_config"]["compute_precision"] for r in qz.get_quantization_recipe()] == ["INTEGER", "FLOAT"]
  qz.add_static_config(".*", q.TFLOperationName.ALL_SUPPORTED, activation_num_bits=8, weight_num_bits=8)
  assert qz.need_calibration and qz.need_calibration() is True
  qz.update_quantization_recipe(".*", q.TFLOperationName.FULLY_CONNECTED, algorithm_key="no_quantize")
  assert [r["algorithm_key"] for r in qz.get_quantization_recipe() if r["operation"] == "FULLY_CONNECTED"
          and r["regex"] == ".*"] == ["no_quantize"]
  assert recipe.dynamic_legacy_wi8_afp32()[0]["op_config"]["min_weight_elements"] == 1024
  # a policy file: int8 weight-only for FULLY_CONNECTED and nothing else
  policy_json = _json.dumps({
      "configs": {"only": {"weight_tensor_config": {"num_bits": 8, "symmetric": [True],
                                                    "granularity": ["CHANNELWISE", "TENSORWISE"], "dtype": "INT"},
                           "explicit_dequantize": True, "compute_precision": "FLOAT"}},
      "ops_per_config": {"only": ["FULLY_CONNECTED"]}})
  policy = default_policy.update_default_config_policy(policy_json)
  fc, cfg = q.TFLOperationName.FULLY_CONNECTED, q.OpQuantizationConfig
  ok = cfg(weight_tensor_config=q.TensorQuantizationConfig(num_bits=8, granularity=q.QuantGranularity.TENSORWISE),
           compute_precision=q.ComputePrecision.FLOAT, explicit_dequantize=True, min_weight_elements=77)
  default_policy.check_if_valid_op_config(fc, ok, policy)
  with pytest.raises(ValueError, match="was not found in the policy"):
    default_policy.check_if_valid_op_config(
        fc, cfg(weight_tensor_config=q.TensorQuantizationConfig(num_bits=4), compute_precision=q.ComputePrecision.FLOAT,
                explicit_dequantize=True), policy)
  with pytest.raises(ValueError, match="No policy was specified for op"):
    default_policy.check_if_valid_op_config(q.TFLOperationName.CONV_2D, ok, policy)
  path = tmp_path / "policy.json"
  path.write_text(policy_json)
  try:
    qz.load_config_policy(str(path))
    with pytest.raises(ValueError, match="was not found in the policy"):
      am.check_op_quantization_config("min_max_uniform_quantize", fc, cfg(
          weight_tensor_config=q.TensorQuantizationConfig(num_bits=4), compute_precision=q.ComputePrecision.INTEGER))
  finally:
    am.register_config_check_policy_func("min_max_uniform_quantize", default_policy.DEFAULT_CONFIG_CHECK_POLICY)


# ---- runtime: the registry of mapped model files ---------------------------
def _mapped_weights(tmp_path, name, nbytes):
  from mi355q.utils import tfl_flatbuffer_utils
  path = tmp_path / name
  path.write_bytes(bytes(range(256)) * (nbytes // 256))
  view = tfl_flatbuffer_utils.get_model_content(str(path))
  return path, view


def test_file_mapping_registry_dies_with_the_mapping(tmp_path, monkeypatch):
  """A weight that is a view of a live registered mapping resolves to (descriptor, offset); once
  the mapping is gone no array -- wherever the allocator puts it -- resolves to the dead file."""
  import gc
  from mi355q import runtime as rt
  monkeypatch.setattr(rt, "_UPLOAD_MIN_FILE_BYTES", 1 << 20)
  before = len(rt._FILE_MAPPINGS)
  path, view = _mapped_weights(tmp_path, "a.bin", 8 << 20)
  assert len(rt._FILE_MAPPINGS) == before + 1
  rec = rt._FILE_MAPPINGS[-1]
  w = np.frombuffer(view[4096:4096 + (4 << 20)], dtype=np.float32)
  fd, off = rt._file_range_of(w)
  assert off == 4096 and fd == rec.fd and os.fstat(fd).st_ino == os.stat(path).st_ino
  # the file is replaced under the same path: the descriptor still names the mapped inode
  os.replace(path, str(path) + ".old")
  path.write_bytes(b"\0" * 16)
  assert os.fstat(rt._file_range_of(w)[0]).st_ino == os.stat(str(path) + ".old").st_ino
  # an array that is NOT a view of the mapping but lies inside its address range (simulated: a
  # fake record over a plain array's addresses) is not taken for the file
  plain = np.zeros(1 << 20, np.uint8)
  fake = rt._FileMapping.__new__(rt._FileMapping)
  fake.base, fake.length, fake.fd, fake.ref = plain.ctypes.data, plain.nbytes, rec.fd, rec.ref
  rt._FILE_MAPPINGS.append(fake)
  try:
    assert rt._file_range_of(plain[: 1 << 19]) is None
  finally:
    rt._FILE_MAPPINGS.remove(fake)
  held_fd = rec.fd
  del w, view
  gc.collect()
  assert rec not in rt._FILE_MAPPINGS and rec.fd == -1 and len(rt._FILE_MAPPINGS) == before
  with pytest.raises(OSError):
    os.fstat(held_fd)
  fresh = [np.zeros(4 << 20, np.uint8) for _ in range(4)]
  assert all(rt._file_range_of(a) is None for a in fresh)


# ---- GPTQ statistics with a Hessian on one side only -----------------------------
def test_gptq_update_with_a_hessian_on_one_side_only():
  """A calibration resumed from a result in the reference's layout (every runtime tensor carries a
  Hessian) against this build's default (only read Hessians exist): the merge must not index a
  missing "hessian"; the existing one is kept, the counts add, min / max follow the moving average."""
  from mi355q.utils import qsv_utils
  h = np.eye(4) * 3.0
  with_h = {"min": np.array([[-1.0]], np.float32), "max": np.array([[2.0]], np.float32), "hessian": h, "num_samples": 5}
  without = {"min": np.array([[-3.0]], np.float32), "max": np.array([[1.0]], np.float32), "num_samples": 7}
  for a, b in ((with_h, without), (without, with_h)):
    out = qsv_utils.gptq_and_moving_average_update(dict(a), dict(b))
    assert out["num_samples"] == 12 and out["hessian"] is h
    ema = qsv_utils.moving_average_update(dict(a), dict(b))
    assert np.array_equal(out["min"], ema["min"]) and np.array_equal(out["max"], ema["max"])
  out = qsv_utils.gptq_and_moving_average_update(dict(without), dict(without))
  assert "hessian" not in out and out["num_samples"] == 14


def test_sharded_quantize_refuses_a_gptq_op_whose_hessian_went_elsewhere():
  from mi355q import distributed as D
  w = q.TensorT(name=b"w", shape=[4, 8], buffer=1)
  x = q.TensorT(name=b"x", shape=[1, 8], buffer=0)
  y = q.TensorT(name=b"y", shape=[1, 4], buffer=0)
  buffers = [q.BufferT(), q.BufferT(data=np.zeros(128, np.uint8))]
  gi = q.GraphInfo([x, w, y], buffers)
  op = q.OperatorT(inputs=[0, 1, -1], outputs=[2])
  item = (gi, op, None, q.TFLOperationName.FULLY_CONNECTED, am.AlgorithmName.GPTQ, None)
  ok = {"x": {"min": 0, "max": 1, "num_samples": 3, "hessian": np.eye(8)}}
  D._require_hessians_where_read([item], ok, 1)
  D._require_hessians_where_read([item], {}, 1)                      # uncalibrated: the reference's min / max path
  with pytest.raises(RuntimeError, match="reduced to another rank"):
    D._require_hessians_where_read([item], {"x": {"min": 0, "max": 1, "num_samples": 3}}, 1)
  other = (gi, op, None, q.TFLOperationName.FULLY_CONNECTED, am.AlgorithmName.MIN_MAX_UNIFORM_QUANT, None)
  D._require_hessians_where_read([other], {"x": {"min": 0, "max": 1, "num_samples": 3}}, 1)


# ---- payloads that stay on their rank (sharded runs that write a file) -------------------------
def test_remote_payload_records_instead_of_bytes(tmp_path):
  """Inside runtime.remote_payloads() a device-resident payload pickles to a RemoteBuffer record (its bytes
  stay registered on the owning rank); the serializer's copy_into() of such a record notes the file
  offset it was given instead of copying, and refuses a destination that is no registered output file."""
  import mmap
  import pickle
  import torch
  from mi355q import runtime as rt
  big = rt.HbmArray(torch.arange(1 << 19, dtype=torch.int32).to(torch.int8).reshape(1024, 512))
  big.packed = rt.HbmArray(torch.zeros(1 << 18, dtype=torch.uint8))
  small = rt.HbmArray(torch.zeros(16, dtype=torch.int8))
  with rt.remote_payloads(3):
    got, got_small = pickle.loads(pickle.dumps(big)), pickle.loads(pickle.dumps(small))
  assert isinstance(got, rt.RemoteBuffer) and got.rank == 3 and got.shape == (1024, 512) and got.dtype == np.int8
  assert got.nbytes == 1 << 19 and got.packed.nbytes == 1 << 18 and got.packed.key == got.key + "/packed"
  assert rt._REMOTE_LOCAL[got.key] is big and rt._REMOTE_LOCAL[got.packed.key] is big.packed
  assert not isinstance(got_small, rt.RemoteBuffer)            # small payloads travel as bytes
  with rt.remote_payloads(3):                                  # ... and so do float arrays (blockwise scales are read as values)
    assert not isinstance(pickle.loads(pickle.dumps(rt.HbmArray(torch.zeros(1 << 18, dtype=torch.float32)))), rt.RemoteBuffer)
  assert not isinstance(pickle.loads(pickle.dumps(big)), rt.RemoteBuffer)   # outside the block: host data as before
  with pytest.raises(RuntimeError, match="rank 3"):
    np.asarray(got)
  path = tmp_path / "out.bin"
  fd = os.open(str(path), os.O_RDWR | os.O_CREAT, 0o644)
  os.ftruncate(fd, 1 << 20)
  mapping = mmap.mmap(fd, 1 << 20)
  rt.register_output_mapping(mapping, fd)
  try:
    rt.take_remote_writes()
    got.packed.copy_into(np.frombuffer(mapping, dtype=np.uint8, count=1 << 18, offset=4096))
    ((rank, key, where, offset, nbytes),), _ = rt.take_remote_writes()
    assert (rank, key, offset, nbytes) == (3, got.packed.key, 4096, 1 << 18) and os.path.samefile(where, path)
    # a place that is not inside a registered output file: noted as a host slot, filled when the owner's bytes arrive
    plain = np.zeros(1 << 18, np.uint8)
    got.packed.copy_into(plain)
    ((rank, key, where, slot, nbytes),), slots = rt.take_remote_writes()
    assert (rank, key, where, slot, nbytes) == (3, got.packed.key, None, 0, 1 << 18) and slots[0] is plain
    with pytest.raises(RuntimeError):                          # a size that is not the payload's
      got.packed.copy_into(np.frombuffer(mapping, dtype=np.uint8, count=100, offset=0))
  finally:
    rt.forget_output_mapping(mapping)
    rt._REMOTE_LOCAL.clear()
    mapping.close()
    os.close(fd)


def test_a_hessian_merged_into_a_qsv_without_one_keeps_its_own_sample_count():
  """ref utils/qsv_utils.py:71-102 raises KeyError when only one side has a Hessian; here the Hessian is kept as the mean
  over ITS samples, the QSV's count covers all of them, and a later merge weighs the Hessian by its own count."""
  from mi355q.utils import qsv_utils

  def qsv(n, h=None):
    out = {"min": np.float32(-1), "max": np.float32(1), "num_samples": n}
    if h is not None:
      out["hessian"] = h
    return out
  h1, h2, h3 = (np.eye(2, dtype=np.float32) * v for v in (2.0, 8.0, 5.0))     # (float64 Hessians are merged on the GPU)
  a = qsv_utils.gptq_and_moving_average_update(qsv(3), qsv(2, h1))
  assert a["num_samples"] == 5 and a["hessian_num_samples"] == 2 and np.array_equal(a["hessian"], h1)
  b = qsv_utils.gptq_and_moving_average_update(a, qsv(4, h2))
  assert b["num_samples"] == 9 and b["hessian_num_samples"] == 6
  np.testing.assert_allclose(b["hessian"], (2 * h1 + 4 * h2) / 6)         # not (5 h1 + 4 h2) / 9
  c = qsv_utils.gptq_and_moving_average_update(b, qsv(1))                   # ... and the other way round
  assert c["num_samples"] == 10 and c["hessian_num_samples"] == 6 and np.array_equal(c["hessian"], b["hessian"])
  d = qsv_utils.gptq_and_moving_average_update(qsv(2, h1), qsv(4, h3))      # both sides from the start: nothing extra
  assert d["num_samples"] == 6 and "hessian_num_samples" not in d
  np.testing.assert_allclose(d["hessian"], (2 * h1 + 4 * h3) / 6)


def test_one_array_pickled_twice_is_one_remote_payload_and_compares_equal():
  """A constant two quantized ops read (tied embedding / lm_head): both results hold the SAME array (the (buffer, config)
  cache, ref common_utils.py:48-77), so both records name one payload and qtyping's value comparison
  (params_generator's sharing checks, ref params_generator.py:516-560) answers without the bytes; records of different
  payloads are unequal, never an exception."""
  import pickle
  import torch
  from mi355q import runtime as rt
  shared = rt.HbmArray(torch.zeros((1024, 512), dtype=torch.int8))
  other = rt.HbmArray(torch.zeros((1024, 512), dtype=torch.int8))

  def params(data):
    return q.UniformQuantParams(num_bits=8, quantized_dimension=0, scale=np.ones((1024, 1), np.float32),
                                zero_point=np.zeros((1024, 1), np.int8), symmetric=True, quantized_data=data)
  with rt.remote_payloads(1):
    a, b, c = pickle.loads(pickle.dumps([params(shared), params(shared), params(other)]))
    again = pickle.loads(pickle.dumps(params(shared)))           # a second pickle of the same gather
  rt._REMOTE_LOCAL.clear()
  assert isinstance(a.quantized_data, rt.RemoteBuffer) and a.quantized_data.key == b.quantized_data.key == again.quantized_data.key
  assert a == b and a == again and not (a == c) and a != c
  host = params(np.zeros((1024, 512), np.int8))
  assert not (a == host) and not (host == a)                     # bytes here, a record there: not the same payload


def test_ops_that_read_one_constant_are_planned_onto_one_rank():
  from mi355q import distributed as D
  costs = [(1.0, ("hessian", "a"), 5.0), (2.0, ("hessian", "a"), 5.0), (3.0, None, 0.0), (0.5, ("hessian", "b"), 1.0),
           (0.1, None, 0.0)]
  assert D.plan_op_shards(costs, 2) == [0, 0, 1, 1, 1]
  linked = D.plan_op_shards(costs, 2, [[], [("buffer", 3)], [("buffer", 3)], [], []])
  assert linked[0] == linked[1] == linked[2] and linked[3] != linked[0]
  for world in (3, 8):
    own = D.plan_op_shards(costs, world, [[("buffer", 9)], [], [], [("buffer", 9)], []])
    assert own[0] == own[1] == own[3]                            # chained: Hessian a joins ops 0 and 1, buffer 9 ops 0 and 3
    loads = D.plan_loads(costs, own, world)
    assert abs(sum(loads) - (sum(c[0] for c in costs) + 5.0 + 1.0)) < 1e-9     # each shared part is paid once


def test_late_vectors_are_laid_out_from_their_size_and_filled_last(tmp_path):
  """The external-buffer writer reserves a vector whose values arrive later (runtime.LateVector: per-channel scales
  still in HBM) from its size and dtype, hands every payload its place, calls before_values() and only then reads
  the vector: the file equals the one written with the values known from the start, and nothing reads them before."""
  import torch
  from mi355q import runtime as rt
  from mi355q.utils import tflite_flatbuffer as fb
  model = tfl_flatbuffer_utils.read_model(os.path.join(os.path.dirname(__file__), "golden", "models", "conv_fc_mnist.tflite"))
  sg = model.subgraphs[0]
  scales = np.linspace(0.01, 0.5, 37, dtype=np.float32)
  target = next(t for t in sg.tensors if t.buffer and model.buffers[t.buffer].data is not None and model.buffers[t.buffer].data.nbytes > 4096)
  quant = q.QuantizationParametersT()
  quant.scale = scales
  quant.zeroPoint = np.zeros(37, np.int64)
  target.quantization = quant
  eager = bytes(fb.serialize_with_external_buffers(model, 1024))
  reads = []

  class Watched(rt.HbmArray):
    def numpy(self):
      reads.append("read")
      return super().numpy()
  quant.scale = rt.late_vector(Watched(torch.from_numpy(scales.copy())), np.dtype(np.float32))
  assert isinstance(quant.scale, rt.LateVector) and quant.scale.size == 37 and quant.scale.nbytes == 148
  order = []
  late = bytes(fb.serialize_with_external_buffers(model, 1024, before_values=lambda: order.append(len(reads))))
  assert order == [0] and reads, (order, reads)          # nothing was read before the hook, something after
  assert late == eager
  # the inline writer (no place to fill in later) and plain NumPy consumers read the values on the spot
  assert bytes(fb.write_model(model)) == bytes(fb.write_model(model))
  assert np.array_equal(np.asarray(quant.scale), scales) and quant.scale.tolist() == scales.tolist() and (quant.scale == scales).all()
  # a value that is already on the host, or of another dtype, is an ndarray as before
  assert isinstance(rt.late_vector(scales, np.dtype(np.float32)), np.ndarray)
  assert isinstance(rt.late_vector(rt.HbmArray(torch.zeros(4, dtype=torch.float64)), np.dtype(np.float32)), np.ndarray)


def test_late_constants_share_buffers_like_the_values_first_path(monkeypatch):
  """transformation_utils.get_constant_buffer with constants still in HBM (while a verifying writer is at work): added
  unread when no host buffer has their size; a host constant of that size makes them be read; verify_late_constants
  finds two equal ones (SharingNotDecided) and passes distinct ones -- and outside a writer nothing is late."""
  import torch
  from mi355q import runtime as rt
  from mi355q.transformations import transformation_utils as tu

  def model_with(*host):
    m = q.ModelT(version=3)
    m.buffers = [q.BufferT()] + [q.BufferT(data=np.asarray(h).view(np.uint8)) for h in host]
    return m
  a = rt.HbmArray(torch.arange(512, dtype=torch.float16))
  b = rt.HbmArray(torch.arange(512, dtype=torch.float16) + 1)
  twin = rt.HbmArray(torch.arange(512, dtype=torch.float16))
  # outside a writer: read and compared on the spot (twin shares a's buffer)
  m = model_with(np.zeros(7, np.float32))
  ia, ib, it = (tu.get_constant_buffer(x, m) for x in (a, b, twin))
  assert ia != ib and it == ia and isinstance(m.buffers[ia].data, np.ndarray)
  for x in (a, b, twin):
    x._host = None
  monkeypatch.setattr(rt, "_LATE_CONSTANTS", [1])
  m = model_with(np.zeros(7, np.float32))
  ia, ib = tu.get_constant_buffer(a, m), tu.get_constant_buffer(b, m)
  assert m.buffers[ia].data is a and m.buffers[ib].data is b and a._host is None and b._host is None     # unread
  tu.verify_late_constants(m)                                                                             # distinct: fine
  it = tu.get_constant_buffer(twin, m)
  assert it not in (ia, ib) and m.buffers[it].data is twin
  with pytest.raises(tu.SharingNotDecided):
    tu.verify_late_constants(m)
  # a HOST constant of the same size arrives: the late ones are read now and take part in the comparison
  m = model_with(np.zeros(7, np.float32))
  ia = tu.get_constant_buffer(a, m)
  same_bytes = np.arange(512, dtype=np.float16)
  assert tu.get_constant_buffer(same_bytes, m) == ia
  assert tu.get_constant_buffer(np.arange(512, dtype=np.float16) + 5, m) not in (ia,)
  # a host buffer of that size already in the model: the device constant is read at once
  for x in (a, b, twin):
    x._host = None
  m = model_with(np.arange(512, dtype=np.float16))
  assert tu.get_constant_buffer(twin, m) == 1
