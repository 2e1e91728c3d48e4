"""INTEGRATION.md section 3 says the reference's OWN objects can be handed to this package's functions (they only use
attribute access on `op_info` / `tensor_quant_config`, and the enums compare by value). This makes that a tested
statement for the host-only entries -- the ones that run without a GPU: `tensor_zp_scale_from_min_max` (a2), the
activation branch of `get_tensor_quant_params` (a4, tensor_content None), the QSV merges (a9) and the recipe manager.

Build container only: the real reference is imported from /root/reference through the committed stand-ins for its
absent third-party packages (tests/golden/gen/shim, the same bootstrap tests/golden/gen/make_golden.py uses); skipped
where /root/reference does not exist (the GPU box).
Ref: algorithm_manager_api.py:191-227, qtyping.py:205-313, 384-445, 567-581, 701-710."""
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/ai_edge_quantizer"
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")


@pytest.fixture(scope="module")
def ref():
  """The reference's modules, imported unmodified (never written to: no bytecode)."""
  saved_flag, saved_path = sys.dont_write_bytecode, list(sys.path)
  sys.dont_write_bytecode = True
  sys.path.insert(0, os.path.join(HERE, "golden", "gen", "shim"))
  had = {k: v for k, v in sys.modules.items() if k == "ai_edge_quantizer" or k.startswith("ai_edge_quantizer.")}
  pkg = types.ModuleType("ai_edge_quantizer")
  pkg.__path__ = [REF]
  sys.modules["ai_edge_quantizer"] = pkg
  try:
    from ai_edge_quantizer import qtyping, recipe_manager
    from ai_edge_quantizer.algorithms.uniform_quantize import naive_min_max_quantize as mm
    from ai_edge_quantizer.algorithms.uniform_quantize import uniform_quantize_tensor as uqt
    from ai_edge_quantizer.utils import qsv_utils
    import ml_dtypes   # noqa: F401  (the stand-in, while the shim directory leads sys.path)
    yield types.SimpleNamespace(qtyping=qtyping, uqt=uqt, mm=mm, qsv_utils=qsv_utils, recipe_manager=recipe_manager)
  finally:
    for k in [k for k in sys.modules if k == "ai_edge_quantizer" or k.startswith("ai_edge_quantizer.")]:
      del sys.modules[k]
    sys.modules.update(had)
    sys.path[:] = saved_path
    sys.dont_write_bytecode = saved_flag


def _same(a, b):
  a, b = np.asarray(a), np.asarray(b)
  return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("gran", ["TENSORWISE", "CHANNELWISE", "BLOCKWISE_32", "BLOCKWISE_128"])
@pytest.mark.parametrize("bits,symmetric", [(8, True), (8, False), (4, True), (4, False), (16, True)])
def test_zp_scale_from_min_max_takes_the_references_granularity_enum(ref, gran, bits, symmetric):
  from mi355q.algorithms.uniform_quantize import uniform_quantize_tensor as ours
  rng = np.random.default_rng(bits * 7 + len(gran))
  shape = (6, 5) if gran.startswith("BLOCKWISE") else (6, 1) if gran == "CHANNELWISE" else (1, 1)
  lo = -np.abs(rng.standard_normal(shape)).astype(np.float32) * 3
  hi = np.abs(rng.standard_normal(shape)).astype(np.float32) * 3
  lo[0, 0], hi[0, 0] = 0.0, 0.0                         # the 1e-9 floor
  g_ref = ref.qtyping.QuantGranularity[gran]
  import ml_dtypes     # the stand-in: blockwise scales go through `.astype(ml_dtypes.bfloat16)` in the reference, which
  aware = (lambda a: a.view(ml_dtypes.Bf16Aware)) if gran.startswith("BLOCKWISE") else (lambda a: a)   # this subclass resolves
  for clip in (None, np.full(shape, 1.5, np.float32)):
    want_zp, want_scale = ref.uqt.tensor_zp_scale_from_min_max(aware(lo), aware(hi), bits, symmetric, g_ref, clip)
    want_zp, want_scale = np.asarray(want_zp).view(np.ndarray), np.asarray(want_scale).view(np.ndarray)
    got_zp, got_scale = ours.tensor_zp_scale_from_min_max(lo, hi, bits, symmetric, g_ref, clip)      # THEIR enum
    assert _same(got_scale, want_scale) and _same(got_zp, want_zp), (gran, bits, symmetric, clip is None)


@pytest.mark.parametrize("bits,symmetric", [(8, False), (8, True), (16, True)])
def test_activation_quant_params_take_the_references_op_info_and_config(ref, bits, symmetric):
  """The branch with no tensor content (activations: parameters from calibrated min / max) is host arithmetic."""
  from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as ours
  Q = ref.qtyping
  cfg = Q.TensorQuantizationConfig(num_bits=bits, symmetric=symmetric, granularity=Q.QuantGranularity.TENSORWISE)
  info = Q.OpInfo(op=Q.OperatorT(), op_name=Q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                  op_quant_config=Q.OpQuantizationConfig(activation_tensor_config=cfg, weight_tensor_config=cfg,
                                                         compute_precision=Q.ComputePrecision.INTEGER))
  qsv = {"min": np.array([[-1.25]], np.float32), "max": np.array([[3.5]], np.float32)}
  want = ref.mm.get_tensor_quant_params(info, cfg, None, qsv)
  got = ours.get_tensor_quant_params(info, cfg, None, qsv)                                           # THEIR objects
  assert _same(got.scale, want.scale) and _same(got.zero_point, want.zero_point)
  assert (got.num_bits, got.symmetric, got.quantized_dimension, got.block_size) == (
      want.num_bits, want.symmetric, want.quantized_dimension, want.block_size)
  assert got.quantized_data is None and want.quantized_data is None
  with pytest.raises(ValueError) as theirs:
    ref.mm.get_tensor_quant_params(info, cfg, None, {"min": qsv["min"]})
  with pytest.raises(ValueError) as mine:
    ours.get_tensor_quant_params(info, cfg, None, {"min": qsv["min"]})
  assert str(mine.value) == str(theirs.value)


def test_qsv_merges_equal_the_references_on_the_same_records(ref):
  from mi355q.utils import qsv_utils as ours
  rng = np.random.default_rng(3)
  for _ in range(50):
    a = {"min": rng.standard_normal((1, 1)).astype(np.float32), "max": rng.standard_normal((1, 1)).astype(np.float32)}
    b = {"min": rng.standard_normal((1, 1)).astype(np.float32), "max": rng.standard_normal((1, 1)).astype(np.float32)}
    for name in ("moving_average_update", "min_max_update"):
      want, got = getattr(ref.qsv_utils, name)(dict(a), dict(b)), getattr(ours, name)(dict(a), dict(b))
      assert set(got) == set(want) and all(_same(got[k], want[k]) for k in want), name


def test_recipe_manager_takes_the_references_names_and_configs(ref):
  """Scopes added with THEIR op names / configs / algorithm keys resolve like THEIR manager resolves them."""
  from mi355q import recipe_manager as ours
  Q = ref.qtyping
  w8 = Q.OpQuantizationConfig(weight_tensor_config=Q.TensorQuantizationConfig(num_bits=8, symmetric=True,
                                                                              granularity=Q.QuantGranularity.CHANNELWISE),
                              compute_precision=Q.ComputePrecision.INTEGER)
  w4 = Q.OpQuantizationConfig(weight_tensor_config=Q.TensorQuantizationConfig(num_bits=4, symmetric=True,
                                                                              granularity=Q.QuantGranularity.BLOCKWISE_32),
                              compute_precision=Q.ComputePrecision.INTEGER)
  mine, theirs = ours.RecipeManager(), ref.recipe_manager.RecipeManager()
  for rm in (mine, theirs):
    rm.add_quantization_config(".*", Q.TFLOperationName.FULLY_CONNECTED, op_config=w8)
    rm.add_quantization_config(".*/attn/.*", Q.TFLOperationName.FULLY_CONNECTED, op_config=w4)
  for scope in ("model/mlp/fc;", "model/attn/q;", "other;"):
    for op in (Q.TFLOperationName.FULLY_CONNECTED, Q.TFLOperationName.CONV_2D):
      alg_m, cfg_m = mine.get_quantization_configs(op, scope)
      alg_t, cfg_t = theirs.get_quantization_configs(op, scope)
      assert str(getattr(alg_m, "value", alg_m)) == str(getattr(alg_t, "value", alg_t)), (scope, op)
      wm, wt = cfg_m.weight_tensor_config, cfg_t.weight_tensor_config
      if wt is None:                          # nothing matched: NO_QUANTIZE with an empty config on both sides
        assert wm is None and str(getattr(alg_t, "value", alg_t)) == "no_quantize"
        continue
      assert (wm.num_bits, wm.symmetric, str(wm.granularity.value)) == (wt.num_bits, wt.symmetric, str(wt.granularity.value))
  # ... and the recipe THEIR manager dumps loads into OURS (the JSON form is the contract between the two)
  again = ours.RecipeManager()
  again.load_quantization_recipe(theirs.get_quantization_recipe())
  assert again.get_quantization_recipe() == theirs.get_quantization_recipe() == mine.get_quantization_recipe()
