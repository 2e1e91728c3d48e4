"""Records what the REAL reference does to whole `.tflite` models (file-level goldens).

Runs in the build container only (needs /root/reference). The reference's own
`ModelModifier.modify_model` (ParamsGenerator -> transformation instructions ->
TransformationPerformer, reference processing order) is executed on models parsed from the
reference's test `.tflite` files; the only part that cannot run here is the final flatbuffer
serialization (third-party wheel), which is replaced by "hand back the quantized ModelT".
The files are parsed with this repository's reader (mi355q.utils.tflite_flatbuffer) and handed
to the reference as attribute-bag objects, so reader bugs would show up as reference failures.

Outputs (committed):
  tests/golden/models/*.tflite        copies of the reference's test-model DATA files
  tests/golden/ref_model_cases.json   per model x recipe: every tensor's type / shape / buffer /
                                      quantization record and the SHA-256 of every buffer

usage: python tests/golden/gen/make_model_golden.py
"""
import hashlib
import json
import os
import shutil
import sys
sys.dont_write_bytecode = True  # never leave .pyc files in the read-only reference tree
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
ROOT = os.path.dirname(os.path.dirname(GOLDEN))
REF = "/root/reference/ai_edge_quantizer"

sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
pkg = types.ModuleType("ai_edge_quantizer")
pkg.__path__ = [REF]
sys.modules["ai_edge_quantizer"] = pkg

from mi355q import schema as our_schema  # noqa: E402
from mi355q.utils import tflite_flatbuffer as fb  # noqa: E402
from ai_edge_litert.tools import flatbuffer_utils as shim_fu  # noqa: E402

# real BuiltinOperator codes for every op the reference's tables mention
object.__getattribute__(shim_fu.BuiltinOperator, "_codes").update(
    {m.name: int(m) for m in our_schema.BuiltinOperator})

import ml_dtypes  # the shim  # noqa: E402
from ai_edge_quantizer import model_modifier, params_generator, qtyping, recipe, recipe_manager  # noqa: E402
from ai_edge_quantizer.utils import tfl_flatbuffer_utils as fbu  # noqa: E402

MODELS = [
    "single_fc", "single_fc_bias", "single_fc_no_bias", "embedding_lookup", "weight_sharing_fcs",
    "constant_tensor_and_buffer_only_sharing_weight_fcs", "conv_fc_mnist", "branching_conv_fc",
    "single_depthwise_conv2d_bias", "single_conv2d_transpose_bias", "bmm_constant_input",
    "two_signatures", "toy_model_with_kv_cache_multi_signature", "single_fc_bias_sub_channel_weight_only_sym_weight",
    "simple_composite", "reshape_with_empty_shape",
]
RECIPES = {
    "dynamic_wi8_afp32": recipe.dynamic_wi8_afp32(),
    "dynamic_wi4_afp32": recipe.dynamic_wi4_afp32(),
    "dynamic_wi8_octav": recipe.dynamic_wi8_afp32(algorithm_key="OCTAV"),
    "dynamic_wi4b32_afp32": recipe.dynamic_wi4b32_afp32(),
    "dynamic_legacy_wi8_afp32": json.load(open(os.path.join(REF, "recipes/dynamic_legacy_wi8_afp32_recipe.json"))),
    "weight_only_wi8_afp32": recipe.weight_only_wi8_afp32(),
    "weight_only_wi4_afp32": recipe.weight_only_wi4_afp32(),
    "default_af32w8float": json.load(open(os.path.join(REF, "recipes/default_af32w8float_recipe.json"))),
    "default_af32w4float": json.load(open(os.path.join(REF, "recipes/default_af32w4float_recipe.json"))),
    "dynamic_wi2c_afp32": recipe.dynamic_wi2c_afp32(),
    "dynamic_wi8_mse": recipe.dynamic_wi8_afp32(algorithm_key="MSE"),
    "dynamic_wi4_octav": recipe.dynamic_wi4_afp32(algorithm_key="OCTAV"),
    "dynamic_wi8_tensorwise": recipe.dynamic_wi8c_afp32(granularity=qtyping.QuantGranularity.TENSORWISE),
}

_SHIM_CLASS = {
    "Model": shim_fu.ModelT, "SubGraph": shim_fu.SubGraphT, "Tensor": shim_fu.TensorT,
    "Buffer": shim_fu.BufferT, "Operator": shim_fu.OperatorT, "OperatorCode": shim_fu.OperatorCodeT,
    "QuantizationParameters": shim_fu.QuantizationParametersT,
    "BlockwiseQuantization": shim_fu.BlockwiseQuantizationT,
    "FullyConnectedOptions": shim_fu.FullyConnectedOptionsT,
    "StableHLOCompositeOptions": shim_fu.StableHLOCompositeOptionsT,
}


def to_bags(v):
  """Our parsed tree -> the attribute bags the shimmed reference works on."""
  if isinstance(v, fb.TableT):
    cls = _SHIM_CLASS.get(v._table, shim_fu._Bag)
    out = cls()
    for spec in fb.SCHEMA[v._table]:
      if spec[1] != "dead":
        setattr(out, spec[0], to_bags(getattr(v, spec[0])))
    return out
  if isinstance(v, list):
    return [to_bags(e) for e in v]
  return v


def sha(b) -> str:
  return hashlib.sha256(bytes(b)).hexdigest()


def describe(model, raw=False):
  import base64

  def buf(b):
    if b.data is None:
      return None
    flat = np.ravel(np.asarray(b.data)).view(np.uint8)
    rec = dict(nbytes=int(flat.nbytes), sha256=sha(flat))
    if raw and flat.nbytes <= 65536:      # rotated weights are compared with a tolerance
      rec["b64"] = base64.b64encode(flat.tobytes()).decode()
    return rec
  out = dict(n_buffers=len(model.buffers), subgraphs=[], buffers=[buf(b) for b in model.buffers])
  for sg in model.subgraphs:
    tensors = []
    for t in sg.tensors:
      rec = dict(name=t.name.decode() if isinstance(t.name, (bytes, bytearray)) else str(t.name), type=int(t.type), shape=None if t.shape is None else [int(s) for s in t.shape],
                 buffer=int(t.buffer))
      q = t.quantization
      if q is not None:
        qr = dict(quantized_dimension=int(q.quantizedDimension), details_type=int(q.detailsType))
        for k in ("scale", "zeroPoint", "min", "max"):
          a = getattr(q, k, None)
          if a is not None:
            a = np.asarray(a)
            want = {"scale": np.float32, "min": np.float32, "max": np.float32, "zeroPoint": np.int64}[k]
            a = a.astype(want)
            qr[k] = dict(n=int(a.size), sha256=sha(a.tobytes()), head=[float(x) for x in a.ravel()[:4]])
        if q.details is not None and int(q.detailsType) == 2:
          qr["blockwise"] = dict(scales=int(q.details.scales), zero_points=int(q.details.zeroPoints),
                                 block_size=int(q.details.blockSize))
        rec["quantization"] = qr
      tensors.append(rec)
    ops = [[int(model.operatorCodes[op.opcodeIndex].builtinCode), [int(i) for i in op.inputs],
            [int(i) for i in op.outputs]] for op in (sg.operators or [])]
    out["subgraphs"].append(dict(tensors=tensors, n_operators=len(sg.operators or []), operators=ops,
                                 inputs=[int(i) for i in sg.inputs], outputs=[int(i) for i in sg.outputs]))
  out["signatures"] = [[int(sig.subgraphIndex), [int(i.tensorIndex) for i in (sig.inputs or [])],
                        [int(i.tensorIndex) for i in (sig.outputs or [])]] for sig in (model.signatureDefs or [])]
  return out


def run(model_name, recipe_name, rcp, qsvs=None, path=None):
  path = path or os.path.join(REF, "tests/models", model_name + ".tflite")
  model = to_bags(fb.read_model(open(path, "rb").read()))
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(rcp)
  if rm.need_calibration() and qsvs is None:
    return None
  orig = fbu.get_tensor_data
  # blockwise scales pass through `.astype(ml_dtypes.bfloat16)`: hand weights over as the
  # bf16-aware ndarray subclass (see shim/ml_dtypes.py)
  fbu.get_tensor_data = lambda t, b, _o=orig: (
      None if _o(t, b) is None else (_o(t, b).view(ml_dtypes.Bf16Aware) if _o(t, b).dtype == np.float32 else _o(t, b)))
  try:
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm, qsvs)
      mod = model_modifier.ModelModifier(model)
      mod._serialize_small_model = lambda m: m
      mod._serialize_model = lambda m, packed, serialize_to_path=None: m
      quantized = mod.modify_model(params)
  finally:
    fbu.get_tensor_data = orig
  return describe(quantized, raw="hadamard" in recipe_name)


HADAMARD_RECIPES = {
    "dynamic_wi4_afp32_hadamard": json.load(open(os.path.join(REF, "recipes/dynamic_wi4_afp32_hadamard_recipe.json"))),
    "dynamic_wi8_afp32_hadamard": json.load(open(os.path.join(REF, "recipes/dynamic_wi8_afp32_hadamard_recipe.json"))),
}
HADAMARD_MODELS = ["single_fc", "single_fc_bias", "embedding_lookup", "weight_sharing_fcs", "conv_fc_mnist",
                   "branching_conv_fc", "constant_tensor_and_buffer_only_sharing_weight_fcs"]


def policy_table():
  """The reference's config-check policy as data: op -> accepted config rows."""
  from ai_edge_quantizer import default_policy as dp
  v = lambda x: getattr(x, "value", x)  # noqa: E731
  out = {}
  for op, cfgs in dp.DEFAULT_CONFIG_CHECK_POLICY.items():
    rows = set()
    for c in cfgs:
      a, w = c.activation_tensor_config, c.weight_tensor_config
      rows.add((None if a is None else a.num_bits, None if a is None else a.symmetric,
                None if a is None else v(a.granularity), w.num_bits, w.symmetric, v(w.granularity),
                v(c.compute_precision), c.explicit_dequantize))
    out[v(op)] = sorted(rows, key=str)
  with open(os.path.join(GOLDEN, "ref_policy.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py (reference default_policy."
                             "DEFAULT_CONFIG_CHECK_POLICY)",
                   row="[act_bits, act_sym, act_gran, w_bits, w_sym, w_gran, compute_precision,"
                       " explicit_dequantize]", policy=out), f, indent=0, sort_keys=True)
  print("policy:", len(out), "ops,", sum(len(r) for r in out.values()), "rows")


def main():
  policy_table()
  cases = {}
  for name in MODELS:
    shutil.copyfile(os.path.join(REF, "tests/models", name + ".tflite"),
                    os.path.join(GOLDEN, "models", name + ".tflite"))
    for rname, rcp in RECIPES.items():
      key = f"{name}/{rname}"
      try:
        res = run(name, rname, rcp)
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, result=res)
        print("ok  ", key)
      except Exception as e:  # the reference itself rejects this combination
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, error=type(e).__name__,
                          message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
  for name in HADAMARD_MODELS:
    for rname, rcp in HADAMARD_RECIPES.items():
      key = f"{name}/{rname}"
      try:
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, result=run(name, rname, rcp))
        print("ok  ", key)
      except Exception as e:
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
  with open(os.path.join(GOLDEN, "ref_model_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py",
                   reference_version=open("/root/reference/VERSION").read().strip(),
                   numpy=np.__version__, cases=json.loads(json.dumps(cases, default=str))),
              f, separators=(",", ":"), sort_keys=True)
  print("wrote", len(cases), "cases")


if __name__ == "__main__" and not {"--srq", "--insts", "--oscar", "--srq-all", "--more", "--mixed", "--fp16", "--dwr", "--random", "--random-mixed"} & set(sys.argv):
  main()


# ---------------------------------------------------------------------------------------
# static (SRQ) recipes: what ParamsGenerator hands the transformation layer for every tensor
# ---------------------------------------------------------------------------------------
SRQ_MODELS = ["single_fc_bias", "single_fc_no_bias", "single_fc", "single_depthwise_conv2d_bias",
              "single_conv2d_transpose_bias", "single_fc_bias_relu"]


def _params_summary(p):
  if p is None:
    return None
  out = dict(kind=type(p).__name__, num_bits=int(p.num_bits))
  if hasattr(p, "scale"):
    sc = np.asarray(p.scale)
    zp = None if p.zero_point is None else np.asarray(p.zero_point)
    out.update(symmetric=bool(p.symmetric), quantized_dimension=p.quantized_dimension,
               block_size=int(getattr(p, "block_size", 0) or 0),
               scale=dict(shape=list(sc.shape), dtype=str(sc.dtype), sha256=sha(np.ascontiguousarray(sc).tobytes()),
                          head=[float(x) for x in sc.ravel()[:3]]),
               zero_point=None if zp is None else dict(shape=list(zp.shape), dtype=str(zp.dtype),
                                                       sha256=sha(np.ascontiguousarray(zp).tobytes()),
                                                       head=[int(x) for x in zp.ravel()[:3]]))
    qd = p.quantized_data
    out["quantized_data"] = None if qd is None else dict(
        shape=list(np.shape(qd)), dtype=str(np.asarray(qd).dtype), sha256=sha(np.ascontiguousarray(qd).tobytes()))
  return out


def srq_cases():
  from ai_edge_quantizer import recipe as ref_recipe
  out = {}
  for name in SRQ_MODELS:
    for rname, rcp in (("static_wi8_ai8", ref_recipe.static_wi8_ai8()), ("static_wi8_ai16", ref_recipe.static_wi8_ai16())):
      path = os.path.join(REF, "tests/models", name + ".tflite")
      shutil.copyfile(path, os.path.join(GOLDEN, "models", name + ".tflite"))
      model = to_bags(fb.read_model(open(path, "rb").read()))
      # synthetic calibration result: an (asymmetric) range for every non-constant tensor
      rng = np.random.default_rng(len(name) * 7 + len(rname))
      qsvs = {}
      for sg in model.subgraphs:
        for t in sg.tensors:
          if model.buffers[t.buffer].data is None:
            lo, hi = sorted(rng.uniform(-6, 6, 2))
            qsvs[t.name.decode()] = {"min": np.array([[min(lo, -0.1)]], np.float32),
                                     "max": np.array([[max(hi, 0.1)]], np.float32)}
      rm = recipe_manager.RecipeManager()
      rm.load_quantization_recipe(rcp)
      key = f"{name}/{rname}"
      try:
        with warnings.catch_warnings():
          warnings.simplefilter("ignore")
          params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm, qsvs)
      except Exception as e:
        out[key] = dict(model=name, recipe=rcp, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:100])
        continue
      rec = {}
      for tname, tp in params.items():
        links = []
        for role, link_list in (("producer", [tp.producer] if tp.producer is not None else []),
                                ("consumer", tp.consumers or [])):
          for link in link_list:
            links.append(dict(role=role, op=int(link.subgraph_op_id),
                              transformations=[t.name for t in link.transformations],
                              parameters=_params_summary(link.parameters)))
        rec[tname] = links
      out[key] = dict(model=name, recipe=rcp, qsvs={k: {"min": float(v["min"].ravel()[0]), "max": float(v["max"].ravel()[0])}
                                                     for k, v in qsvs.items()}, params=rec,
                      result=run(name, rname, rcp, qsvs))
      print("ok  ", key)
  with open(os.path.join(GOLDEN, "ref_srq_params.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py (srq_cases)", numpy=np.__version__,
                   cases=json.loads(json.dumps(out, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--srq" in sys.argv:
  srq_cases()


# ---------------------------------------------------------------------------------------
# instruction generator known answers: synthetic per-tensor parameters -> instruction lists
# ---------------------------------------------------------------------------------------
def instruction_cases():
  import itertools
  from ai_edge_quantizer import transformation_instruction_generator as tig
  T = qtyping.QuantTransformation
  path = os.path.join(REF, "tests/models", "branching_conv_fc.tflite")
  model = to_bags(fb.read_model(open(path, "rb").read()))
  gen = tig.TransformationInstructionsGenerator()
  gen.flatbuffer_model = model
  gen._create_tensor_name_to_graph_info_map()
  # a tensor with several consumers
  sg = model.subgraphs[0]
  info = max(gen._tensor_name_to_graph_info.items(), key=lambda kv: len(kv[1].consumers))
  name, gi = info
  consumers = list(gi.consumers)

  def qp(k):
    return qtyping.UniformQuantParams(num_bits=8, quantized_dimension=None,
                                      scale=np.array([0.1 * (k + 1)], np.float32),
                                      zero_point=np.array([k], np.int64), symmetric=False)
  P = [qp(0), qp(1)]
  consumer_options = [([T.ADD_QUANTIZE], 0), ([T.ADD_QUANTIZE], 1), ([T.NO_QUANTIZE], None),
                      ([T.ADD_QUANTIZE, T.ADD_DEQUANTIZE], 0)]
  producer_options = [None, ([T.ADD_DEQUANTIZE], 0), ([T.NO_QUANTIZE], None), ([T.ADD_DEQUANTIZE], 1)]
  out = []
  for prod in producer_options:
    for combo in itertools.product(range(len(consumer_options)), repeat=len(consumers)):
      def link(op_id, opt):
        tr, k = opt
        return qtyping.OpToTensorParams(subgraph_op_id=op_id, transformations=list(tr),
                                        parameters=None if k is None else P[k])
      param = qtyping.TensorTransformationParams(
          tensor_name=name,
          producer=None if prod is None else link(gi.producer, prod),
          consumers=[link(c, consumer_options[i]) for c, i in zip(consumers, combo)])
      # graph info lists are mutated by the generator: rebuild per case
      gen._create_tensor_name_to_graph_info_map()
      rec = dict(producer=None if prod is None else [[t.name for t in prod[0]], prod[1]],
                 consumers=[[[t.name for t in consumer_options[i][0]], consumer_options[i][1]] for i in combo])
      try:
        res = gen._quant_params_to_transformation_insts(param)
        rec["instructions"] = [[i.transformation.name, int(i.tensor_id), int(i.producer), [int(c) for c in i.consumers],
                                None if i.parameters is None else int(np.asarray(i.parameters.zero_point).ravel()[0])]
                               for i in res.instructions]
      except Exception as e:
        rec["error"] = type(e).__name__
      out.append(rec)
  with open(os.path.join(GOLDEN, "ref_instruction_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py (instruction_cases)",
                   model="branching_conv_fc", tensor=name, op_ids=[int(c) for c in consumers],
                   producer_op=int(gi.producer), cases=out), f, separators=(",", ":"), sort_keys=True)
  print("instruction cases:", len(out), "tensor", name, consumers, gi.producer)


if __name__ == "__main__" and "--insts" in sys.argv:
  instruction_cases()


# ---------------------------------------------------------------- OSCAR, model level ---
OSCAR_MODELS = ["single_fc", "single_fc_bias", "weight_sharing_fcs", "conv_fc_mnist", "branching_conv_fc",
                "toy_model_with_kv_cache_multi_signature"]


def oscar_recipe(bits, gran, precision="INTEGER", explicit=False):
  return [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="OSCAR", op_config=dict(
      weight_tensor_config=dict(num_bits=bits, symmetric=True, granularity=gran, dtype="INT"),
      compute_precision=precision, explicit_dequantize=explicit, skip_checks=False,
      min_weight_elements=0))]


def oscar_qsvs(model, seed):
  """Synthetic calibration result: a range + a second-moment vector over the trailing axis for
  every runtime tensor (regenerated from `seed` by the test)."""
  rng = np.random.default_rng(seed)
  qsvs = {}
  for sg in model.subgraphs:
    for t in sg.tensors:
      if model.buffers[t.buffer].data is None and len(t.shape):
        lo, hi = sorted(rng.uniform(-6, 6, 2))
        mu2 = np.exp(rng.normal(size=int(t.shape[-1])) * 1.5)
        qsvs[t.name.decode()] = {"min": np.array([[min(lo, -0.1)]], np.float32),
                                 "max": np.array([[max(hi, 0.1)]], np.float32),
                                 "mu2": mu2, "num_samples": 7}
  return qsvs


def oscar_model_cases():
  from ai_edge_quantizer.algorithms.uniform_quantize import oscar
  real = oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max
  oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max = lambda mn, mx, *a, **k: real(
      np.asarray(mn).view(ml_dtypes.Bf16Aware), np.asarray(mx).view(ml_dtypes.Bf16Aware), *a, **k)
  out = {}
  try:
    for i, name in enumerate(OSCAR_MODELS):
      shutil.copyfile(os.path.join(REF, "tests/models", name + ".tflite"),
                      os.path.join(GOLDEN, "models", name + ".tflite"))
      model = to_bags(fb.read_model(open(os.path.join(REF, "tests/models", name + ".tflite"), "rb").read()))
      for rname, rcp in (("oscar_wi4_cw", oscar_recipe(4, "CHANNELWISE")),
                         ("oscar_wi4_b32", oscar_recipe(4, "BLOCKWISE_32")),
                         ("oscar_wi8_cw_weight_only", oscar_recipe(8, "CHANNELWISE", "FLOAT", True))):
        for with_mu2 in (True, False):
          seed = 4000 + i
          qsvs = oscar_qsvs(model, seed) if with_mu2 else {}
          key = f"{name}/{rname}/{'mu2' if with_mu2 else 'nomu2'}"
          try:
            res = run(name, rname, rcp, qsvs)
          except Exception as e:
            out[key] = dict(model=name, recipe=rcp, seed=seed, with_mu2=with_mu2, error=type(e).__name__,
                            message=str(e)[:300])
            print("err ", key, type(e).__name__, str(e)[:100])
            continue
          out[key] = dict(model=name, recipe=rcp, seed=seed, with_mu2=with_mu2, result=res)
          print("ok  ", key)
  finally:
    oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max = real
  with open(os.path.join(GOLDEN, "ref_oscar_model_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --oscar", numpy=np.__version__,
                   cases=out), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--oscar" in sys.argv:
  oscar_model_cases()


# ------------------------------------------------ static recipes on EVERY reference test model ---
def srq_all_cases(names=None, recipes=None, out_name="ref_srq_all_params.json", skip_covered=True):
  """static_wi8_ai8 / static_wi8_ai16 over all of the reference's test models (every op kind the
  registry knows: scale constraints, fixed output scales, ignored operands), with a synthetic
  calibration result. Records per-tensor parameters and the quantized model (or the error)."""
  import zlib
  from ai_edge_quantizer import recipe as ref_recipe
  out = {}
  names = names or sorted(f[:-7] for f in os.listdir(os.path.join(REF, "tests/models")) if f.endswith(".tflite"))
  recipes = recipes or (("static_wi8_ai8", ref_recipe.static_wi8_ai8()), ("static_wi8_ai16", ref_recipe.static_wi8_ai16()))
  for name in names:
    if skip_covered and name in SRQ_MODELS:
      continue                                   # already in ref_srq_params.json
    path = os.path.join(REF, "tests/models", name + ".tflite")
    shutil.copyfile(path, os.path.join(GOLDEN, "models", name + ".tflite"))
    for rname, rcp in recipes:
      model = to_bags(fb.read_model(open(path, "rb").read()))
      rng = np.random.default_rng(zlib.crc32(f"{name}/{rname}".encode()))
      qsvs = {}
      for sg in model.subgraphs:
        for t in sg.tensors:
          if model.buffers[t.buffer].data is None:
            lo, hi = sorted(rng.uniform(-6, 6, 2))
            qsvs[t.name.decode()] = {"min": np.array([[min(lo, -0.1)]], np.float32),
                                     "max": np.array([[max(hi, 0.1)]], np.float32)}
      seed_qsvs = {k: {"min": float(v["min"].ravel()[0]), "max": float(v["max"].ravel()[0])} for k, v in qsvs.items()}
      key = f"{name}/{rname}"
      rm = recipe_manager.RecipeManager()
      rm.load_quantization_recipe(rcp)
      try:
        with warnings.catch_warnings():
          warnings.simplefilter("ignore")
          params = params_generator.ParamsGenerator(model).generate_quantization_parameters(rm, dict(qsvs))
          rec = {}
          for tname, tp in params.items():
            links = []
            for role, link_list in (("producer", [tp.producer] if tp.producer is not None else []),
                                    ("consumer", tp.consumers or [])):
              for link in link_list:
                links.append(dict(role=role, op=int(link.subgraph_op_id),
                                  transformations=[t.name for t in link.transformations],
                                  parameters=_params_summary(link.parameters)))
            rec[tname] = links
          result = run(name, rname, rcp, dict(qsvs))
      except Exception as e:
        out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
        continue
      out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs, params=rec, result=result)
      print("ok  ", key)
  with open(os.path.join(GOLDEN, out_name), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --srq-all / --mixed", numpy=np.__version__,
                   cases=json.loads(json.dumps(out, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--srq-all" in sys.argv:
  srq_all_cases()


# ------------------------------------ dynamic / weight-only recipes on the remaining test models ---
def more_model_cases():
  """The models ref_model_cases.json does not cover (single-op models of every registered op,
  already-quantized models, composites) under four dynamic-range / weight-only recipes."""
  names = sorted(f[:-7] for f in os.listdir(os.path.join(REF, "tests/models")) if f.endswith(".tflite"))
  recipes = {k: RECIPES[k] for k in ("dynamic_wi8_afp32", "dynamic_wi4_afp32", "weight_only_wi8_afp32",
                                     "default_af32w8float")}
  cases = {}
  for name in names:
    if name in MODELS or name == "toy_model_with_kv_cache_multi_signature":
      continue
    shutil.copyfile(os.path.join(REF, "tests/models", name + ".tflite"),
                    os.path.join(GOLDEN, "models", name + ".tflite"))
    for rname, rcp in recipes.items():
      key = f"{name}/{rname}"
      try:
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, result=run(name, rname, rcp))
        print("ok  ", key)
      except Exception as e:
        cases[key] = dict(model=name, recipe_name=rname, recipe=rcp, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
  with open(os.path.join(GOLDEN, "ref_model_cases_more.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --more", numpy=np.__version__,
                   cases=json.loads(json.dumps(cases, default=str))), f, separators=(",", ":"), sort_keys=True)
  print("wrote", len(cases), "cases")


if __name__ == "__main__" and "--more" in sys.argv:
  more_model_cases()


def mixed_cases():
  """Recipes that mix modes per op: the shipped sample_advanced_usage recipe (static int8
  everywhere, weight-only int4 FULLY_CONNECTED, CONV_2D left alone) and static int8 with OCTAV
  weights / int4 weights for FULLY_CONNECTED."""
  adv = json.load(open(os.path.join(REF, "recipes/sample_advanced_usage_recipe.json")))
  from ai_edge_quantizer import recipe as ref_recipe
  octav = json.loads(json.dumps(ref_recipe.static_wi8_ai8()))
  for e in octav:
    e["algorithm_key"] = "OCTAV"
  w4 = json.loads(json.dumps(ref_recipe.static_wi8_ai8())) + [dict(
      regex=".*", operation="FULLY_CONNECTED", algorithm_key="min_max_uniform_quantize", op_config=dict(
          activation_tensor_config=dict(num_bits=8, symmetric=False, granularity="TENSORWISE", dtype="INT"),
          weight_tensor_config=dict(num_bits=4, symmetric=True, granularity="CHANNELWISE", dtype="INT"),
          compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False, min_weight_elements=0))]
  srq_all_cases(names=["conv_fc_mnist", "branching_conv_fc", "toy_model_with_kv_cache_multi_signature",
                       "weight_sharing_fcs", "single_fc_bias_logistic", "sdpa_composite", "two_signatures",
                       "single_fc_bias", "partly_quantized_mnist"],
                recipes=(("sample_advanced_usage", adv), ("static_octav_wi8_ai8", octav), ("static_wi4_ai8_fc", w4)),
                out_name="ref_mixed_params.json", skip_covered=False)


if __name__ == "__main__" and "--mixed" in sys.argv:
  mixed_cases()


# --------------------------------------------------------------------------- float casting ---
def fp16_cases():
  """Weight-only FP16 (algorithm "float_casting") on the models that carry weights."""
  def rcp(operation):
    return [dict(regex=".*", operation=operation, algorithm_key="float_casting", op_config=dict(
        weight_tensor_config=dict(num_bits=16, symmetric=True, granularity="CHANNELWISE", dtype="FLOAT"),
        compute_precision="FLOAT", explicit_dequantize=True, skip_checks=False, min_weight_elements=0))]
  recipes = {"fp16_all": rcp("*"), "fp16_fc": rcp("FULLY_CONNECTED")}
  cases = {}
  for name in ["single_fc", "single_fc_bias", "single_fc_no_bias", "conv_fc_mnist", "embedding_lookup",
               "single_conv2d_transpose_bias", "single_depthwise_conv2d_bias", "weight_sharing_fcs",
               "toy_model_with_kv_cache_multi_signature", "branching_conv_fc",
               "constant_tensor_and_buffer_only_sharing_weight_fcs", "bmm_constant_input", "two_signatures"]:
    for rname, r in recipes.items():
      key = f"{name}/{rname}"
      try:
        cases[key] = dict(model=name, recipe_name=rname, recipe=r, result=run(name, rname, r))
        print("ok  ", key)
      except Exception as e:
        cases[key] = dict(model=name, recipe_name=rname, recipe=r, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
  with open(os.path.join(GOLDEN, "ref_fp16_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --fp16", numpy=np.__version__,
                   cases=json.loads(json.dumps(cases, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--fp16" in sys.argv:
  fp16_cases()


# ------------------------------------------------------------- dequantized weight recovery ---
def dwr_model_cases():
  """The reference's fake-quantized FC models under dequantized_weight_recovery recipes (matching
  and mismatching granularities: the mismatches are refused by the reference)."""
  os.makedirs(os.path.join(GOLDEN, "models", "dequantized_weights"), exist_ok=True)
  # the recovered scales are plain ndarrays (np.hstack drops the bf16-aware subclass of the shim):
  # re-register the algorithm through the reference's own registry API with a wrapper that only
  # re-attaches the subclass, so that quantize_tensor's `.astype(ml_dtypes.bfloat16)` resolves
  import dataclasses
  import functools
  from ai_edge_quantizer import algorithm_manager as ref_am
  from ai_edge_quantizer.algorithms.uniform_quantize import common_quantize as ref_cq
  from ai_edge_quantizer.algorithms.uniform_quantize import dequantized_weight_recovery as ref_dwr

  def bf16_aware_params(*a, **k):
    res = ref_dwr.get_tensor_quant_params(*a, **k)
    return dataclasses.replace(res, scale=np.asarray(res.scale).view(ml_dtypes.Bf16Aware))
  ref_am.register_quantized_op(
      algorithm_key="dequantized_weight_recovery", tfl_op_name=qtyping.TFLOperationName.FULLY_CONNECTED,
      init_qsv_func=ref_dwr.init_qsvs, calibration_func=ref_dwr.calibrate,
      materialize_func=functools.partial(ref_cq.materialize_fc_conv, bf16_aware_params))

  def rcp(gran, precision="INTEGER", explicit=False):
    return [dict(regex=".*", operation="FULLY_CONNECTED", algorithm_key="dequantized_weight_recovery", op_config=dict(
        weight_tensor_config=dict(num_bits=4, symmetric=True, granularity=gran, dtype="INT"),
        compute_precision=precision, explicit_dequantize=explicit, skip_checks=False, min_weight_elements=0))]
  cases = {}
  for stem in ("channel_i4rangedvalues_fc", "tensor_i4rangedvalues_fc", "blockwise_i4rangedvalues_fc"):
    name = "dequantized_weights/" + stem
    shutil.copyfile(os.path.join(REF, "tests/models", name + ".tflite"), os.path.join(GOLDEN, "models", name + ".tflite"))
    for rname, r in (("dwr_cw", rcp("CHANNELWISE")), ("dwr_tw", rcp("TENSORWISE")), ("dwr_b32", rcp("BLOCKWISE_32")),
                     ("dwr_cw_weight_only", rcp("CHANNELWISE", "FLOAT", True))):
      key = f"{name}/{rname}"
      try:
        cases[key] = dict(model=name, recipe_name=rname, recipe=r, result=run(name, rname, r))
        print("ok  ", key)
      except Exception as e:
        cases[key] = dict(model=name, recipe_name=rname, recipe=r, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:150])
  with open(os.path.join(GOLDEN, "ref_dwr_model_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --dwr", numpy=np.__version__,
                   cases=json.loads(json.dumps(cases, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--dwr" in sys.argv:
  dwr_model_cases()


# ----------------------------------------------------------------------- random graphs ---
def random_graph(seed: int, wide: bool = False):
  """A small random float graph built with this repository's own flatbuffer classes: chains and
  fan-outs of the ops the registry knows (unary, binary with activation or constant operand,
  shape-preserving data movement with constant index operands, FULLY_CONNECTED with shared or
  private weights), several graph outputs. Shapes are nominal: the quantizer never checks them."""
  from mi355q import qtyping as q
  B = our_schema.BuiltinOperator
  rng = np.random.default_rng(seed)
  model = q.ModelT(version=3, description=b"mi355q random graph")
  model.buffers = [q.BufferT()]
  model.operatorCodes = []
  sg = q.SubGraphT(name=b"main", tensors=[], operators=[], inputs=[], outputs=[])
  codes = {}

  def code(op):
    if op not in codes:
      model.operatorCodes.append(q.OperatorCodeT(builtinCode=int(op), deprecatedBuiltinCode=min(int(op), 127)))
      codes[op] = len(model.operatorCodes) - 1
    return codes[op]

  def act(shape=(1, 8)):
    sg.tensors.append(q.TensorT(name=f"t{len(sg.tensors)}".encode(), shape=list(shape), type=0, buffer=0))
    return len(sg.tensors) - 1

  def const(arr, ttype=0, share=None):
    if share is None:
      model.buffers.append(q.BufferT(data=np.ascontiguousarray(arr).reshape(-1).view(np.uint8)))
      share = len(model.buffers) - 1
    sg.tensors.append(q.TensorT(name=f"c{len(sg.tensors)}".encode(), shape=list(arr.shape), type=ttype, buffer=share))
    return len(sg.tensors) - 1, share

  live = [act() for _ in range(int(rng.integers(1, 3)))]
  sg.inputs = list(live)
  fc_weights = []
  unary = [B.TANH, B.LOGISTIC, B.RELU, B.GELU, B.SQRT, B.RSQRT, B.HARD_SWISH, B.SOFTMAX]
  binary = [B.ADD, B.SUB, B.MUL, B.DIV, B.MAXIMUM, B.SQUARED_DIFFERENCE]
  moves = [(B.RESHAPE, np.array([1, 8], np.int32)), (B.TRANSPOSE, np.array([0, 1], np.int32)),
           (B.PAD, np.zeros((2, 2), np.int32)), (B.MEAN, np.array([1], np.int32)), (B.SUM, np.array([1], np.int32))]
  if wide:     # the second wave of graphs draws from more op kinds (multi-output, pooling, gathers)
    moves += [(B.SLICE, None), (B.STRIDED_SLICE, None), (B.RESIZE_BILINEAR, np.array([2, 2], np.int32)),
              (B.RESIZE_NEAREST_NEIGHBOR, np.array([2, 2], np.int32)), (B.GATHER, np.array([0, 1], np.int32)),
              (B.BROADCAST_TO, np.array([1, 8], np.int32)), (B.MIRROR_PAD, np.zeros((2, 2), np.int32)),
              (B.REDUCE_MIN, np.array([1], np.int32)), (B.GATHER_ND, np.array([[0]], np.int32))]
    unary += [B.AVERAGE_POOL_2D, B.MAX_POOL_2D, B.SPACE_TO_DEPTH]
    binary += [B.PACK, B.EQUAL, B.NOT_EQUAL]
  for _ in range(int(rng.integers(3, 9))):
    kinds = ["unary", "binary", "binary_const", "move", "fc", "fc", "concat"]
    if wide:
      kinds += ["split", "unpack", "select", "bmm", "embedding"]
    kind = rng.choice(kinds)
    x = int(rng.choice(live))
    out = act()
    if kind == "split":        # axis constant first, two outputs
      c, _ = const(np.array(1, np.int32), ttype=2)
      out2 = act()
      sg.operators.append(q.OperatorT(opcodeIndex=code(B.SPLIT), inputs=[c, x], outputs=[out, out2]))
      live += [out, out2]
      continue
    if kind == "unpack":
      out2 = act()
      sg.operators.append(q.OperatorT(opcodeIndex=code(B.UNPACK), inputs=[x], outputs=[out, out2]))
      live += [out, out2]
      continue
    if kind == "select":       # boolean condition first
      sg.tensors.append(q.TensorT(name=f"cond{len(sg.tensors)}".encode(), shape=[1, 8], type=6, buffer=0))
      cond = len(sg.tensors) - 1
      sg.inputs.append(cond)
      op = q.OperatorT(opcodeIndex=code(B.SELECT_V2 if rng.random() < 0.5 else B.SELECT),
                       inputs=[cond, x, int(rng.choice(live))], outputs=[out])
      sg.operators.append(op)
      live.append(out)
      continue
    if kind == "bmm":
      c, _ = const(rng.standard_normal((1, 8, 8)).astype(np.float32))
      op = q.OperatorT(opcodeIndex=code(B.BATCH_MATMUL), inputs=[x, c], outputs=[out], builtinOptionsType=101,
                       builtinOptions=q.BatchMatMulOptionsT(adjY=bool(rng.random() < 0.5)))
      sg.operators.append(op)
      live.append(out)
      continue
    if kind == "embedding":
      sg.tensors.append(q.TensorT(name=f"ids{len(sg.tensors)}".encode(), shape=[1], type=2, buffer=0))
      ids = len(sg.tensors) - 1
      sg.inputs.append(ids)
      c, _ = const(rng.standard_normal((16, 32)).astype(np.float32))
      sg.operators.append(q.OperatorT(opcodeIndex=code(B.EMBEDDING_LOOKUP), inputs=[ids, c], outputs=[out]))
      live.append(out)
      continue
    if kind == "unary":
      op = q.OperatorT(opcodeIndex=code(unary[int(rng.integers(len(unary)))]), inputs=[x], outputs=[out])
    elif kind == "binary":
      op = q.OperatorT(opcodeIndex=code(binary[int(rng.integers(len(binary)))]), inputs=[x, int(rng.choice(live))], outputs=[out])
    elif kind == "binary_const":
      c, _ = const(rng.standard_normal((1, 8)).astype(np.float32))
      op = q.OperatorT(opcodeIndex=code(binary[int(rng.integers(3))]), inputs=[x, c], outputs=[out])
    elif kind == "move":
      bop, arg = moves[int(rng.integers(len(moves)))]
      if arg is None:            # SLICE (begin, size) / STRIDED_SLICE (begin, end, strides)
        extra = [const(np.array([0, 0], np.int32), ttype=2)[0] for _ in range(2 if bop == B.SLICE else 3)]
        op = q.OperatorT(opcodeIndex=code(bop), inputs=[x] + extra, outputs=[out])
      else:
        c, _ = const(arg, ttype=2)
        op = q.OperatorT(opcodeIndex=code(bop), inputs=[x, c], outputs=[out])
    elif kind == "concat":
      op = q.OperatorT(opcodeIndex=code(B.CONCATENATION), inputs=[x, int(rng.choice(live))], outputs=[out])
    else:
      if fc_weights and rng.random() < 0.35:          # a second tensor over the same weight buffer
        w, _ = const(fc_weights[0][1], share=fc_weights[0][0])
      else:
        arr = rng.standard_normal((8, 8)).astype(np.float32) * np.float32(rng.uniform(0.1, 3))
        w, bufid = const(arr)
        fc_weights.append((bufid, arr))
      ins = [x, w, -1]
      if rng.random() < 0.5:
        ins[2], _ = const(rng.standard_normal(8).astype(np.float32))
      op = q.OperatorT(opcodeIndex=code(B.FULLY_CONNECTED), inputs=ins, outputs=[out], builtinOptionsType=8,
                       builtinOptions=q.FullyConnectedOptionsT())
    sg.operators.append(op)
    live.append(out)
  consumed = {t for op in sg.operators for t in op.inputs}
  sg.outputs = [t for t in live if t not in consumed and t not in sg.inputs] or [live[-1]]
  if rng.random() < 0.4 and len(live) > 2:
    extra = int(rng.choice(live[1:-1]))
    if extra not in sg.outputs and extra not in sg.inputs:
      sg.outputs.append(extra)
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(
      signatureKey=b"serving_default", subgraphIndex=0,
      inputs=[q.TensorMapT(name=sg.tensors[t].name, tensorIndex=t) for t in sg.inputs],
      outputs=[q.TensorMapT(name=sg.tensors[t].name, tensorIndex=t) for t in sg.outputs])]
  return model


def random_graph_cases(count=96):
  from ai_edge_quantizer import recipe as ref_recipe
  os.makedirs(os.path.join(GOLDEN, "models", "random"), exist_ok=True)
  out = {}
  for seed in range(count):
    model = random_graph(7000 + seed, wide=seed >= 48)
    name = f"random/graph_{seed:02d}"
    path = os.path.join(GOLDEN, "models", name + ".tflite")
    with open(path, "wb") as f:
      f.write(fb.write_model(model))
    parsed = to_bags(fb.read_model(open(path, "rb").read()))
    rng = np.random.default_rng(8000 + seed)
    qsvs = {}
    for t in parsed.subgraphs[0].tensors:
      if parsed.buffers[t.buffer].data is None:
        lo, hi = sorted(rng.uniform(-6, 6, 2))
        qsvs[t.name.decode()] = {"min": np.array([[min(lo, -0.1)]], np.float32), "max": np.array([[max(hi, 0.1)]], np.float32)}
    seed_qsvs = {k: {"min": float(v["min"].ravel()[0]), "max": float(v["max"].ravel()[0])} for k, v in qsvs.items()}
    for rname, rcp, needs in (("static_wi8_ai8", ref_recipe.static_wi8_ai8(), True),
                              ("static_wi8_ai16", ref_recipe.static_wi8_ai16(), True),
                              ("dynamic_wi8_afp32", RECIPES["dynamic_wi8_afp32"], False),
                              ("weight_only_wi4_afp32", RECIPES["weight_only_wi4_afp32"], False)):
      key = f"{name}/{rname}"
      try:
        res = run(name, rname, rcp, {k: dict(v) for k, v in qsvs.items()} if needs else None, path=path)
        out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs if needs else None, result=res)
        print("ok  ", key)
      except Exception as e:
        out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs if needs else None, error=type(e).__name__,
                        message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:120])
  with open(os.path.join(GOLDEN, "ref_random_graph_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --random", numpy=np.__version__,
                   cases=json.loads(json.dumps(out, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--random" in sys.argv[1:2]:
  random_graph_cases()


def random_graph_mixed_cases():
  """The second wave of random graphs under recipes that mix modes by scope and by op: float
  islands inside static graphs, weight-only FULLY_CONNECTED next to static elementwise ops."""
  from ai_edge_quantizer import recipe as ref_recipe
  def act(bits, sym):
    return dict(num_bits=bits, symmetric=sym, granularity="TENSORWISE", dtype="INT")
  def w(bits, gran="CHANNELWISE"):
    return dict(num_bits=bits, symmetric=True, granularity=gran, dtype="INT")
  def entry(regex, op, key, **cfg):
    return dict(regex=regex, operation=op, algorithm_key=key, op_config=dict(
        skip_checks=False, min_weight_elements=0, **cfg)) if cfg else dict(regex=regex, operation=op, algorithm_key=key)
  mm = "min_max_uniform_quantize"
  srq8 = dict(activation_tensor_config=act(8, False), weight_tensor_config=w(8), compute_precision="INTEGER",
              explicit_dequantize=False)
  srq16 = dict(activation_tensor_config=act(16, True), weight_tensor_config=w(8), compute_precision="INTEGER",
               explicit_dequantize=False)
  recipes = {
      "static8_fc_weight_only": [entry(".*", "*", mm, **srq8),
                                 entry(".*[02468];", "FULLY_CONNECTED", mm, weight_tensor_config=w(8),
                                       compute_precision="FLOAT", explicit_dequantize=True)],
      "static16_float_island": [entry(".*", "*", mm, **srq16), entry("t[3-6];", "*", "no_quantize")],
      "dynamic4_fc_static8_elementwise": [
          entry(".*", "FULLY_CONNECTED", mm, weight_tensor_config=w(4), compute_precision="INTEGER",
                explicit_dequantize=False),
          entry(".*", "ADD", mm, **srq8), entry(".*", "MUL", mm, **srq8), entry(".*", "TANH", mm, **srq8)],
  }
  out = {}
  for seed in range(48, 96):
    name = f"random/graph_{seed:02d}"
    path = os.path.join(GOLDEN, "models", name + ".tflite")
    parsed = to_bags(fb.read_model(open(path, "rb").read()))
    rng = np.random.default_rng(8000 + seed)
    qsvs = {}
    for t in parsed.subgraphs[0].tensors:
      if parsed.buffers[t.buffer].data is None:
        lo, hi = sorted(rng.uniform(-6, 6, 2))
        qsvs[t.name.decode()] = {"min": np.array([[min(lo, -0.1)]], np.float32), "max": np.array([[max(hi, 0.1)]], np.float32)}
    seed_qsvs = {k: {"min": float(v["min"].ravel()[0]), "max": float(v["max"].ravel()[0])} for k, v in qsvs.items()}
    for rname, rcp in recipes.items():
      key = f"{name}/{rname}"
      try:
        res = run(name, rname, rcp, {k: dict(v) for k, v in qsvs.items()}, path=path)
        out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs, result=res)
        print("ok  ", key)
      except Exception as e:
        out[key] = dict(model=name, recipe=rcp, qsvs=seed_qsvs, error=type(e).__name__, message=str(e)[:300])
        print("err ", key, type(e).__name__, str(e)[:140])
  with open(os.path.join(GOLDEN, "ref_random_graph_mixed_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_model_golden.py --random-mixed", numpy=np.__version__,
                   cases=json.loads(json.dumps(out, default=str))), f, separators=(",", ":"), sort_keys=True)


if __name__ == "__main__" and "--random-mixed" in sys.argv:
  random_graph_mixed_cases()
