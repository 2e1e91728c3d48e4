#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/gen/make_golden.py

The reference package cannot be imported as shipped (LiteRT, absl, ml_dtypes,
immutabledict, flatbuffers are absent), so this script
  1. registers an empty package object for `ai_edge_quantizer` whose __path__
     points at /root/reference/ai_edge_quantizer (skips its __init__, which
     pulls the LiteRT interpreter), and
  2. puts tests/golden/gen/shim/ first on sys.path (non-arithmetic stand-ins
     plus a bfloat16 RNE stand-in for ml_dtypes),
then imports the reference's arithmetic modules unmodified and records their
outputs on seeded inputs:
    tests/golden/ref_cases.npz     inputs + outputs of small cases
    tests/golden/ref_cases.json    per-case parameters (what was called, how)
    tests/golden/ref_digests.json  SHA-256 of outputs at BASELINE sizes
Nothing from the reference is copied: fixtures are inputs and outputs only.
"""
import hashlib
import json
import os
import sys
sys.dont_write_bytecode = True  # never leave .pyc files in the read-only reference tree
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
REF = "/root/reference/ai_edge_quantizer"


def bootstrap():
  sys.path.insert(0, os.path.join(HERE, "shim"))
  pkg = types.ModuleType("ai_edge_quantizer")
  pkg.__path__ = [REF]
  sys.modules["ai_edge_quantizer"] = pkg


bootstrap()
import ml_dtypes  # the shim  # noqa: E402
from ai_edge_quantizer import qtyping  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import common_quantize  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import gptq  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import hadamard_rotation  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import mse  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import naive_min_max_quantize as mm  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import octav  # noqa: E402
from ai_edge_quantizer.algorithms.uniform_quantize import uniform_quantize_tensor as uqt  # noqa: E402
from ai_edge_quantizer.transformations import transformation_utils  # noqa: E402
from ai_edge_quantizer.utils import qsv_utils  # noqa: E402

G = qtyping.QuantGranularity
OPN = qtyping.TFLOperationName

ARR = {}     # name -> ndarray (npz payload)
CASES = []   # list of dicts (json payload)
DIGESTS = {}


def sha(a) -> str:
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def op_info(op_name, cfg):
  return qtyping.OpInfo(
      op=qtyping.OperatorT(), op_name=op_name, subgraph_op_index=0,
      op_quant_config=qtyping.OpQuantizationConfig(weight_tensor_config=cfg))


def cfg_of(bits, sym, gran, **algo):
  return qtyping.TensorQuantizationConfig(num_bits=bits, symmetric=sym,
                                          granularity=gran, algorithm_params=algo)


def plain(a):
  return None if a is None else np.asarray(a).view(np.ndarray)


def as_input(w, gran):
  # blockwise scales go through `.astype(ml_dtypes.bfloat16)`; feed the
  # subclass so the unpatched reference source resolves it (see shim/ml_dtypes).
  return w.view(ml_dtypes.Bf16Aware) if "BLOCKWISE" in gran.name else w


def record(name, algo, w, params, res, extra=None):
  ARR[f"{name}/w"] = plain(w)
  ARR[f"{name}/scale"] = plain(res.scale)
  ARR[f"{name}/zero_point"] = plain(res.zero_point)
  if res.quantized_data is not None:
    ARR[f"{name}/q"] = plain(res.quantized_data)
  case = dict(name=name, algo=algo, quantized_dimension=res.quantized_dimension,
              block_size=res.block_size, **params)
  if extra:
    for k, v in extra.items():
      if isinstance(v, np.ndarray):
        ARR[f"{name}/{k}"] = plain(v)
      else:
        case[k] = v
  CASES.append(case)


def gen_weight(seed, shape, kind="normal"):
  rng = np.random.default_rng(seed)
  w = rng.standard_normal(shape, dtype=np.float32)
  if kind == "outlier":
    idx = rng.integers(0, w.size, size=max(1, w.size // 1000))
    w.reshape(-1)[idx] *= 50
  elif kind == "zero_row":
    w[1] = 0
    w[3, 5] = 1e4
  elif kind == "small":
    w *= np.float32(0.02)
  elif kind == "uniform":
    w = rng.uniform(-10, 10, size=shape).astype(np.float32)
  return w


# ---------------------------------------------------------------- min/max ---
def minmax_cases():
  specs = [
      ("mm_cw_i8", (48, 256), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i4", (48, 256), 4, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i2", (16, 128), 2, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i8_outlier", (48, 256), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "outlier"),
      ("mm_cw_i8_zero_row", (48, 256), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "zero_row"),
      ("mm_cw_i8_asym", (32, 200), 8, False, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i4_asym", (32, 200), 4, False, G.CHANNELWISE, OPN.FULLY_CONNECTED, "uniform"),
      ("mm_tw_i8", (40, 100), 8, True, G.TENSORWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_tw_i4", (40, 100), 4, True, G.TENSORWISE, OPN.FULLY_CONNECTED, "outlier"),
      ("mm_tw_i8_asym", (40, 100), 8, False, G.TENSORWISE, OPN.FULLY_CONNECTED, "uniform"),
      ("mm_cw_i8_ragged", (7, 37), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i4_ragged", (5, 33), 4, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_cw_i8_one", (1, 1), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_emb_cw_i8", (100, 64), 8, True, G.CHANNELWISE, OPN.EMBEDDING_LOOKUP, "normal"),
      ("mm_conv_cw_i8", (8, 3, 3, 16), 8, True, G.CHANNELWISE, OPN.CONV_2D, "normal"),
      ("mm_dwconv_cw_i8", (1, 3, 3, 24), 8, True, G.CHANNELWISE, OPN.DEPTHWISE_CONV_2D, "normal"),
      ("mm_fc3d_cw_i8", (6, 5, 64), 8, True, G.CHANNELWISE, OPN.FULLY_CONNECTED, "normal"),
      ("mm_bw32_i4", (24, 256), 4, True, G.BLOCKWISE_32, OPN.FULLY_CONNECTED, "uniform"),
      ("mm_bw64_i4", (24, 256), 4, True, G.BLOCKWISE_64, OPN.FULLY_CONNECTED, "normal"),
      ("mm_bw128_i4", (24, 512), 4, True, G.BLOCKWISE_128, OPN.FULLY_CONNECTED, "small"),
      ("mm_bw256_i4", (24, 512), 4, True, G.BLOCKWISE_256, OPN.FULLY_CONNECTED, "normal"),
      ("mm_bw32_i8", (24, 256), 8, True, G.BLOCKWISE_32, OPN.FULLY_CONNECTED, "normal"),
      ("mm_bw128_i2", (8, 256), 2, True, G.BLOCKWISE_128, OPN.FULLY_CONNECTED, "normal"),
      ("mm_bw32_i4_zero_block", (24, 128), 4, True, G.BLOCKWISE_32, OPN.FULLY_CONNECTED, "zero_row"),
      ("mm_bw32_i4_tiny", (8, 128), 4, True, G.BLOCKWISE_32, OPN.FULLY_CONNECTED, "tiny"),
      ("mm_bw32_i4_huge", (8, 128), 4, True, G.BLOCKWISE_32, OPN.FULLY_CONNECTED, "huge"),
      ("mm_emb_bw32_i4", (50, 64), 4, True, G.BLOCKWISE_32, OPN.EMBEDDING_LOOKUP, "normal"),
  ]
  for i, (name, shape, bits, sym, gran, opn, kind) in enumerate(specs):
    if kind == "tiny":
      w = gen_weight(100 + i, shape) * np.float32(1e-7)
    elif kind == "huge":
      w = gen_weight(100 + i, shape) * np.float32(1e7)
    else:
      w = gen_weight(100 + i, shape, kind)
    cfg = cfg_of(bits, sym, gran)
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      res = mm.get_tensor_quant_params(op_info(opn, cfg), cfg, as_input(w, gran))
    record(name, "min_max", w,
           dict(num_bits=bits, symmetric=sym, granularity=gran.name, op=opn.name),
           res)

  # activation-style: params from a supplied QSV, no content (ref naive_min_max:101-102)
  for j, (bits, sym) in enumerate([(8, False), (8, True), (16, True)]):
    cfg = cfg_of(bits, sym, G.TENSORWISE)
    qsv = {"min": np.array([[-3.25 - j]], np.float32),
           "max": np.array([[7.5 + j]], np.float32)}
    res = mm.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, None, qsv)
    name = f"mm_act_{bits}_{'sym' if sym else 'asym'}"
    ARR[f"{name}/min"], ARR[f"{name}/max"] = qsv["min"], qsv["max"]
    ARR[f"{name}/scale"], ARR[f"{name}/zero_point"] = plain(res.scale), plain(res.zero_point)
    CASES.append(dict(name=name, algo="min_max_qsv", num_bits=bits, symmetric=sym,
                      granularity="TENSORWISE", op="FULLY_CONNECTED",
                      quantized_dimension=res.quantized_dimension, block_size=0))


# ------------------------------------------------------------------ OCTAV ---
def octav_cases():
  specs = [
      ("oct_cw_i4", (32, 512), 4, G.CHANNELWISE, "normal"),
      ("oct_cw_i8", (32, 512), 8, G.CHANNELWISE, "normal"),
      ("oct_cw_i4_outlier", (32, 512), 4, G.CHANNELWISE, "outlier"),
      ("oct_cw_i4_small", (32, 512), 4, G.CHANNELWISE, "small"),
      ("oct_tw_i4", (16, 200), 4, G.TENSORWISE, "normal"),
      ("oct_bw32_i4", (16, 256), 4, G.BLOCKWISE_32, "normal"),
      ("oct_bw128_i4", (16, 512), 4, G.BLOCKWISE_128, "uniform"),
      ("oct_cw_i4_ragged", (5, 77), 4, G.CHANNELWISE, "normal"),
      ("oct_cw_i4_zero_row", (8, 64), 4, G.CHANNELWISE, "zero_row"),
  ]
  for i, (name, shape, bits, gran, kind) in enumerate(specs):
    w = gen_weight(300 + i, shape, kind)
    cfg = cfg_of(bits, True, gran)
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      res = octav.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg,
                                          as_input(w, gran))
      # also record the raw clipping constants + iteration behaviour
      if "BLOCKWISE" in gran.name:
        data, axis = uqt.reshape_data_for_blockwise(w, OPN.FULLY_CONNECTED, gran)
      elif gran == G.CHANNELWISE:
        data, axis = w, (1,)
      else:
        data, axis = w, None
      clip = octav._guess_clipping_with_octav(data, bits, axis, 10, 3.0)
      clip_noes = octav._guess_clipping_with_octav(data, bits, axis, 10, 3.0,
                                                   early_stop=False)
    record(name, "octav", w,
           dict(num_bits=bits, symmetric=True, granularity=gran.name,
                op="FULLY_CONNECTED"), res,
           extra=dict(clip=np.asarray(clip), clip_no_early_stop=np.asarray(clip_noes)))


# -------------------------------------------------------------------- MSE ---
def mse_cases():
  for i, (name, shape, bits, gran) in enumerate([
      ("mse_cw_i8", (32, 300), 8, G.CHANNELWISE),
      ("mse_cw_i4", (32, 300), 4, G.CHANNELWISE),
      ("mse_cw_i4_wide", (4, 5000), 4, G.CHANNELWISE),
      ("mse_tw_i8", (16, 64), 8, G.TENSORWISE),
  ]):
    w = gen_weight(400 + i, shape)
    cfg = cfg_of(bits, True, gran)
    res = mse.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, w)
    record(name, "mse", w, dict(num_bits=bits, symmetric=True,
                                granularity=gran.name, op="FULLY_CONNECTED"), res)


# --------------------------------------------------------------- Hadamard ---
def hadamard_cases():
  for i, (name, shape, bits, gran, mx) in enumerate([
      ("had_cw_i8_h256", (16, 256), 8, G.CHANNELWISE, None),
      ("had_cw_i4_h512", (16, 512), 4, G.CHANNELWISE, None),
      ("had_cw_i4_h32_of_96", (16, 96), 4, G.CHANNELWISE, None),
      ("had_cw_i4_max128", (16, 512), 4, G.CHANNELWISE, 128),
      ("had_cw_i4_max100", (16, 512), 4, G.CHANNELWISE, 100),
      ("had_bw32_i4", (16, 256), 4, G.BLOCKWISE_32, None),
      ("had_3d_i8", (4, 6, 64), 8, G.CHANNELWISE, None),
  ]):
    w = gen_weight(500 + i, shape)
    algo = {} if mx is None else {"max_hadamard_size": mx}
    cfg = cfg_of(bits, True, gran, **algo)
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      res = hadamard_rotation.get_tensor_quant_params(
          op_info(OPN.FULLY_CONNECTED, cfg), cfg, as_input(w, gran))
      rot, h, _ = hadamard_rotation._rotate_with_diagonal_hadamard(
          w, axis=w.ndim - 1, max_size=mx)
    record(name, "hadamard", w,
           dict(num_bits=bits, symmetric=True, granularity=gran.name,
                op="FULLY_CONNECTED", max_hadamard_size=mx), res,
           extra=dict(rotated=np.asarray(rot), hadamard_size=int(h),
                      random_binary_vector=res.hadamard.random_binary_vector))


# ------------------------------------------------------------------- GPTQ ---
def gptq_cases():
  for i, (name, rows, d, bits, sym, gran, n_s, n_t) in enumerate([
      ("gptq_cw_i4", 24, 64, 4, True, G.CHANNELWISE, 4, 48),
      ("gptq_cw_i8_2blk", 16, 160, 8, True, G.CHANNELWISE, 2, 200),
      ("gptq_tw_i4", 16, 96, 4, True, G.TENSORWISE, 3, 64),
      ("gptq_bw32_i4", 16, 128, 4, True, G.BLOCKWISE_32, 4, 64),
      ("gptq_cw_i4_asym", 16, 64, 4, False, G.CHANNELWISE, 4, 48),
  ]):
    rng = np.random.default_rng(600 + i)
    w = (rng.standard_normal((rows, d), dtype=np.float32) * np.float32(0.05))
    x = rng.standard_normal((n_s, n_t, d), dtype=np.float32)
    x[..., 3] *= 4.0  # an outlier channel so the Hessian is not near-identity
    # reference Hessian via gptq.calibrate's formula on a live tensor (ref gptq.py:100-107)
    num_samples = np.array(x.shape[0])
    x2 = x.reshape([-1, x.shape[-1]])
    hess = (2.0 / num_samples) * x2.T.dot(x2)
    cfg = cfg_of(bits, sym, gran)
    qsv = {"activation_tensor_qsv": {"hessian": hess.copy(), "num_samples": int(n_s)}}
    hinv = gptq._prepare_hessian_inverse(hess.copy())
    with warnings.catch_warnings():
      warnings.simplefilter("ignore")
      res = gptq.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg,
                                         as_input(w, gran), qsv)
    record(name, "gptq", w,
           dict(num_bits=bits, symmetric=sym, granularity=gran.name,
                op="FULLY_CONNECTED"), res,
           extra=dict(x=x, hessian=np.asarray(hess), hinv=np.asarray(hinv)))


# ---------------------------------------------------- activations and QSVs ---
def activation_cases():
  rng = np.random.default_rng(700)
  tensors = {
      "act_plain": rng.standard_normal((2, 8, 32), dtype=np.float32) * 3,
      "act_inf": np.array([[-np.inf, 1.0, 5.0, np.inf, 3.39e38]], np.float32),
      "act_neg_sentinel": np.array([[6.0, 7.0, -3.39e38, 9.0, np.inf]], np.float32),
      "act_all_masked_hi": np.array([[3.2e38, 3.39e38, np.inf]], np.float32),
      "act_all_masked_lo": np.array([[-3.2e38, -3.39e38, -np.inf]], np.float32),
      "act_scalar": np.array(2.5, np.float32),
      "act_int": np.array([1, 2, -10, 10], np.int32),
  }
  for name, x in tensors.items():
    q = common_quantize.get_activation_min_max(x, -3e38, 3e38)
    ARR[f"{name}/x"], ARR[f"{name}/min"], ARR[f"{name}/max"] = x, q["min"], q["max"]
    CASES.append(dict(name=name, algo="activation_min_max", lo=-3e38, hi=3e38))

  # EMA replay over 16 samples (default 0.95) and min_max_update, scalar shaped (1,1,1)
  mins = rng.standard_normal((16, 1, 1, 1)).astype(np.float32) - 3
  maxs = rng.standard_normal((16, 1, 1, 1)).astype(np.float32) + 3
  q_ema, q_mm = None, None
  for a, b in zip(mins, maxs):
    new = {"min": a, "max": b}
    q_ema = qsv_utils.moving_average_update(q_ema, new)
    q_mm = qsv_utils.min_max_update(q_mm, new)
  ARR["qsv_replay/mins"], ARR["qsv_replay/maxs"] = mins, maxs
  ARR["qsv_replay/ema_min"], ARR["qsv_replay/ema_max"] = q_ema["min"], q_ema["max"]
  ARR["qsv_replay/mm_min"], ARR["qsv_replay/mm_max"] = q_mm["min"], q_mm["max"]
  CASES.append(dict(name="qsv_replay", algo="qsv_replay", smoothing_factor=0.95))

  # Hessian merge across 3 uneven sample groups
  hs = [rng.standard_normal((8, 8)).astype(np.float32) for _ in range(3)]
  ns = [2, 5, 1]
  q = None
  for h, n in zip(hs, ns):
    new = {"min": np.float32(-1), "max": np.float32(1), "hessian": h @ h.T,
           "num_samples": n}
    q = qsv_utils.gptq_and_moving_average_update(q, new)
  for k, h in enumerate(hs):
    ARR[f"qsv_hessian/h{k}"] = h @ h.T
  ARR["qsv_hessian/merged"] = q["hessian"]
  CASES.append(dict(name="qsv_hessian", algo="qsv_hessian", num_samples=ns,
                    total=int(q["num_samples"])))


# ------------------------------------------------------------------- pack ---
def pack_cases():
  rng = np.random.default_rng(800)
  for bits, n in [(4, 15), (4, 16), (4, 1), (4, 4097), (2, 10), (2, 3), (2, 4096),
                  (2, 4099), (8, 33)]:
    lo, hi = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    data = rng.integers(lo, hi + 1, size=n).astype(np.int8)
    out = transformation_utils.pack_data(bits, data.view(np.uint8))
    name = f"pack_i{bits}_{n}"
    ARR[f"{name}/data"], ARR[f"{name}/packed"] = data, np.asarray(out)
    CASES.append(dict(name=name, algo="pack", num_bits=bits))


# ----------------------------------------------- direct a3 / bias vectors ---
def direct_cases():
  rng = np.random.default_rng(900)
  x = (rng.standard_normal((16, 64)) * 4).astype(np.float32)
  # asymmetric int8 per-tensor with int8 zp, and int32 zp (promotes through f64)
  for name, zp_dtype in [("uq_asym_i8zp", np.int8), ("uq_asym_i32zp", np.int32)]:
    p = qtyping.UniformQuantParams(num_bits=8, quantized_dimension=None,
                                   scale=np.array([[0.0731]], np.float32),
                                   zero_point=np.array([[-7]], zp_dtype),
                                   symmetric=False)
    ARR[f"{name}/x"] = x
    ARR[f"{name}/scale"], ARR[f"{name}/zero_point"] = p.scale, p.zero_point
    ARR[f"{name}/q"] = uqt.uniform_quantize(x, p)
    CASES.append(dict(name=name, algo="uniform_quantize", num_bits=8, symmetric=False,
                      quantized_dimension=None, block_size=0))
  # dequantize channelwise
  q8 = rng.integers(-127, 128, size=(16, 64)).astype(np.int8)
  sc = (rng.random((16, 1)).astype(np.float32) + 0.01)
  p = qtyping.UniformQuantParams(num_bits=8, quantized_dimension=0, scale=sc,
                                 zero_point=np.zeros((16, 1), np.int8))
  ARR["dq_cw/q"], ARR["dq_cw/scale"] = q8, sc
  ARR["dq_cw/zero_point"] = p.zero_point
  ARR["dq_cw/out"] = uqt.uniform_dequantize(q8, p)
  CASES.append(dict(name="dq_cw", algo="uniform_dequantize", quantized_dimension=0,
                    block_size=0))
  # bias (ref uniform_quantize_tensor.py:412-489)
  bias = (rng.standard_normal(16) * 20).astype(np.float32)
  in_p = qtyping.UniformQuantParams(num_bits=8, quantized_dimension=None,
                                    scale=np.array([0.05], np.float32),
                                    zero_point=np.array([3], np.int8), symmetric=False)
  for name, in_bits in [("bias_i32", 8), ("bias_i64", 16)]:
    in_p2 = qtyping.UniformQuantParams(num_bits=in_bits, quantized_dimension=None,
                                       scale=in_p.scale, zero_point=in_p.zero_point,
                                       symmetric=False)
    w_p = qtyping.UniformQuantParams(num_bits=8, quantized_dimension=0, scale=sc,
                                     zero_point=np.zeros((16, 1), np.int8))
    r = uqt.symmetric_quantize_bias_tensor(bias, in_p2, w_p)
    ARR[f"{name}/bias"], ARR[f"{name}/in_scale"], ARR[f"{name}/w_scale"] = bias, in_p.scale, sc
    ARR[f"{name}/q"], ARR[f"{name}/scale"] = r.quantized_data, r.scale
    CASES.append(dict(name=name, algo="bias", in_num_bits=in_bits, num_bits=r.num_bits,
                      quantized_dimension=r.quantized_dimension))


# --------------------------------------------- BASELINE-size digests only ---
def digest_cases():
  t = {}
  # C2 (SURVEY Appendix C anchors)
  w = np.random.default_rng(1234).standard_normal((4096, 4096), dtype=np.float32)
  cfg = cfg_of(8, True, G.CHANNELWISE)
  r = mm.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, w)
  t["c2"] = dict(seed=1234, shape=[4096, 4096], w=sha(w), q=sha(r.quantized_data),
                 scale=sha(r.scale), zero_point=sha(r.zero_point),
                 scale_head=[float(v) for v in r.scale[:3, 0]],
                 q_head=[int(v) for v in r.quantized_data[0, :8]])
  w2 = w.copy()
  w2[7, :] = 0
  w2[9, 5] = 1e4
  r = mm.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, w2)
  t["c2_variant"] = dict(seed=1234, shape=[4096, 4096], edits="w[7,:]=0; w[9,5]=1e4",
                         q=sha(r.quantized_data), scale=sha(r.scale),
                         scale_7=float(r.scale[7, 0]), scale_9=float(r.scale[9, 0]))
  cfg4 = cfg_of(4, True, G.CHANNELWISE)
  r = mm.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg4), cfg4, w)
  t["c2_int4"] = dict(seed=1234, shape=[4096, 4096], q=sha(r.quantized_data),
                      packed=sha(transformation_utils.pack_data(
                          4, np.ravel(r.quantized_data).view(np.uint8))),
                      scale=sha(r.scale))
  del w2
  # C3 layer 0 and 1
  for layer in (0, 1):
    w3 = np.random.default_rng(1000 + layer).standard_normal(
        (4096, 11008), dtype=np.float32) * np.float32(0.02)
    cfgb = cfg_of(4, True, G.BLOCKWISE_128)
    r = mm.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfgb), cfgb,
                                   w3.view(ml_dtypes.Bf16Aware))
    q = plain(r.quantized_data)
    sc = plain(r.scale)
    packed = transformation_utils.pack_data(4, np.ravel(q).view(np.uint8))
    f16 = ml_dtypes.round_to_bf16(sc).astype(np.float16)
    t[f"c3_layer{layer}"] = dict(
        seed=1000 + layer, shape=[4096, 11008], mul=0.02, w=sha(w3), q=sha(q),
        packed=sha(packed), scale=sha(sc), scale_f16=sha(f16),
        scale_head=[float(v) for v in sc[0, :3]],
        packed_head=[int(v) for v in packed[:4]])
    del w3, q, packed
  # OCTAV anchor
  wo = np.random.default_rng(77).standard_normal((64, 512), dtype=np.float32)
  r = octav.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg4), cfg4, wo)
  t["octav_anchor"] = dict(seed=77, shape=[64, 512], q=sha(r.quantized_data),
                           scale=sha(r.scale),
                           scale_head=[float(v) for v in r.scale[:3, 0]])
  DIGESTS.update(t)


# ------------------------------------------ C1: full orchestration in memory ---
def c1_cases():
  """BASELINE config 1 through the reference's own ParamsGenerator ->
  transformation instructions -> TransformationPerformer on an in-memory model
  (flatbuffer serialization is third-party and not available)."""
  from ai_edge_quantizer import params_generator, recipe, recipe_manager
  from ai_edge_quantizer import transformation_instruction_generator as tig
  from ai_edge_quantizer import transformation_performer as tp

  def tensor(name, shape, buf):
    t = qtyping.TensorT()
    t.name, t.shape, t.buffer, t.type = name.encode(), list(shape), buf, 0
    return t

  def build():
    w = np.random.default_rng(1234).standard_normal((256, 256), dtype=np.float32)
    model = qtyping.ModelT()
    bufs = [qtyping.BufferT() for _ in range(3)]
    bufs[1].data = w.view(np.uint8).reshape(-1)
    model.buffers = bufs
    sg = qtyping.SubGraphT()
    sg.tensors = [tensor("x", (1, 256), 0), tensor("w", (256, 256), 1), tensor("y", (1, 256), 2)]
    op = qtyping.OperatorT()
    op.inputs, op.outputs, op.opcodeIndex = [0, 1, -1], [2], 0
    sg.operators, sg.inputs, sg.outputs = [op], [0], [2]
    oc = qtyping.OperatorCodeT()
    oc.builtinCode = qtyping.BuiltinOperator.FULLY_CONNECTED
    oc.deprecatedBuiltinCode = 9
    model.operatorCodes, model.subgraphs = [oc], [sg]
    return model, sg, w

  recipes = {
      "c1_dynamic_wi8_afp32": recipe.dynamic_wi8_afp32(),
      "c1_dynamic_wi4_afp32": recipe.dynamic_wi4_afp32(),
      "c1_dynamic_wi4b32_afp32": recipe.dynamic_wi4b32_afp32(),
      "c1_dynamic_wi8_octav": recipe.dynamic_wi8_afp32(algorithm_key="OCTAV"),
  }
  for name, rcp in recipes.items():
    model, sg, w = build()
    if "b32" in name:  # blockwise scales pass through `.astype(ml_dtypes.bfloat16)`
      model.buffers[1].data = w.view(np.uint8).reshape(-1)
    rm = recipe_manager.RecipeManager()
    rm.load_quantization_recipe(rcp)
    pg = params_generator.ParamsGenerator(model)
    if "b32" in name:
      # feed the weight as the bf16-aware subclass (see shim/ml_dtypes.py)
      from ai_edge_quantizer.utils import tfl_flatbuffer_utils as fbu
      orig = fbu.get_tensor_data
      fbu.get_tensor_data = lambda t, b, _o=orig: (
          None if _o(t, b) is None else _o(t, b).view(ml_dtypes.Bf16Aware))
    try:
      with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        params = pg.generate_quantization_parameters(rm)
        insts = tig.TransformationInstructionsGenerator().quant_params_to_transformation_insts(
            params, model)
        tp.TransformationPerformer().transform_graph(insts, model)
    finally:
      if "b32" in name:
        fbu.get_tensor_data = orig
    wt = sg.tensors[1]
    q = wt.quantization
    rec = dict(name=name, algo="c1", recipe=rcp, tensor_type=int(wt.type),
               quantized_dimension=int(q.quantizedDimension), n_tensors=len(sg.tensors),
               transformations={k: [t.name for c in (v.consumers or []) for t in c.transformations]
                                for k, v in params.items()})
    ARR[f"{name}/buffer"] = np.asarray(model.buffers[1].data).view(np.uint8).copy()
    if q.scale is not None:
      ARR[f"{name}/scale"] = np.asarray(q.scale)
      ARR[f"{name}/zero_point"] = np.asarray(q.zeroPoint)
    if q.details is not None:
      st = sg.tensors[q.details.scales]
      rec.update(block_size=int(q.details.blockSize), zero_points=int(q.details.zeroPoints),
                 scales_tensor_name=st.name.decode(), scales_tensor_type=int(st.type),
                 scales_tensor_shape=list(st.shape))
      ARR[f"{name}/scales_f16"] = np.frombuffer(
          bytes(np.asarray(model.buffers[st.buffer].data)), dtype=np.float16).copy()
    CASES.append(json.loads(json.dumps(rec, default=str)))
    DIGESTS[name] = dict(buffer=sha(ARR[f"{name}/buffer"]))


# ------------------------------------------------------------------ OSCAR ---
def oscar_cases():
  """Separate payload (ref_oscar_cases.npz/json): stage outputs + final parameters of the
  reference's OSCAR on seeded problems. The reference strips array subclasses (np.asarray), so
  the bf16 stand-in is attached where its FP64 bounds enter tensor_zp_scale_from_min_max."""
  from ai_edge_quantizer.algorithms.uniform_quantize import oscar
  real = oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max

  def bf16_aware(mn, mx, *a, **k):
    return real(np.asarray(mn).view(ml_dtypes.Bf16Aware), np.asarray(mx).view(ml_dtypes.Bf16Aware), *a, **k)
  arr, cases = {}, []

  def problem(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    w = rng.normal(size=(n, d)).astype(np.float32)
    w[:, :4] *= 20.0
    mu2 = np.exp(rng.normal(size=d))
    mu2[4:8] *= 100.0
    if kind == "dead":            # dead channels + a zero weight column
      mu2[::5] = 0.0
      w[:, 9] = 0.0
    elif kind == "flat":          # scaling cannot help: identity wins
      w = rng.normal(size=(n, d)).astype(np.float32)
      mu2 = np.ones(d)
    elif kind == "grid":          # many equal magnitudes (ties), uniform masses
      w = np.round(w * 2) / 2
      mu2 = None
    elif kind == "nomu2":
      mu2 = None
    return w.astype(np.float32), mu2

  specs = [
      ("oscar_cw_i4", 0, 16, 32, 4, G.CHANNELWISE, "std"),
      ("oscar_b32_i4", 1, 8, 64, 4, G.BLOCKWISE_32, "std"),
      ("oscar_tw_i4", 2, 6, 40, 4, G.TENSORWISE, "std"),
      ("oscar_cw_i8", 3, 24, 96, 8, G.CHANNELWISE, "std"),
      ("oscar_cw_i4_nomu2", 4, 16, 64, 4, G.CHANNELWISE, "nomu2"),
      ("oscar_b32_i4_nomu2", 5, 12, 128, 4, G.BLOCKWISE_32, "nomu2"),
      ("oscar_cw_i4_dead", 6, 20, 50, 4, G.CHANNELWISE, "dead"),
      ("oscar_cw_i4_flat", 7, 16, 48, 4, G.CHANNELWISE, "flat"),
      ("oscar_cw_i4_grid", 8, 16, 200, 4, G.CHANNELWISE, "grid"),
      ("oscar_b64_i4", 9, 40, 512, 4, G.BLOCKWISE_64, "std"),
      ("oscar_b128_i8", 10, 10, 384, 8, G.BLOCKWISE_128, "std"),
      ("oscar_cw_i4_wide", 11, 5, 3000, 4, G.CHANNELWISE, "std"),
      ("oscar_cw_i4_tall", 12, 700, 24, 4, G.CHANNELWISE, "std"),
      ("oscar_b32_i4_big", 13, 300, 1024, 4, G.BLOCKWISE_32, "std"),
      ("oscar_cw_i2", 14, 16, 64, 2, G.CHANNELWISE, "std"),
      ("oscar_tw_i8_nomu2", 15, 9, 33, 8, G.TENSORWISE, "nomu2"),
  ]
  oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max = bf16_aware
  try:
    for name, seed, n, d, bits, gran, kind in specs:
      w, mu2 = problem(1000 + seed, n, d, kind)
      cfg = cfg_of(bits, True, gran)
      qsv = None if mu2 is None else {"mu2": mu2}
      res = oscar.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, w, qsv)
      arr[f"{name}/w"] = w
      if mu2 is not None:
        arr[f"{name}/mu2"] = mu2
        block = uqt.extract_block_size_from_granularity(gran) if uqt.is_blockwise(gran) else 0
        s, gain = oscar._compute_channel_scales(np.asarray(w, np.float64), np.asarray(mu2, np.float64), block)
        if s is not None:
          arr[f"{name}/s"] = s
      mult = res.custom_algorithm_param["multiplier"]
      s64 = 1.0 if mu2 is None else (np.ones(d) if f"{name}/s" not in arr else arr[f"{name}/s"])
      arr[f"{name}/bounds"] = oscar.get_clip_bounds(
          OPN.FULLY_CONNECTED, np.asarray(w, np.float64) * s64,
          None if mu2 is None else np.asarray(mu2, np.float64) / (s64 * s64), bits, gran)
      arr[f"{name}/scale"] = plain(res.scale)
      arr[f"{name}/zero_point"] = plain(res.zero_point)
      arr[f"{name}/q"] = plain(res.quantized_data)
      arr[f"{name}/multiplier"] = plain(mult)
      cases.append(dict(name=name, num_bits=bits, granularity=gran.name, has_mu2=mu2 is not None,
                        scaled=f"{name}/s" in arr, quantized_dimension=res.quantized_dimension,
                        block_size=res.block_size, scale_dtype=str(np.asarray(res.scale).dtype)))
  finally:
    oscar.uniform_quantize_tensor.tensor_zp_scale_from_min_max = real

  # calibration statistic + its merge rule
  rng = np.random.default_rng(1100)
  xs = [(rng.standard_normal((b, 5, 24)) * (1 + i)).astype(np.float32) for i, b in enumerate([1, 3, 2, 1])]
  qsv = None
  for i, x in enumerate(xs):
    arr[f"oscar_calib/x{i}"] = x
    xx = np.asarray(x, np.float64).reshape([-1, x.shape[-1]])
    new = common_quantize.get_activation_min_max(x, valid_float_range_min=-3e38, valid_float_range_max=3e38)
    new["num_samples"] = np.array(x.shape[0])
    new["mu2"] = np.mean(xx * xx, axis=0)       # oscar.calibrate's statistic (oscar.py:318-324)
    arr[f"oscar_calib/mu2_{i}"] = new["mu2"]
    qsv = qsv_utils.oscar_and_moving_average_update(qsv, new)
    arr[f"oscar_calib/merged_mu2_{i}"] = np.asarray(qsv["mu2"])
    arr[f"oscar_calib/merged_min_{i}"] = np.asarray(qsv["min"])
  cases.append(dict(name="oscar_calib", steps=len(xs), num_samples=int(qsv["num_samples"])))

  np.savez_compressed(os.path.join(GOLDEN, "ref_oscar_cases.npz"), **arr)
  with open(os.path.join(GOLDEN, "ref_oscar_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_golden.py --oscar", numpy=np.__version__,
                   reference_version=open("/root/reference/VERSION").read().strip(), cases=cases),
              f, indent=1, sort_keys=True)
  print(f"wrote {len(arr)} OSCAR arrays, {len(cases)} cases")


def dwr_cases():
  """dequantized_weight_recovery on seeded fake-quantized weights -> ref_dwr_cases.npz/json."""
  from ai_edge_quantizer.algorithms.uniform_quantize import dequantized_weight_recovery as dwr
  arr, cases = {}, []
  specs = [
      ("dwr_cw_i4", 0, (24, 96), 4, G.CHANNELWISE, "FULLY_CONNECTED"),
      ("dwr_cw_i8", 1, (16, 300), 8, G.CHANNELWISE, "FULLY_CONNECTED"),
      ("dwr_b32_i4", 2, (12, 128), 4, G.BLOCKWISE_32, "FULLY_CONNECTED"),
      ("dwr_b128_i4", 3, (6, 512), 4, G.BLOCKWISE_128, "FULLY_CONNECTED"),
      ("dwr_tw_i8", 4, (10, 40), 8, G.TENSORWISE, "FULLY_CONNECTED"),
      ("dwr_conv_cw_i8", 5, (8, 3, 3, 5), 8, G.CHANNELWISE, "CONV_2D"),
      ("dwr_emb_cw_i4", 6, (50, 64), 4, G.CHANNELWISE, "EMBEDDING_LOOKUP"),
      ("dwr_cw_i4_sparse", 7, (20, 64), 4, G.CHANNELWISE, "FULLY_CONNECTED"),
      ("dwr_cw_i4_wide", 8, (3, 5000), 4, G.CHANNELWISE, "FULLY_CONNECTED"),
      ("dwr_tw_i4_tiny_scale", 9, (8, 16), 4, G.TENSORWISE, "FULLY_CONNECTED"),
  ]
  for name, seed, shape, bits, gran, op in specs:
    rng = np.random.default_rng(2000 + seed)
    qmax = 2 ** (bits - 1) - 1
    q = rng.integers(-qmax, qmax + 1, size=shape).astype(np.int8)
    if "sparse" in name:
      q[rng.random(shape) < 0.7] = 0
      q[3] = 0                                   # an all-zero row: scale falls back to 1e-9
    if gran == G.TENSORWISE:
      scale = np.float32(3e-10 if "tiny" in name else 0.0123)
      w = (q.astype(np.float32) * scale).astype(np.float32)
    elif uqt.is_blockwise(gran):
      b = uqt.extract_block_size_from_granularity(gran)
      sc = (rng.random((shape[0], shape[1] // b)).astype(np.float32) * 0.05 + 0.001)
      w = (q.reshape(shape[0], -1, b).astype(np.float32) * sc[:, :, None]).reshape(shape).astype(np.float32)
    else:
      sc = (rng.random((shape[0],) + (1,) * (len(shape) - 1)).astype(np.float32) * 0.05 + 0.001)
      w = (q.astype(np.float32) * sc).astype(np.float32)
    cfg = cfg_of(bits, True, gran)
    info = op_info(OPN[op], cfg)
    arr[f"{name}/w"] = w
    try:
      res = dwr.get_tensor_quant_params(info, cfg, as_input(w, gran))
    except Exception as e:
      cases.append(dict(name=name, num_bits=bits, granularity=gran.name, op=op, error=type(e).__name__,
                        message=str(e)[:200]))
      print("err", name, str(e)[:150])
      continue
    arr[f"{name}/scale"] = plain(res.scale)
    arr[f"{name}/zero_point"] = plain(res.zero_point)
    arr[f"{name}/q"] = plain(res.quantized_data)
    cases.append(dict(name=name, num_bits=bits, granularity=gran.name, op=op,
                      quantized_dimension=res.quantized_dimension, block_size=res.block_size,
                      scale_dtype=str(np.asarray(res.scale).dtype), recovered=bool(np.array_equal(res.quantized_data, q))))
    print("ok ", name, res.scale.dtype, res.scale.shape, cases[-1]["recovered"])
  # not fake-quantized at all: the reference refuses
  w = np.random.default_rng(2099).standard_normal((16, 64)).astype(np.float32)
  arr["dwr_random/w"] = w
  cfg = cfg_of(4, True, G.CHANNELWISE)
  try:
    dwr.get_tensor_quant_params(op_info(OPN.FULLY_CONNECTED, cfg), cfg, w)
    raise SystemExit("expected a failure")
  except RuntimeError as e:
    cases.append(dict(name="dwr_random", num_bits=4, granularity="CHANNELWISE", op="FULLY_CONNECTED",
                      error="RuntimeError", message=str(e)[:400]))
  np.savez_compressed(os.path.join(GOLDEN, "ref_dwr_cases.npz"), **arr)
  with open(os.path.join(GOLDEN, "ref_dwr_cases.json"), "w") as f:
    json.dump(dict(generator="tests/golden/gen/make_golden.py --dwr", numpy=np.__version__, cases=cases), f,
              indent=1, sort_keys=True)
  print(f"wrote {len(arr)} DWR arrays, {len(cases)} cases")


def main():
  if "--oscar" in sys.argv:
    oscar_cases()
    return
  if "--dwr" in sys.argv:
    dwr_cases()
    return
  # cross-check the bf16 stand-in against an independent implementation
  import torch
  probe = np.random.default_rng(5).standard_normal(200000).astype(np.float32)
  probe = np.concatenate([probe, probe * 1e-30, probe * 1e30,
                          np.array([0.0, -0.0, 65280.0, 1e-9 / 7], np.float32)])
  via_torch = torch.from_numpy(probe).to(torch.bfloat16).to(torch.float32).numpy()
  assert np.array_equal(ml_dtypes.round_to_bf16(probe), via_torch), "bf16 RNE mismatch"

  minmax_cases()
  octav_cases()
  mse_cases()
  hadamard_cases()
  gptq_cases()
  activation_cases()
  pack_cases()
  direct_cases()
  digest_cases()
  c1_cases()

  np.savez_compressed(os.path.join(GOLDEN, "ref_cases.npz"), **ARR)
  meta = dict(generator="tests/golden/gen/make_golden.py",
              reference_version=open("/root/reference/VERSION").read().strip(),
              numpy=np.__version__, cases=CASES)
  with open(os.path.join(GOLDEN, "ref_cases.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
  with open(os.path.join(GOLDEN, "ref_digests.json"), "w") as f:
    json.dump(dict(generator=meta["generator"], numpy=np.__version__,
                   digest="sha256(ndarray.tobytes())", cases=DIGESTS),
              f, indent=1, sort_keys=True)
  print(f"wrote {len(ARR)} arrays, {len(CASES)} cases, {len(DIGESTS)} digests")


if __name__ == "__main__":
  main()
