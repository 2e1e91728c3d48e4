"""CPU oracle for the AEQ calibration + requantization hot path.

TEST INFRASTRUCTURE ONLY. This module is a NumPy restatement of the arithmetic
in the reference (google-ai-edge/ai-edge-quantizer @ /root/reference, v0.10.0).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and only as the checker / the timed CPU baseline. The product path
(`ai-edge-quantizer_amd/mi355q`) never imports it and has no CPU fallback.

Parity pin: every function below is checked against
  * the known-answer vectors of the reference's own unit tests, transcribed as
    data in tests/golden/ref_known_answers.json, and
  * outputs of the real reference imported in the build container
    (tests/golden/gen/make_golden.py -> tests/golden/*.npz, *.json),
by tests/test_oracle_golden.py.

All `ref:` citations are relative to /root/reference/ai_edge_quantizer/.
Dtype/promotion behaviour follows NumPy >= 2 (weak Python scalars).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------

TENSORWISE = "TENSORWISE"
CHANNELWISE = "CHANNELWISE"


def is_blockwise(granularity: str) -> bool:
  """ref: algorithms/uniform_quantize/uniform_quantize_tensor.py:32-34."""
  return "BLOCKWISE" in str(granularity)


def block_size_of(granularity: str) -> int:
  """ref: uniform_quantize_tensor.py:48-61 (32/64/128/256, else 0)."""
  g = str(granularity).split(".")[-1]
  if g.startswith("BLOCKWISE_"):
    n = int(g.split("_")[1])
    return n if n in (32, 64, 128, 256) else 0
  return 0


def qrange(num_bits: int, signed: bool = True) -> tuple[float, float]:
  """ref: uniform_quantize_tensor.py:37-45."""
  if signed:
    return float(-(2 ** (num_bits - 1))), float(2 ** (num_bits - 1) - 1)
  return 0.0, float(2**num_bits - 1)


def int_dtype(num_bits: int, signed: bool = True):
  """ref: uniform_quantize_tensor.py:88-109."""
  for limit, s, u in ((8, np.int8, np.uint8), (16, np.int16, np.uint16),
                      (32, np.int32, np.uint32)):
    if num_bits <= limit:
      return s if signed else u
  return np.int64 if signed else np.uint64


def round_to_bf16(x: np.ndarray) -> np.ndarray:
  """float32 -> bfloat16 (round-to-nearest-even), widened back to float32.

  Stands in for `x.astype(ml_dtypes.bfloat16)` (ref: uniform_quantize_tensor.py:580,
  transformations/quantize_tensor.py:131). ml_dtypes is a third-party dep that is
  not vendored in the reference; its published conversion is IEEE RNE, which is
  what this integer formulation implements (cross-checked with torch.bfloat16).
  """
  x = np.ascontiguousarray(x, dtype=np.float32)
  bits = x.view(np.uint32)
  lsb = (bits >> np.uint32(16)) & np.uint32(1)
  out = ((bits + np.uint32(0x7FFF) + lsb) & np.uint32(0xFFFF0000)).view(np.float32)
  nan = np.isnan(x)
  if nan.any():
    out = out.copy()
    out[nan] = np.nan
  return out


def blockwise_scale_round(scale: np.ndarray) -> np.ndarray:
  """f32 -> bf16 -> f16 -> f32. ref: uniform_quantize_tensor.py:577-581."""
  return round_to_bf16(scale).astype(np.float16).astype(np.float32)


def blockwise_scale_f16(scale: np.ndarray) -> np.ndarray:
  """Scale as stored in the `<name>_scales` tensor. ref: quantize_tensor.py:129-137."""
  return round_to_bf16(np.asarray(scale, dtype=np.float32)).astype(np.float16)


def reduce_dims_for(quantized_dim, shape):
  """ref: algorithms/utils/common_utils.py:1196-1207."""
  if quantized_dim is None:
    return None
  return tuple(d for d in range(len(shape)) if d != quantized_dim)


def split_blocks(shape, quantized_dim: int, block: int) -> list[int]:
  """ref: uniform_quantize_tensor.py:164-194."""
  out = []
  for i, v in enumerate(shape):
    if i == quantized_dim:
      if v % block != 0:
        raise ValueError(
            f"Quantized dimension {v} in tensor shape {tuple(shape)} is not"
            f" divisible by block size {block}.")
      out += [v // block, block]
    else:
      out.append(v)
  return out


# --------------------------------------------------------------------------
# a1: weight min/max
# --------------------------------------------------------------------------

def init_tensor_min_max(w: np.ndarray, granularity: str, quantized_dim=None) -> dict:
  """ref: algorithms/uniform_quantize/common_quantize.py:1311-1359.

  `quantized_dim` is what common_utils.get_weight_quantized_dim returns for the
  op (FC/EMBEDDING: 0 channelwise, 1 blockwise; utils/tfl_flatbuffer_utils.py:95-106).
  """
  if granularity == TENSORWISE or str(granularity).endswith("TENSORWISE"):
    return {"min": np.min(w, axis=None, keepdims=True),
            "max": np.max(w, axis=None, keepdims=True)}
  if is_blockwise(granularity):
    b = block_size_of(granularity)
    r = w.reshape(split_blocks(w.shape, quantized_dim, b))
    return {"min": np.min(r, axis=quantized_dim + 1),
            "max": np.max(r, axis=quantized_dim + 1)}
  dims = reduce_dims_for(quantized_dim, w.shape)
  return {"min": np.min(w, axis=dims, keepdims=True),
          "max": np.max(w, axis=dims, keepdims=True)}


# --------------------------------------------------------------------------
# a2: zero point + scale
# --------------------------------------------------------------------------

def zp_scale_from_min_max(min_value, max_value, num_bits: int, symmetric: bool,
                          granularity: str, clipping_values=None):
  """ref: uniform_quantize_tensor.py:492-586 (signed types only, as there)."""
  qmin, qmax = qrange(num_bits, True)
  floor = 1e-9
  pos_clip = clipping_values
  neg_clip = None if clipping_values is None else -clipping_values
  blockwise = is_blockwise(granularity)
  if blockwise:
    hi = np.broadcast_to(np.array(65280) * (2**num_bits - 1), np.shape(max_value))
    lo = np.broadcast_to(np.array(-65280) * (2**num_bits), np.shape(min_value))
    pos_clip = hi if pos_clip is None else np.minimum(pos_clip, hi)
    neg_clip = lo if neg_clip is None else np.maximum(neg_clip, lo)
  if symmetric:
    bound = np.maximum(np.abs(min_value), np.abs(max_value))
    bound = np.maximum(bound, floor)
    if clipping_values is not None:
      bound = np.clip(bound, neg_clip, pos_clip)
    scale = bound / qmax
    zp = np.zeros_like(scale, dtype=np.int32)
  else:
    bmax = np.maximum(max_value, np.zeros_like(max_value))
    bmin = np.minimum(min_value, np.zeros_like(min_value))
    bound = np.maximum(bmax - bmin, floor)
    if clipping_values is not None:
      bound = np.clip(bound, -clipping_values, clipping_values)
    scale = bound / (qmax - qmin)
    zp = np.rint(qmin - bmin / scale)
  if blockwise:
    scale = blockwise_scale_round(scale)
  zp = np.asarray(zp).astype(int_dtype(num_bits, True), copy=False)
  return zp, scale


# --------------------------------------------------------------------------
# a3: uniform quantize / dequantize
# --------------------------------------------------------------------------

def _expand_blockwise(shape, scale, zp, quantized_dim: int, block: int):
  """ref: uniform_quantize_tensor.py:222-270."""
  full = split_blocks(shape, quantized_dim, block)
  s = np.reshape(np.broadcast_to(np.expand_dims(scale, quantized_dim + 1), full), shape)
  if zp is None or np.size(zp) == 0:
    z = np.zeros(shape, dtype=np.int32)
  else:
    z = np.reshape(np.broadcast_to(np.expand_dims(zp, quantized_dim + 1), full), shape)
  return s, z


def _fix_rank(x: np.ndarray, scale, zp, quantized_dim):
  """ref: uniform_quantize_tensor.py:112-161."""
  scale = np.asarray(scale)
  zp = np.asarray(zp)
  if x.ndim == scale.ndim:
    return scale, zp
  if x.ndim == 0:
    if scale.size != 1 or zp.size != 1:
      raise ValueError(
          "Scale and zero_point must contain single element for scalar tensor."
          f" Got scale: {scale}, zero_point: {zp}")
    return np.array(scale.item()), np.array(zp.item())
  dims = [d for d in range(x.ndim) if d != quantized_dim]
  return np.expand_dims(scale, axis=dims), np.expand_dims(zp, axis=dims)


def _validate(x, scale, zp, quantized_dim, block):
  """ref: uniform_quantize_tensor.py:589-638."""
  if scale.shape != zp.shape and zp.size != 1:
    raise ValueError(
        "scale and zero_point must have the same shape or zero_point must have"
        f" only one element. Got {scale.shape} and {zp.shape}")
  if x.ndim != scale.ndim or x.ndim != zp.ndim:
    raise ValueError(
        f"Ranks of scales ({scale.ndim}) and zps ({zp.ndim}) must be the same as"
        f" the tensor rank ({x.ndim}).")
  if block != 0 and x.shape[quantized_dim] % block != 0:
    raise ValueError(
        "Tensor dimension must be divisible by block size. Got dimension:"
        f" {x.shape[quantized_dim]} and block size: {block}")


def _round_clip(q: np.ndarray, lo: float, hi: float) -> np.ndarray:
  """ref: uniform_quantize_tensor.py:64-85 (in place on arrays)."""
  if np.isscalar(q):
    return np.clip(np.rint(q), lo, hi)
  return np.clip(np.rint(q, out=q), lo, hi, out=q)


def uniform_quantize(x: np.ndarray, scale, zp, num_bits: int, symmetric: bool,
                     quantized_dim=None, block_size: int = 0,
                     is_blockwise_quant: bool = False) -> np.ndarray:
  """q = cast(clip(rint(x / scale + zp))). ref: uniform_quantize_tensor.py:273-362.

  Includes the >32 MiB row-chunk loop (ref :323-354) exactly as the reference
  iterates it (range over the *element* count, step = rows per 32 MiB), because
  bench.py times this function as the CPU baseline.
  """
  x = np.asarray(x)
  scale = np.asarray(scale)
  zp = np.asarray(zp)
  if is_blockwise_quant:
    if quantized_dim is None:
      raise ValueError("Quantized dimension must be specified.")
    if not block_size or block_size <= 0:
      raise ValueError("Block size must be specified and positive.")
    scale, zp = _expand_blockwise(x.shape, scale, zp, quantized_dim, block_size)
  scale, zp = _fix_rank(x, scale, zp, quantized_dim)
  _validate(x, scale, zp, quantized_dim, block_size)
  if not np.issubdtype(zp.dtype, np.signedinteger):
    raise ValueError(
        f"zero_points need to be {np.signedinteger}. But the actual type is"
        f" {zp.dtype}.")
  qmin, qmax = qrange(num_bits, True)
  lo = qmin + 1 if (symmetric and num_bits >= 8) else qmin
  out_dtype = int_dtype(num_bits, True)
  if (x.ndim > 1 and x.nbytes > 32 * 1024 * 1024
      and x.dtype.itemsize * 8 != num_bits):
    shape = x.shape
    x2 = x.reshape([-1, shape[-1]])
    s2 = np.broadcast_to(scale, x2.shape)   # rank > 2 raises here, as in the reference
    z2 = np.broadcast_to(zp, x2.shape)
    ret = np.zeros(x2.shape, dtype=out_dtype)
    rows_per_chunk = (32 * 1024 * 1024) // (shape[-1] * x2.dtype.itemsize)
    for k in range(0, x2.size, rows_per_chunk):
      end = min(k + rows_per_chunk, x2.size)
      q = np.divide(x2[k:end, :], s2[k:end, :])
      q = np.add(q, z2[k:end, :], out=q)
      q = _round_clip(q, lo, qmax)
      ret[k:end, :] = q.astype(out_dtype, copy=False)
    return ret.reshape(shape)
  q = np.divide(x, scale)
  q = np.add(q, zp, out=None if np.isscalar(q) else q)
  q = _round_clip(q, lo, qmax)
  return np.asarray(q).astype(out_dtype, copy=False)


def uniform_dequantize(q: np.ndarray, scale, zp, quantized_dim=None,
                       block_size: int = 0) -> np.ndarray:
  """(q - zp) * scale. ref: uniform_quantize_tensor.py:365-409."""
  q = np.asarray(q)
  scale = np.asarray(scale)
  zp = np.asarray(zp)
  if block_size != 0:
    if quantized_dim == 0:  # XNNPack-style dim -> AEQ-style (ref :383-387)
      quantized_dim = 1
    sshape = list(q.shape)
    sshape[quantized_dim] //= block_size
    scale = scale.reshape(sshape)
    scale, zp = _expand_blockwise(q.shape, scale, zp, quantized_dim, block_size)
  scale, zp = _fix_rank(q, scale, zp, quantized_dim)
  _validate(q, scale, zp, quantized_dim, block_size)
  return np.multiply(q - zp, scale)


def quantize_bias(bias: np.ndarray, in_scale, w_scale, in_num_bits: int = 8):
  """int32 (int64 for 16-bit activations) symmetric bias.

  ref: uniform_quantize_tensor.py:412-489. Returns (q, scale, zp, num_bits,
  quantized_dimension).
  """
  eff = np.squeeze(np.asarray(in_scale) * np.asarray(w_scale))
  if not eff.shape:
    eff = np.expand_dims(eff, axis=0)
  zp = np.zeros_like(eff, dtype=np.int32)
  qdim = None if len(eff) == 1 else 0
  q = uniform_quantize(bias, eff, zp, 32, True, quantized_dim=qdim)
  bits = 32
  if in_num_bits == 16:
    q = q.astype(np.int64)
    bits = 64
  return q, eff, zp, bits, qdim


# --------------------------------------------------------------------------
# a4: min/max algorithm
# --------------------------------------------------------------------------

def weight_quantized_dim(granularity: str, op: str = "FULLY_CONNECTED", rank: int = 2,
                         adj_y: bool = False):
  """ref: common_utils.py:1162-1193 + tfl_flatbuffer_utils.py:95-106; BATCH_MATMUL's
  right-hand side uses its last axis (second to last when adj_y), ref common_utils.py:1143-1159."""
  cw = {"FULLY_CONNECTED": 0, "DEPTHWISE_CONV_2D": 3, "CONV_2D": 0,
        "EMBEDDING_LOOKUP": 0, "CONV_2D_TRANSPOSE": 0}
  bw = {"FULLY_CONNECTED": 1, "EMBEDDING_LOOKUP": 1}
  if str(granularity).endswith(CHANNELWISE):
    if op == "BATCH_MATMUL":
      return rank - 2 if adj_y else rank - 1
    return cw.get(op)
  if is_blockwise(granularity):
    return bw[op]
  return None


def min_max_quant_params(w, num_bits: int, symmetric: bool, granularity: str,
                         op: str = "FULLY_CONNECTED", qsv=None) -> dict:
  """ref: algorithms/uniform_quantize/naive_min_max_quantize.py:34-110."""
  qdim = weight_quantized_dim(granularity, op, np.ndim(w))
  if qsv is None or "min" not in qsv:
    if w is None:
      raise ValueError("not found in tensor_name_to_qsv")
    mm = init_tensor_min_max(w, granularity, qdim)
  else:
    mm = qsv
  zp, scale = zp_scale_from_min_max(mm["min"], mm["max"], num_bits, symmetric,
                                    granularity, None)
  out = dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=symmetric,
             quantized_dimension=qdim, block_size=block_size_of(granularity),
             quantized_data=None)
  if w is not None:
    out["quantized_data"] = uniform_quantize(
        w, scale, zp, num_bits, symmetric, quantized_dim=qdim,
        block_size=out["block_size"], is_blockwise_quant=is_blockwise(granularity))
  return out


# --------------------------------------------------------------------------
# a5: bit packing
# --------------------------------------------------------------------------

def pack_data(bitwidth: int, data: np.ndarray) -> np.ndarray:
  """Low-bits-first int4 / int2 packing. ref: transformations/transformation_utils.py:293-353."""
  data = np.asarray(data).reshape(-1)
  if bitwidth not in (2, 4):
    return data
  per = 8 // bitwidth
  mask = (1 << bitwidth) - 1
  n_out = -(-data.size // per)
  out = np.zeros(n_out, dtype=np.uint8)
  for slot in range(per):
    lane = (data[slot::per].astype(np.uint8)) & np.uint8(mask)
    out[: lane.size] |= (lane << np.uint8(slot * bitwidth)).astype(np.uint8)
  return out


# --------------------------------------------------------------------------
# a6: OCTAV
# --------------------------------------------------------------------------

def octav_clip(x: np.ndarray, bits: int, axis, max_iterations: int = 10,
               exponent_divisor: float = 3.0, early_stop: bool = True,
               return_iters: bool = False):
  """Newton iteration for the clipping constant. ref: algorithms/uniform_quantize/octav.py:30-112."""
  if axis is not None:
    axis = (axis,) if isinstance(axis, int) else tuple(axis)
    reduced = tuple(1 if k in axis else d for k, d in enumerate(x.shape))
    count = np.prod([x.shape[d] for d in axis])     # np.int64 -> s*count is f64
  else:
    reduced = (1,)
    count = x.size                                  # python int -> stays f32
  guess = np.ones(reduced, dtype=np.float32)
  mask = np.zeros(x.shape, dtype=bool)
  s = np.asarray(4.0 ** (-bits) / exponent_divisor, dtype=np.float32)
  iters = 0
  for _ in range(max_iterations):
    iters += 1
    old = guess
    mask = np.greater_equal(x, old, out=mask)
    denom = np.count_nonzero(mask, axis=axis, keepdims=True).astype(np.float32)
    guess = np.sum(x, axis=axis, where=mask, keepdims=True, dtype=guess.dtype)
    mask = np.less_equal(x, -old, out=mask)
    denom = np.add(denom, np.count_nonzero(mask, axis=axis, keepdims=True), out=denom)
    guess = np.subtract(guess, np.sum(x, axis=axis, where=mask, keepdims=True),
                        out=guess)
    denom = np.multiply(denom, 1.0 - s, out=denom)
    denom = np.add(denom, s * count, out=denom)
    guess = np.divide(guess, denom, out=guess)
    if early_stop and np.allclose(old, guess):
      break
  return (guess, iters) if return_iters else guess


def octav_quant_params(w, num_bits: int, granularity: str,
                       op: str = "FULLY_CONNECTED", symmetric: bool = True, adj_y: bool = False) -> dict:
  """ref: octav.py:115-227 (weights; min/max computed on the spot)."""
  if not symmetric:
    raise ValueError(f"Unsupported symmetry: {symmetric}. OCTAV supports symmetric"
                     " quantization only for now.")
  qdim = weight_quantized_dim(granularity, op, np.ndim(w), adj_y)
  mm = init_tensor_min_max(w, granularity, qdim)
  if is_blockwise(granularity):
    b = block_size_of(granularity)
    data, axes = w.reshape(split_blocks(w.shape, qdim, b)), qdim + 1
  else:
    data, axes = w, reduce_dims_for(qdim, w.shape)
  clip = octav_clip(data, num_bits, axes, max_iterations=10, exponent_divisor=3.0)
  if is_blockwise(granularity):
    clip = clip.reshape(mm["min"].shape)
  zp, scale = zp_scale_from_min_max(mm["min"], mm["max"], num_bits, True,
                                    granularity, clip)
  bs = block_size_of(granularity)
  q = uniform_quantize(w, scale, zp, num_bits, True, quantized_dim=qdim,
                       block_size=bs, is_blockwise_quant=is_blockwise(granularity))
  return dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=True,
              quantized_dimension=qdim, block_size=bs, quantized_data=q, clip=clip)


# --------------------------------------------------------------------------
# a7: Hadamard rotation
# --------------------------------------------------------------------------

def hadamard_matrix(size: int) -> np.ndarray:
  """Sylvester H_size / sqrt(size) in float32. ref: hadamard_rotation.py:48-90."""
  size = int(size)
  if size <= 0 or size & (size - 1):
    raise ValueError("Hadamard matrix size must be a power of 2. ")
  h2 = np.array([[1, 1], [1, -1]], dtype=np.int8)
  h = h2
  n = 2
  while n < size:
    h = np.kron(h, h2)
    n *= 2
  return h / np.sqrt(n, dtype=np.float32)


def hadamard_size_for(last_dim: int, max_size=None) -> int:
  """ref: hadamard_rotation.py:118-123."""
  h = int(np.gcd(last_dim, 2**30))
  if max_size:
    h = min(h, 1 << (int(max_size).bit_length() - 1))
  return h


def hadamard_rotate(w: np.ndarray, max_size=None):
  """W.reshape(-1, h) @ H_h. ref: hadamard_rotation.py:93-134.

  Note the reference's size-1 corner: _make_hadamard_matrix(1) returns the 2x2
  matrix (loop never runs); reshape(-1, 1) @ 2x2 then fails in matmul. Not
  reproduced: odd last dims are rejected by the caller here as there (error).
  """
  h = hadamard_size_for(w.shape[-1], max_size)
  hm = hadamard_matrix(h) if h >= 2 else hadamard_matrix(2)
  rot = np.matmul(w.reshape((-1, h)), hm)
  return rot.reshape(w.shape), h


def hadamard_quant_params(w, num_bits: int, granularity: str,
                          op: str = "FULLY_CONNECTED", max_size=None) -> dict:
  """rotate -> OCTAV. ref: hadamard_rotation.py:137-203."""
  if w is None:
    raise ValueError("Hadamard rotation is only supported for weight tensors.")
  if w.ndim < 2:
    raise ValueError("Hadamard rotation is only supported for tensors with rank >= 2.")
  rot, h = hadamard_rotate(w, max_size)
  out = octav_quant_params(rot, num_bits, granularity, op)
  out["hadamard_size"] = h
  out["random_binary_vector"] = np.ones(h, dtype=np.int8)
  out["rotated"] = rot
  return out


# --------------------------------------------------------------------------
# a8: activation statistics
# --------------------------------------------------------------------------

def activation_min_max(x: np.ndarray, lo=None, hi=None) -> dict:
  """ref: common_quantize.py:1362-1413 (strict range mask with all-masked fallback)."""
  shape = (1,) * x.ndim
  if np.issubdtype(x.dtype, np.integer):
    t_min, t_max = np.min(x), np.max(x)
  else:
    if lo is not None:
      t_min = np.min(x, where=x > lo, initial=np.inf, axis=None)
      if t_min == np.inf:
        t_min = np.min(x)
    else:
      t_min = np.min(x)
    if hi is not None:
      t_max = np.max(x, where=x < hi, initial=-np.inf, axis=None)
      if t_max == -np.inf:
        t_max = np.max(x)
    else:
      t_max = np.max(x)
  return {"min": np.reshape(t_min, shape), "max": np.reshape(t_max, shape)}


def activation_qsv(x: np.ndarray, valid_range=(-3e38, 3e38)) -> dict:
  """min/max + num_samples. ref: common_quantize.py:1416-1456,
  naive_min_max_quantize.py:181-226 (default valid_range)."""
  q = activation_min_max(x, valid_range[0], valid_range[1])
  q["num_samples"] = np.array(x.shape[0] if x.ndim > 0 else 1)
  return q


# --------------------------------------------------------------------------
# a9: QSV merge rules
# --------------------------------------------------------------------------

def moving_average_update(qsv, new_qsv, smoothing_factor: float = 0.95):
  """ref: utils/qsv_utils.py:25-68."""
  if not qsv:
    return new_qsv
  f = smoothing_factor
  return {"min": f * qsv["min"] + (1.0 - f) * new_qsv["min"],
          "max": f * qsv["max"] + (1.0 - f) * new_qsv["max"]}


def min_max_update(qsv, new_qsv):
  """ref: utils/qsv_utils.py:105-122."""
  if not qsv:
    return new_qsv
  return {"min": np.minimum(qsv["min"], new_qsv["min"]),
          "max": np.maximum(qsv["max"], new_qsv["max"])}


def gptq_and_moving_average_update(qsv, new_qsv):
  """ref: utils/qsv_utils.py:71-102."""
  if not qsv:
    return new_qsv
  out = moving_average_update(qsv, new_qsv)
  n0, n1 = qsv["num_samples"], new_qsv["num_samples"]
  total = n0 + n1
  if total == 0:
    out["hessian"], out["num_samples"] = new_qsv["hessian"], 0
  else:
    out["hessian"] = (qsv["hessian"] * n0 + new_qsv["hessian"] * n1) / total
    out["num_samples"] = total
  return out


def replay_moving_average(mins, maxs, smoothing_factor: float = 0.95):
  """Apply moving_average_update over per-sample (min, max) in dataset order.

  ref: calibrator.py:395-421 (one update per sample per tensor). `mins`/`maxs`
  are sequences of per-sample arrays; the first sample is taken as is.
  """
  qsv = None
  for mn, mx in zip(mins, maxs):
    qsv = moving_average_update(qsv, {"min": mn, "max": mx}, smoothing_factor)
  return qsv


# --------------------------------------------------------------------------
# a10-a12: GPTQ
# --------------------------------------------------------------------------

def gptq_hessian(x: np.ndarray) -> np.ndarray:
  """H = (2 / num_samples) * X^T X, num_samples = x.shape[0]. ref: gptq.py:100-107."""
  n = np.array(x.shape[0] if x.ndim > 0 else 1)
  x2 = x.reshape([-1, x.shape[-1]])
  return (2.0 / n) * x2.T.dot(x2)


def gptq_hessian_inverse(hessian: np.ndarray, damp_factor: float = 0.01,
                         product: str = "einsum") -> np.ndarray:
  """damp -> Cholesky -> triangular inverse -> L^-T L^-1. ref: gptq.py:111-128.

  product="matmul" forms the final product with sgemm instead of the reference's two-operand
  einsum (NumPy's own loops, no BLAS: 2.3 s at d = 2048, hours at d = 16384). Same float32
  products, another addition order -- for checks at sizes the einsum cannot reach, and as one of
  the reference-side reorderings the T2 noise floor is measured with."""
  import scipy.linalg
  hessian = np.array(hessian, copy=True)
  d0 = np.diag(hessian)
  d = np.where(d0, d0, 1.0)
  d = d + damp_factor * np.mean(d)
  np.fill_diagonal(hessian, d)
  l = np.linalg.cholesky(hessian)   # in the Hessian's dtype (f64 when it comes
  # from calibrate(): `(2.0 / np.array(n)) * f32` promotes to float64)
  # The reference always calls the *single precision* LAPACK routine; f2py
  # casts a float64 factor to float32 first, so L^-1 and H^-1 are float32.
  linv, err = scipy.linalg.lapack.strtri(l, lower=True, overwrite_c=True)
  assert err == 0
  if product == "matmul":
    return np.matmul(linv.T, linv)
  return np.einsum("ji,jk->ik", linv, linv)


def gptq_apply(w: np.ndarray, scale, zp, num_bits: int, symmetric: bool,
               hessian: np.ndarray, granularity: str, block_size: int = 0,
               blocksize: int = 64, hinv=None, product: str = "einsum") -> np.ndarray:
  """Blocked OBS update + column-serial quantization. ref: gptq.py:131-216.

  `scale`/`zp` are the up-front min/max parameters (shape [rows,1] channelwise,
  [1,1] tensorwise, [rows, cols/block] blockwise). `hinv` may be passed to pin
  the inverse (T1-style parity of the apply step alone).
  """
  fw = np.array(w, copy=True)
  qdt = int_dtype(num_bits, True)
  qw = np.zeros(fw.shape, dtype=qdt)
  if hinv is None:
    hinv = gptq_hessian_inverse(hessian, product=product)
  ncols = hinv.shape[0]
  blockwise = is_blockwise(granularity)
  cw = str(granularity).endswith(CHANNELWISE)
  for b0 in range(0, ncols, blocksize):
    b1 = min(b0 + blocksize, ncols)
    wb = fw[:, b0:b1]
    qb = np.zeros(wb.shape, dtype=qdt)
    eb = np.zeros_like(wb)
    for i in range(b1 - b0):
      c = b0 + i
      col = wb[:, i]
      if blockwise:
        s_c, z_c, qd = scale[:, c // block_size], zp[:, c // block_size], 0
      else:
        s_c, z_c, qd = scale, zp, (0 if cw else None)
      q = uniform_quantize(np.expand_dims(col, -1), s_c, z_c, num_bits, symmetric,
                           quantized_dim=qd).reshape(-1, 1)
      dq = uniform_dequantize(q, s_c, z_c, quantized_dim=qd).reshape(-1)
      qb[:, i] = q.reshape(-1)
      np.subtract(col, dq, out=eb[:, i])
      eb[:, i] /= hinv[c, c]
      if i < b1 - b0 - 1:
        wb[:, i + 1:] -= np.outer(eb[:, i], hinv[c, c + 1:b1])
    qw[:, b0:b1] = qb
    fw[:, b1:] -= np.matmul(eb, hinv[b0:b1, b1:])
  return qw


def gptq_quant_params(w, num_bits: int, symmetric: bool, granularity: str,
                      qsv=None, op: str = "FULLY_CONNECTED") -> dict:
  """ref: gptq.py:219-300."""
  act = qsv.get("activation_tensor_qsv") if qsv else None
  qdim = weight_quantized_dim(granularity, op, np.ndim(w))
  mm = qsv if (qsv is not None and "min" in qsv) else init_tensor_min_max(
      w, granularity, qdim)
  zp, scale = zp_scale_from_min_max(mm["min"], mm["max"], num_bits, symmetric,
                                    granularity, None)
  bs = block_size_of(granularity)
  out = dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=symmetric,
             quantized_dimension=qdim, block_size=bs, quantized_data=None)
  if w is None or act is None or "hessian" not in act:
    return out
  out["quantized_data"] = gptq_apply(w, scale, zp, num_bits, symmetric,
                                     act["hessian"], granularity, bs)
  return out


# --------------------------------------------------------------------------
# a14: MSE
# --------------------------------------------------------------------------

_MSE_MUL = {8: 0.05408, 4: 0.37755}


def mse_quant_params(w, num_bits: int, granularity: str,
                     op: str = "FULLY_CONNECTED", symmetric: bool = True, adj_y: bool = False) -> dict:
  """scale = k * sqrt(mean(x^2)). ref: algorithms/uniform_quantize/mse.py:36-128."""
  if is_blockwise(granularity):
    raise ValueError("Blockwise quantization is not supported for MSE quantization.")
  if not symmetric:
    raise ValueError(f"Unsupported symmetry: {symmetric}. MSE supports symmetric"
                     " quantization only for now.")
  qdim = weight_quantized_dim(granularity, op, np.ndim(w), adj_y)
  dims = reduce_dims_for(qdim, w.shape)
  scale = _MSE_MUL[num_bits] * np.sqrt(np.mean(w**2, axis=dims, keepdims=True))
  zp = np.zeros_like(scale, dtype=np.int32)
  q = uniform_quantize(w, scale, zp, num_bits, True, quantized_dim=qdim)
  return dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=True,
              quantized_dimension=qdim, block_size=0, quantized_data=q)


# --------------------------------------------------------------------------
# f4: OSCAR (activation-aware channel scaling + optimal clipping), all FP64
# --------------------------------------------------------------------------

_OSCAR_TINY = 1e-12
_OSCAR_S_RANGE = (1e-4, 1e4)


def oscar_mu2(x: np.ndarray) -> np.ndarray:
  """Per-channel (trailing axis) second moment of one activation sample.
  ref: algorithms/uniform_quantize/oscar.py:318-324."""
  x2 = np.asarray(x, np.float64).reshape([-1, np.shape(x)[-1]])
  return np.mean(x2 * x2, axis=0)


def oscar_and_moving_average_update(qsv, new_qsv):
  """EMA for min/max, sample-weighted mean for mu2. ref: utils/qsv_utils.py:125-171."""
  if not qsv:
    return new_qsv
  out = moving_average_update(qsv, new_qsv)
  has0, has1 = "mu2" in qsv, "mu2" in new_qsv
  if not has0 and not has1:
    out["mu2"], out["num_samples"] = None, 0
  elif not has0:
    out["mu2"], out["num_samples"] = new_qsv.get("mu2"), new_qsv.get("num_samples", 0)
  elif not has1:
    out["mu2"], out["num_samples"] = qsv.get("mu2"), qsv.get("num_samples", 0)
  else:
    n0, n1 = qsv.get("num_samples", 0), new_qsv.get("num_samples", 0)
    if n0 + n1 == 0:
      out["mu2"], out["num_samples"] = new_qsv["mu2"], 0
    else:
      out["mu2"] = (qsv["mu2"] * n0 + new_qsv["mu2"] * n1) / (n0 + n1)
      out["num_samples"] = n0 + n1
  return out


def oscar_floor_masses(mu2) -> np.ndarray:
  """Dead channels get a small positive mass. ref: oscar.py:56-59."""
  mu2 = np.asarray(mu2, np.float64)
  return np.maximum(mu2, float(np.max(mu2)) * 1e-8 + _OSCAR_TINY)


def oscar_group_clip(mag: np.ndarray, masses: np.ndarray, qmax: int) -> np.ndarray:
  """Per row of `mag` [n, g]: the clip bound c minimising
      c^2 * M / (12 qmax^2) + sum_j max(mag_j - c, 0)^2 * m_j,   M = sum(m) + tiny.
  With the k largest magnitudes clipped the objective is a quadratic in c, so every
  segment between consecutive sorted magnitudes has a closed-form candidate; the best of
  the g candidates and the "clip nothing" one wins. ref: oscar.py:62-108."""
  n = mag.shape[0]
  by_size = np.argsort(-mag, axis=1)
  top = np.take_along_axis(mag, by_size, 1)
  m = masses[by_size]
  total = float(masses.sum()) + _OSCAR_TINY
  run_m = np.cumsum(m, 1)
  run_am = np.cumsum(top * m, 1)
  run_a2m = np.cumsum(top * top * m, 1)
  cand = 2.0 * run_am / (total / (6.0 * qmax * qmax) + 2.0 * run_m)
  floor = np.concatenate([top[:, 1:], np.zeros((n, 1))], 1)
  cand = np.clip(cand, floor, top)
  noise = total / (12.0 * qmax * qmax)
  err = (cand ** 2) * noise + run_a2m - 2.0 * cand * run_am + (cand ** 2) * run_m
  all_c = np.concatenate([top[:, :1], cand], 1)
  all_e = np.concatenate([(top[:, :1] ** 2) * noise, err], 1)
  return all_c[np.arange(n), np.argmin(all_e, 1)]


def oscar_scale_objective(w: np.ndarray, s: np.ndarray, mu2: np.ndarray, block: int) -> float:
  """sum over groups of (sum of squared per-row group maxima of |w|*s) * (group mass of
  mu2 / s^2). ref: oscar.py:175-194."""
  m = mu2 / (s * s)
  mag = np.abs(w) * s
  d = w.shape[1]
  g = block if (block and d % block == 0) else d
  total = 0.0
  for lo in range(0, d, g):
    top = mag[:, lo:lo + g].max(1)
    total += float((top * top).sum()) * float(m[lo:lo + g].sum())
  return total


def oscar_channel_scales(w: np.ndarray, mu2: np.ndarray, block: int = 0, iters: int = 3):
  """Per-input-channel scales s (or None when identity is at least as good) and the gain.
  ref: oscar.py:197-263."""
  d = mu2.size
  mu2 = oscar_floor_masses(mu2)
  mu = np.sqrt(mu2)

  def unit_geomean(v):
    v = v / np.exp(np.mean(np.log(v)))
    return np.clip(v, *_OSCAR_S_RANGE)

  col_energy = (w * w).sum(0) + _OSCAR_TINY
  at_identity = oscar_scale_objective(w, np.ones(d), mu2, block)
  s = unit_geomean(np.sqrt(mu / np.sqrt(col_energy)))
  best_loss, best_s = oscar_scale_objective(w, s, mu2, block), s
  g = block if (block and w.shape[1] % block == 0) else w.shape[1]
  rows = np.arange(w.shape[0])
  for _ in range(iters):
    eff = np.zeros(d)
    mag = np.abs(w) * s
    for lo in range(0, w.shape[1], g):
      winner = lo + np.argmax(mag[:, lo:lo + g], 1)
      np.add.at(eff, winner, w[rows, winner] ** 2)
    eff = np.maximum(eff, 0.25 * col_energy)
    target = unit_geomean(np.sqrt(mu / np.sqrt(eff)))
    s = unit_geomean(np.sqrt(s * target))
    loss = oscar_scale_objective(w, s, mu2, block)
    if loss < best_loss:
      best_loss, best_s = loss, s
  if best_loss >= at_identity:
    return None, 1.0
  return best_s, at_identity / max(best_loss, _OSCAR_TINY)


def oscar_clip_bounds(w: np.ndarray, mu2, num_bits: int, granularity: str) -> np.ndarray:
  """Clip bounds in the shape min/max QSVs have. ref: oscar.py:327-383 (+ :111-153)."""
  w = np.asarray(w, np.float64)
  if w.ndim != 2:
    raise ValueError(f"OSCAR expects 2-D weights for FULLY_CONNECTED, got {w.shape}")
  n, d = w.shape
  masses = np.ones(d) if mu2 is None else np.asarray(mu2, np.float64).ravel()
  if masses.size != d:
    raise ValueError(f"OSCAR: activation mu2 has {masses.size} channels but FULLY_CONNECTED"
                     f" weights of shape {w.shape} expect {d}.")
  masses = oscar_floor_masses(masses)
  qmax = 2 ** (num_bits - 1) - 1
  mag = np.abs(w)
  if granularity == TENSORWISE:
    return oscar_group_clip(mag.reshape(1, n * d), np.tile(masses, n), qmax).reshape((1,) * 2)
  if granularity == CHANNELWISE:
    return oscar_group_clip(mag, masses, qmax).reshape(n, 1)
  if is_blockwise(granularity):
    b = block_size_of(granularity)
    if d % b:
      raise ValueError(f"Block size {b} must divide the reduction dimension {d} of"
                       f" FULLY_CONNECTED weights with shape {w.shape}.")
    out = np.empty((n, d // b))
    for k in range(d // b):
      out[:, k] = oscar_group_clip(mag[:, k * b:(k + 1) * b], masses[k * b:(k + 1) * b], qmax)
    return out
  raise ValueError(f"Unsupported granularity: {granularity}")


def oscar_quant_params(w, mu2, num_bits: int, granularity: str, symmetric: bool = True) -> dict:
  """FULLY_CONNECTED weight -> scales s, W' = W*s quantized with the optimal bounds, and the
  activation multiplier 1/s (float32). ref: oscar.py:400-478, 541-551."""
  if not symmetric:
    raise ValueError("OSCAR supports symmetric weight quantization only, got asymmetric"
                     " config for op FULLY_CONNECTED.")
  w = np.asarray(w, np.float64)
  d = w.shape[1]
  block = block_size_of(granularity) if is_blockwise(granularity) else 0
  s = None
  if mu2 is not None:
    masses = np.asarray(mu2, np.float64).ravel()
    if masses.size != d:
      raise ValueError(f"OSCAR: activation mu2 has {masses.size} channels but FULLY_CONNECTED"
                       f" weights of shape {w.shape} expect {d}.")
    s, _ = oscar_channel_scales(w, masses, block)
  if s is None:
    s = np.ones(d, np.float64)
  scaled = w * s
  scaled_masses = None if mu2 is None else np.asarray(mu2, np.float64).ravel() / (s * s)
  bounds = oscar_clip_bounds(scaled, scaled_masses, num_bits, granularity)
  zp, scale = zp_scale_from_min_max(-bounds, bounds, num_bits, True, granularity, None)
  qdim = weight_quantized_dim(granularity, "FULLY_CONNECTED", 2)
  q = uniform_quantize(scaled, scale, zp, num_bits, True, quantized_dim=qdim, block_size=block,
                       is_blockwise_quant=is_blockwise(granularity))
  return dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=True,
              quantized_dimension=qdim, block_size=block, quantized_data=q,
              multiplier=(1.0 / s).astype(np.float32))


# --------------------------------------------------------------------------
# dequantized weight recovery (fake-quantized weights -> their integers)
# --------------------------------------------------------------------------

def dwr_scales(w: np.ndarray, quantized_dim, block: int = 0, min_scale: float = 1e-9) -> np.ndarray:
  """Per-group smallest positive step between sorted magnitudes, 0 included.
  ref: algorithms/uniform_quantize/dequantized_weight_recovery.py:48-61, 118-186."""
  mag = np.abs(w)
  if quantized_dim is None:
    vals = np.unique(np.append(np.ravel(mag), 0))           # promotes to float64
    step = float(np.maximum(np.min(np.diff(vals)), min_scale)) if vals.size > 1 else min_scale
    return np.array([[step]])
  if block > 0:
    rows = mag.reshape(-1, block) if quantized_dim == mag.ndim - 1 else None
    shape = list(mag.shape)
    shape[quantized_dim] //= block
  else:
    rows = np.moveaxis(mag, quantized_dim, 0).reshape(mag.shape[quantized_dim], -1)
    shape = [1] * mag.ndim
    shape[quantized_dim] = mag.shape[quantized_dim]
  rows = np.sort(np.hstack([rows, np.zeros((rows.shape[0], 1), rows.dtype)]), axis=1)
  steps = np.diff(rows, axis=1)
  least = np.min(np.where(steps > 1e-9, steps, np.inf), axis=1)
  out = np.maximum(least, min_scale)
  out[out == np.inf] = min_scale
  return out.reshape(shape)


def dwr_quant_params(w, num_bits: int, granularity: str, op: str = "FULLY_CONNECTED",
                     check: bool = True) -> dict:
  """ref: dequantized_weight_recovery.py:189-283 (symmetric weights only)."""
  block = block_size_of(granularity) if is_blockwise(granularity) else 0
  qdim = weight_quantized_dim(granularity, op, np.ndim(w))
  scale = dwr_scales(w, qdim, block)
  zp = np.zeros_like(scale, dtype=np.int32)
  q = uniform_quantize(w, scale, zp, num_bits, True, quantized_dim=qdim, block_size=block,
                       is_blockwise_quant=is_blockwise(granularity))
  if check:
    back = uniform_dequantize(q, scale, zp, quantized_dim=qdim, block_size=block)
    worst = np.ravel(np.abs(back - w)).max()
    if worst > 1e-4:
      raise RuntimeError("Failed to recover the original quantized values from dequantized values."
                         f" Max diff between recovered and original values: {worst} (tolerance: 0.0001)")
  return dict(scale=scale, zero_point=zp, num_bits=num_bits, symmetric=True, quantized_dimension=qdim,
              block_size=block, quantized_data=q)
