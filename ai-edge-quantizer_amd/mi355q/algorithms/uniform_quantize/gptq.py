"""GPTQ (Hessian-aware weight quantization), GPU backed.

Mirror of ref: algorithms/uniform_quantize/gptq.py. calibrate() collects H = (2/num_samples) X^T X
on the matrix cores, _prepare_hessian_inverse runs the blocked Cholesky / inverse on the GPU and
_apply_gptq the column-serial OBS update (mi355q_gptq_*). QSVs stay dictionaries at this
interface, as in the reference; the min/max up-front scales are computed exactly as there.

What is organised differently from the reference (same values):
  * a Hessian lives in HBM as a `HessianAccumulator`: the sample-weighted mean the reference's
    merge rule chains per sample (utils/qsv_utils.py:71-102) equals (2/N) X^T X over all tokens seen,
    so the tokens of successive samples are collected in a slab and multiplied 16384 at a time;
  * inside `Calibrator` only tensors some GPTQ op will read a Hessian from get one (the reference
    computes one for every runtime tensor of the op, outputs included, that nothing ever reads);
  * inside `ParamsGenerator`'s batching block the ops that share a Hessian (q / k / v, gate / up)
    are applied as ONE row-concatenated update, the up-front scales are computed on the device and
    the factorization's `info` is checked once per block instead of per tensor.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
from collections.abc import Mapping, MutableMapping, Sequence
from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import requant_queue
from ... import runtime as rt
from ..utils import common_utils
from . import common_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "GPTQ"

# Names of the runtime tensors whose Hessian some op will read (set by Calibrator for the duration
# of a walk); None = every tensor, as the reference does.
_HESSIAN_READERS: Optional[set] = None


@contextlib.contextmanager
def hessians_only_for(names: Optional[set]):
  global _HESSIAN_READERS
  saved, _HESSIAN_READERS = _HESSIAN_READERS, names
  try:
    yield
  finally:
    _HESSIAN_READERS = saved


def calibrate(tfl_op, graph_info: qtyping.GraphInfo,
              tensor_content_map: MutableMapping[str, np.ndarray],
              inputs_to_ignore: Sequence[int] | None = None,
              outputs_to_ignore: Sequence[int] | None = None,
              valid_range: tuple[float, float] = (-3e38, 3e38)) -> dict[str, qtyping.QSV]:
  """min/max + Hessian of every runtime tensor of the op (ref :55-108)."""
  lo, hi = valid_range
  out = {}
  for tid in common_quantize.get_tensor_indices_requiring_calibration(
      tfl_op, graph_info, inputs_to_ignore, outputs_to_ignore):
    res = common_quantize.collect_activation_tensor_statistics(
        tid, graph_info, tensor_content_map, valid_float_range_min=lo, valid_float_range_max=hi)
    if res is None:
      continue
    name, content, qsv = res
    if _HESSIAN_READERS is None or name in _HESSIAN_READERS:
      qsv["hessian"] = hessian_of(content, qsv["num_samples"])
    out[name] = qsv
  return out


_BORROWED: Optional[list] = None


@contextlib.contextmanager
def borrowing():
  """Inside this block (one calibration step) the per-sample statistic that `calibrate` returns for a staged
  activation only REFERS to the sample's tokens; the merge that follows (qsv_utils._gptq_merge_hessian ->
  absorb) copies them once, into the running statistic's slab, instead of twice (a slab of the sample's own
  first). What has not been merged when the block exits -- a tensor's first sample becomes the running
  statistic itself -- takes its copy then: the tokens may change afterwards."""
  global _BORROWED
  if os.environ.get("MI355Q_GPTQ_BORROW", "1") == "0":     # (for A / B timing)
    yield
    return
  outer, _BORROWED = _BORROWED, []
  mine = _BORROWED
  try:
    yield
  finally:
    _BORROWED = outer
    for acc in mine:
      acc.own()


class HessianAccumulator(rt.HbmArray):
  """float64 [d, d] = the sample-weighted mean of (2/n_i) X_i^T X_i over the samples added so far.

  The mean over samples i of H_i weighted by n_i is (2/N) sum_i X_i^T X_i, N = sum n_i: tokens that
  have not been multiplied yet wait in a float32 slab and leave SLAB_TOKENS at a time through
  mi355q_gptq_xtx_accum_f32 into ONE float32 product (the bf16-split product needs >= 1024
  tokens to run at all, and one product of 16384 tokens costs a tenth of 32 products of 512;
  slabs add to the product in float32 like the K loop's own steps). The product becomes the
  float64 Hessian -- (2/N) product, mi355q_gptq_xtx_finish_f64 -- when somebody reads the array
  (device_tensor, np.asarray, ...) and is given back by finalize(). A Hessian that arrives as a
  finished float64 array (resumed calibration, another process's share) joins through
  mi355q_gptq_hessian_merge_f64, the reference's rule (utils/qsv_utils.py:71-88)."""
  SLAB_TOKENS = 16384        # tokens per product for narrow Hessians; wide ones: SLAB_BYTES of staging
  SLAB_BYTES = 256 << 20     # (d = 16384: 4096 tokens -- a slab's product joins the float32 product with one
                             # read-modify-write of the triangle, 0.3 ms against 6 ms of multiplying, and a
                             # 16384-token slab would be 1 GiB of staging per Hessian)

  def __init__(self, d: int):  # pylint: disable=super-init-not-called
    self.d = int(d)
    self.SLAB_TOKENS = max(4096, min(HessianAccumulator.SLAB_TOKENS, (self.SLAB_BYTES // (4 * self.d)) // 1024 * 1024))
    self._mean = None             # float64 [d, d] over _n_done samples (only what joined as float64)
    self._n_done = 0.0
    self._prod = None             # float32 [d, d]: sum of X^T X over _n_prod samples
    self._n_prod = 0.0
    self._slab = None             # float32 [capacity, d]
    self._fill = 0
    self._n_pending = 0.0
    self._value = None            # the float64 statistic, while nothing changed since it was formed
    self._borrowed = None         # (x2d, num_samples) of ONE sample still in the caller's tensor (see borrowing())
    self._host = None
    self.cache = {}
    self.packed = None
    self.ready = None             # event behind a reduce across ranks that is still filling _prod on another stream

  @classmethod
  def of(cls, x2d, num_samples: float, borrow: bool = False) -> "HessianAccumulator":
    """The statistic of one sample. `borrow`: the tokens stay where they are until this statistic is merged
    into another one (absorb: ONE copy, into the running slab) or read (own())."""
    acc = cls(x2d.shape[1])
    if borrow:
      acc._borrowed = (x2d, float(num_samples))
    else:
      acc.add(x2d, num_samples)
    return acc

  @classmethod
  def resumed(cls, hessian, num_samples: float) -> "HessianAccumulator":
    """An accumulator whose float64 share is a finished Hessian (the mean over `num_samples` samples: an ndarray or an
    array resident in HBM); samples added from here on are weighed against it by the reference's rule."""
    h = rt.on_device(hessian)
    acc = cls(int(h.shape[0]))
    acc._join(h, float(num_samples))
    return acc

  def own(self) -> None:
    """A borrowed sample is taken over (copied into this statistic's own slab, or multiplied)."""
    if self._borrowed is not None:
      (x2d, n), self._borrowed = self._borrowed, None
      self.add(x2d, n)

  def _touched(self) -> None:
    self._value = self._host = None
    self.cache.clear()

  def add(self, x2d, num_samples: float) -> None:
    """x2d: float32 device tensor [tokens, d] (copied: the caller's buffer may change)."""
    import torch
    self.own()
    t = int(x2d.shape[0])
    self._touched()
    if t >= self.SLAB_TOKENS:        # a slab's worth on its own: multiplied where it lies
      self.flush()
      self._prod = ops.gptq_xtx_accum(x2d.contiguous(), self._prod)
      self._n_prod += float(num_samples)
      return
    if self._slab is not None and self._fill + t > self._slab.shape[0] and self._fill:
      if self._slab.shape[0] < self.SLAB_TOKENS and self._fill + t <= self.SLAB_TOKENS:
        grown = torch.empty((self.SLAB_TOKENS, self.d), dtype=torch.float32, device=x2d.device)
        grown[:self._fill].copy_(self._slab[:self._fill])
        self._slab = grown
      else:
        self.flush()
    if self._slab is None or self._slab.shape[0] < t:
      self._slab = torch.empty((t, self.d), dtype=torch.float32, device=x2d.device)
    self._slab[self._fill:self._fill + t].copy_(x2d)
    self._fill += t
    self._n_pending += float(num_samples)

  def add_block(self, xs, ns) -> None:
    """add(xs[0], ns[0]); add(xs[1], ns[1]); ... with the same products of the same tokens in the same order (a product
    covers the samples that fit SLAB_TOKENS; the next sample that does not fit closes it), hence the same float32
    bits -- but a run of samples that fills a product exactly and lies back to back in the caller's memory is
    multiplied where it lies, and the others reach the slab in one copy per run instead of one per sample."""
    import torch
    self.own()
    self._touched()
    cap = self.SLAB_TOKENS
    i, n = 0, len(xs)
    while i < n:
      t = int(xs[i].shape[0])
      if t >= cap:
        self.add(xs[i], ns[i])
        i += 1
        continue
      if self._fill and self._fill + t > cap:
        self.flush()
      j, total = i, self._fill
      while j < n and int(xs[j].shape[0]) < cap and total + int(xs[j].shape[0]) <= cap:
        total += int(xs[j].shape[0])
        j += 1
      run, count = xs[i:j], float(sum(float(v) for v in ns[i:j]))
      tokens = total - self._fill
      if not self._fill and (total == cap or j < n):   # this product is complete: the next sample does not fit in
        joined = _back_to_back(run, tokens, self.d)
        if joined is not None:
          self._prod = ops.gptq_xtx_accum(joined, self._prod)
          self._n_prod += count
          i = j
          continue
      if self._slab is None or self._slab.shape[0] < cap:
        grown = torch.empty((cap, self.d), dtype=torch.float32, device=run[0].device)
        if self._fill:
          grown[:self._fill].copy_(self._slab[:self._fill])
        self._slab = grown
      dst = self._slab[self._fill:total]
      if len(run) == 1:
        dst.copy_(run[0])
      else:
        torch.cat(run, out=dst)
      self._fill = total
      self._n_pending += count
      i = j

  def _join(self, h, n: float) -> None:
    """float64 mean <- weighted mean with Hessian h of n samples (ref utils/qsv_utils.py:71-88)."""
    if self._mean is None:
      self._mean, self._n_done = h, n
    else:
      self._mean = ops.gptq_hessian_merge(self._mean, self._n_done, h, n)
      self._n_done += n

  def absorb(self, other: "HessianAccumulator") -> None:
    """self <- the mean over both sets of samples."""
    self._wait_ready()
    other._wait_ready()   # pylint: disable=protected-access
    self._touched()
    if other._borrowed is not None:   # pylint: disable=protected-access
      (x2d, n), other._borrowed = other._borrowed, None   # pylint: disable=protected-access
      self.add(x2d, n)
    if other._fill:   # pylint: disable=protected-access
      self.add(other._slab[:other._fill], other._n_pending)   # pylint: disable=protected-access
    if other._prod is not None:   # pylint: disable=protected-access
      self.flush()
      if self._prod is None:
        self._prod, self._n_prod = other._prod.clone(), other._n_prod   # pylint: disable=protected-access
      else:
        self._prod += other._prod   # pylint: disable=protected-access
        self._n_prod += other._n_prod   # pylint: disable=protected-access
    if other._mean is not None:   # pylint: disable=protected-access
      self._join(other._mean, other._n_done)   # pylint: disable=protected-access

  def _wait_ready(self) -> None:
    if self.ready is not None:
      # the product is the sum over the ranks once this event has fired (distributed.reduce_products_beside_compute):
      # every reader comes through here, and only the stream that reads waits
      import torch
      torch.cuda.current_stream().wait_event(self.ready)
      self.ready = None

  def flush(self) -> None:
    self._wait_ready()
    self.own()
    if not self._fill:
      return
    self._prod = ops.gptq_xtx_accum(self._slab[:self._fill], self._prod)
    self._n_prod += self._n_pending
    self._fill, self._n_pending = 0, 0.0

  def finalize(self) -> None:
    """Multiplies what is pending and gives the slab back. The statistic stays in its float32
    product form: the damped inverse is taken straight from it (product_form), and the float64
    array is only made if somebody reads it."""
    self.flush()
    self._slab = None

  def product_form(self):
    """(product float32 [d, d], alpha) with hessian = alpha * product, or None when float64
    Hessians have joined the statistic (it then only exists as the float64 array)."""
    self.flush()
    if self._mean is None and self._prod is not None:
      return self._prod, 2.0 / self._n_prod
    return None

  @property
  def device_tensor(self):
    if self._value is None:
      self.flush()
      if self._prod is None:
        self._value = self._mean
      else:
        h = ops.gptq_xtx_finish(self._prod, 2.0 / self._n_prod)
        self._value = h if self._mean is None else ops.gptq_hessian_merge(self._mean, self._n_done, h, self._n_prod)
    return self._value

  @device_tensor.setter
  def device_tensor(self, value) -> None:
    self._mean, self._prod, self._n_prod, self._fill, self._n_pending = value, None, 0.0, 0, 0.0
    self._borrowed = None
    self._value = value

  @property
  def shape(self):
    return (self.d, self.d)

  @property
  def ndim(self) -> int:
    return 2

  @property
  def dtype(self):
    return np.dtype(np.float64)

  @property
  def size(self) -> int:
    return self.d * self.d

  @property
  def nbytes(self) -> int:
    return self.d * self.d * 8

  def __repr__(self):
    return (f"HessianAccumulator(d={self.d}, samples={self._n_done + self._n_prod + self._n_pending:g},"
            f" pending_tokens={self._fill})")


def _back_to_back(run, tokens: int, d: int):
  """One [tokens, d] view over samples that follow one another in the same allocation, or None."""
  import torch
  first = run[0]
  if not all(x.is_contiguous() for x in run):
    return None
  at = first.data_ptr()
  for x in run:
    if x.data_ptr() != at:
      return None
    at += x.numel() * 4
  if len(run) == 1:
    return first
  store = first.untyped_storage()
  if any(x.untyped_storage().data_ptr() != store.data_ptr() for x in run):
    return None
  try:
    return torch.as_strided(first, (tokens, d), (d, 1))
  except RuntimeError:
    return None


def hessian_of(tensor_content: np.ndarray, num_samples):
  """(2.0 / num_samples) * x.T.dot(x), x = content.reshape(-1, last) (ref :100-107).

  float32 content: X^T X on the matrix cores, scaled into float64 (NumPy's promotion of
  `2.0 / np.array(n)`), resident in HBM. float64 content (the reference's own test feeds 1e39):
  the FP64 MFMA GEMM.
  """
  alpha = 2.0 / np.asarray(num_samples)
  rt.require_gpu()
  rec = rt.staged(tensor_content)        # the calibrator put this sample's activations in HBM
  if rec is not None and tensor_content.dtype == np.float32:
    xd = rec["dev"].reshape(-1, tensor_content.shape[-1])
    acc = HessianAccumulator.of(xd, float(np.asarray(num_samples)), borrow=_BORROWED is not None)
    if _BORROWED is not None:
      _BORROWED.append(acc)
    return acc
  x = np.ascontiguousarray(tensor_content.reshape([-1, tensor_content.shape[-1]]))
  if x.dtype == np.float32:   # stays in HBM: merged per sample and consumed by the GPU again
    return HessianAccumulator.of(rt.to_device(x), float(np.asarray(num_samples)))
  if x.dtype == np.float64:
    xd = rt.to_device(x)
    return alpha * rt.to_numpy(ops.gemm(xd, xd, trans_a=True))
  raise TypeError(f"GPTQ calibration expects float32 / float64 activations, got {x.dtype}")


def _prepare_hessian_inverse(hessian: np.ndarray, damp_factor: float = 0.01) -> np.ndarray:
  """Damped inverse through Cholesky; float32 result (ref :111-128)."""
  hinv, info = _device_hessian_inverse(hessian, damp_factor)
  _check_info(info)
  return rt.to_numpy(hinv)


def _check_info(info) -> None:
  if int(info.max().item() if info.numel() > 1 else info.item()) != 0:
    raise np.linalg.LinAlgError("Matrix is not positive definite")


_READERS_LEFT = "gptq readers left in this plan"


def _device_hessian_inverse(hessian, damp_factor: float = 0.01):
  """(hinv float32 [d,d], info) on the device. A Hessian that lives in HBM remembers its
  inverse: q / k / v (and gate / up) share one input activation, hence one Hessian."""
  rt.require_gpu()
  if isinstance(hessian, rt.HbmArray):
    key = ("hinv", float(damp_factor))
    if key not in hessian.cache:
      form = hessian.product_form() if isinstance(hessian, HessianAccumulator) else None
      # (a statistic still in product form is inverted from there: no 2 GiB float64 copy at d = 16384)
      hessian.cache[key] = (ops.gptq_hinv_from_product(form[0], form[1], damp_factor) if form is not None
                            else ops.gptq_hinv(rt.on_device(hessian, torch_f64()), damp_factor))
    return hessian.cache[key]
  return ops.gptq_hinv(rt.to_device(np.ascontiguousarray(hessian, dtype=np.float64)), damp_factor)


def hessian_name_of(plan_item) -> Optional[str]:
  """Name of the activation whose Hessian a planned op reads (its first input, ref :243-300), None when the op
  is not a GPTQ op."""
  from ...utils import tfl_flatbuffer_utils
  graph_info, op, _, op_key, alg, _ = plan_item
  if str(getattr(alg, "value", alg)) != ALGORITHM_KEY or op_key is None or not len(op.inputs):
    return None
  return tfl_flatbuffer_utils.get_tensor_name(graph_info.subgraph_tensors[op.inputs[0]])


def largest_hessian_order(plan_items) -> int:
  """The largest input width among the FULLY_CONNECTED-like GPTQ ops of a plan (0: none): the order of the largest
  Hessian its calibration will produce."""
  best = 0
  for graph_info, op, _, op_key, alg, _ in plan_items:
    if str(getattr(alg, "value", alg)) != ALGORITHM_KEY or op_key is None or len(op.inputs) < 2 or op.inputs[1] == -1:
      continue
    shape = graph_info.subgraph_tensors[op.inputs[1]].shape
    if shape is not None and len(shape) == 2:
      best = max(best, int(shape[-1]))
  return best


def prefetch_hessian_inverses(plan_items, model_qsvs, damp_factor: float = 0.01) -> int:
  """Inverts, in one batched call per order, every Hessian the GPTQ ops among `plan_items`
  (ParamsGenerator.plan_ops tuples) will read and that has no cached inverse yet. A model has
  one Hessian per distinct FULLY_CONNECTED input and most are small (54 of order 2048 in a
  Gemma-2B): alone each is a latency-bound chain of kernels, together they fill the chip
  (mi355q_gptq_hinv_f64_batched). Returns how many were inverted here."""
  from ...utils import tfl_flatbuffer_utils
  todo: dict[int, list] = {}
  seen: set[int] = set()
  readers: dict[int, list] = {}
  for graph_info, op, _, op_key, alg, _ in plan_items:
    if str(getattr(alg, "value", alg)) != ALGORITHM_KEY or op_key is None or not len(op.inputs):
      continue
    name = tfl_flatbuffer_utils.get_tensor_name(graph_info.subgraph_tensors[op.inputs[0]])
    qsv = model_qsvs.get(name) if model_qsvs else None
    h = qsv.get("hessian") if isinstance(qsv, dict) else None
    if not isinstance(h, rt.HbmArray):
      continue
    readers.setdefault(id(h), [h, 0])[1] += 1
    if id(h) in seen or ("hinv", float(damp_factor)) in h.cache:
      continue
    seen.add(id(h))
    if h.shape[0] < 4096:          # larger ones fill the chip on their own (and need 4 GiB of scratch each)
      todo.setdefault(h.shape[0], []).append(h)
  # how many ops of this plan read each Hessian: the apply queue gives an inverse back to the allocator when its
  # last reader's apply is out (a d = 16384 inverse is 1 GiB, and a fresh GiB of HBM costs ~30 ms of hipMalloc:
  # kept until the end of the model, every layer's inverse was a new allocation)
  for h, n in readers.values():
    h.cache[_READERS_LEFT] = n
  count = 0
  for hs in todo.values():
    if len(hs) < 2:
      continue
    rt.require_gpu()
    for h, res in zip(hs, ops.gptq_hinv_batched([rt.on_device(h, torch_f64()) for h in hs], damp_factor)):
      h.cache[("hinv", float(damp_factor))] = res
      count += 1
  return count


def torch_f64():
  import torch
  return torch.float64


def _scale_mode(scale_size: int, rows: int, d: int, blockwise: bool, block_size: int):
  if blockwise:
    return 2, block_size
  if scale_size == 1:
    return 0, 0
  if scale_size == rows:
    return 1, 0
  raise NotImplementedError(f"scale of {scale_size} values for a [{rows}, {d}] weight")


def _apply_gptq(tensor_content: np.ndarray, quant_params: qtyping.UniformQuantParams,
                activation_tensor_qsv: Mapping[str, Any],
                tensor_quant_config: qtyping.TensorQuantizationConfig,
                blocksize: int = 64) -> qtyping.UniformQuantParams:
  """Blocked OBS update + column-serial quantization (ref :131-216)."""
  if not isinstance(blocksize, (int, np.integer)) or blocksize < 1:
    raise ValueError(f"blocksize must be a positive integer, got {blocksize!r}")
  # `blocksize` is how many columns the reference sweeps before it pushes their errors into the columns behind them
  # (ref :162-214): a schedule of the same updates -- in exact arithmetic every blocksize gives the same integers, in
  # float32 it moves where the far columns are rounded. The kernels sweep 64 columns (and push up to four blocks at once
  # where that never moved an integer, csrc/gptq.hip); another blocksize is answered by the same kernels, within the
  # tolerance class (T2) every GPTQ result is in -- tests/test_gpu_gptq.py compares against the oracle at 1, 16, 100, 128, 256.
  if tensor_quant_config.num_bits > 32:
    raise ValueError(f"Unsupported num_bits for quantization: {tensor_quant_config.num_bits}")    # ref :141-151
  if tensor_content.ndim != 2 or tensor_content.dtype != np.float32:
    raise TypeError("GPTQ expects a 2-D float32 weight")
  rt.require_gpu()
  rows, d = tensor_content.shape
  hinv, info = _device_hessian_inverse(activation_tensor_qsv["hessian"], 0.01)
  scale, zp = quant_params.scale, quant_params.zero_point
  blockwise = uniform_quantize_tensor.is_blockwise(tensor_quant_config.granularity)
  mode, bs = _scale_mode(scale.size, rows, d, blockwise, quant_params.block_size)
  if not np.issubdtype(zp.dtype, np.signedinteger):
    raise ValueError(f"zero_points need to be {np.signedinteger}. But the actual type is"
                     f" {zp.dtype}.")
  sdt = np.float64 if scale.dtype == np.float64 else np.float32
  s_dev = rt.to_device(np.ascontiguousarray(scale.reshape(-1), dtype=sdt))
  # all-zero zero points (symmetric recipes) are passed as "none": same results, and the block
  # kernel then runs its specialised form
  z_dev = (rt.to_device(np.ascontiguousarray(np.broadcast_to(zp, scale.shape).reshape(-1)).astype(np.int32))
           if np.any(zp) else None)
  narrow = bool(quant_params.symmetric and quant_params.num_bits >= 8)
  target = (np.int8 if quant_params.num_bits <= 8 else np.int16 if quant_params.num_bits <= 16
            else np.int32)                                            # ref :141-151 (_get_quantized_dtype)
  diff_bits = min(32, np.result_type(target, zp.dtype).itemsize * 8)
  q = ops.gptq_apply(rt.to_device(tensor_content), hinv, s_dev, z_dev, mode, bs,
                     quant_params.num_bits, narrow, zp.dtype.itemsize >= 4, diff_bits)
  _check_info(info)
  if 8 < quant_params.num_bits <= 16:      # (the wide entry point returns int32: narrowed to the reference's container)
    import torch
    q = q.to(torch.int16)
  return dataclasses.replace(quant_params, quantized_data=rt.quantized_result(
      q, quant_params.num_bits, tensor_content.nbytes, tensor_content.shape))


# ------------------------------------------------------------------ queued applies ---
class _Pending(requant_queue.PendingArray):
  """Placeholder whose device tensor the apply queue sets when its group has been issued."""

  @property
  def device_tensor(self):
    if self._tensor is None:
      self._queue.flush()
    return self._tensor

  @device_tensor.setter
  def device_tensor(self, value) -> None:
    self._tensor = value

  @property
  def resolved(self) -> bool:
    return self._tensor is not None

  def numpy(self) -> np.ndarray:
    if self._host is None:
      self._host = self.device_tensor.cpu().numpy()
    return self._host


class _ApplyQueue:
  """GPTQ weights met inside `requant_queue.batching()`: symmetric channel- / blockwise targets.

  submit(): the weight goes to HBM, its up-front min/max scales are made by the fused launch that
  serves the min/max algorithm (mi355q_requant_sym_f32 with only the scale output: the same
  arithmetic as init_tensor_min_max + tensor_zp_scale_from_min_max, bit for bit), and the caller
  gets placeholders. Weights that read the same Hessian with the same target wait for each other;
  when a different Hessian arrives (or the block exits) they leave as ONE row-concatenated
  mi355q_gptq_apply_f32 -- rows are independent (ref :131-216), the column-serial chain is paid
  once. `info` of every inverse used is checked when the block exits."""

  def __init__(self, host_queue):
    self._key = None
    self._entries: list = []
    self._hinv = None
    self._hessian = None
    self._infos: list = []
    self._scales: list = []          # (placeholder, params) of per-row scales to hand over as ndarrays
    self.stats = host_queue.stats
    self.stats.setdefault("gptq_tensors", 0)
    self.stats.setdefault("gptq_applies", 0)

  def submit(self, w_dev, hessian, cfg, block_size: int, shape):
    import torch
    rows, d = shape
    bits = cfg.num_bits
    res = ops.requant_sym(w_dev, block_size, bits, want_q=False)
    scale_dev = res["scale"]
    key = (id(hessian), bits, block_size)
    if key != self._key:
      self.flush()
      self._key = key
      self._hessian = hessian
      self._hinv, info = _device_hessian_inverse(hessian, 0.01)
      self._infos.append(info)
    left = hessian.cache.get(_READERS_LEFT) if isinstance(hessian, rt.HbmArray) else None
    if left is not None:
      hessian.cache[_READERS_LEFT] = left - 1
    scale = _Pending(((rows, d // block_size) if block_size else (rows, 1)), np.dtype(np.float32), self)
    scale.device_tensor = scale_dev.reshape(scale.shape)
    if bits in (2, 4):
      packed = _Pending((rows * d * bits // 8,), np.dtype(np.uint8), self)
      q = _Pending((rows, d), np.dtype(np.int8), self)
      q.packed = packed
    else:
      q, packed = _Pending((rows, d), np.dtype(np.int8), self), None
    self._entries.append((w_dev, scale_dev.reshape(-1), q, packed, bits, block_size))
    self.stats["gptq_tensors"] += 1
    del torch
    return scale, q

  def flush(self) -> None:
    if not self._entries:
      return
    import torch
    entries, self._entries = self._entries, []
    _, _, _, _, bits, bs = entries[0]
    if len(entries) == 1:
      w, s = entries[0][0], entries[0][1]
    else:
      w = torch.cat([e[0] for e in entries], dim=0)
      s = torch.cat([e[1] for e in entries])
    mode = 2 if bs else 1
    q_all = ops.gptq_apply(w, self._hinv, s, None, mode, bs, bits, bits >= 8, False, 8)
    self.stats["gptq_applies"] += 1
    h = self._hessian
    if isinstance(h, rt.HbmArray) and h.cache.get(_READERS_LEFT) == 0:
      # the last op of the plan that reads this Hessian: its inverse goes back to the allocator
      # (stream-ordered: the apply just enqueued still reads it, whoever gets the block next runs behind it)
      h.cache.pop(("hinv", 0.01), None)
      h.cache.pop(_READERS_LEFT, None)
      self._hinv = self._key = None
    packed_all = ops.pack_bits(q_all, bits) if bits in (2, 4) else None
    produced = torch.cuda.Event()      # behind this group's payloads: the file writer waits for THEM, not for the ops queued later
    produced.record()
    row0 = 0
    d = w.shape[1]
    for w_e, _, q, packed, _, _ in entries:
      rows = w_e.shape[0]
      q.device_tensor = q_all[row0:row0 + rows]
      q.ready = produced
      if packed is not None:
        packed.ready = produced
        per = 8 // bits
        packed.device_tensor = packed_all[row0 * d // per:(row0 + rows) * d // per]
      row0 += rows

  def attach(self, scale, params) -> None:
    if params.block_size == 0:
      self._scales.append((scale, params))

  def complete(self) -> None:
    """Block exit: last applies out, one copy for all per-row scales, one look at the infos."""
    import torch
    self.flush()
    self._key = self._hinv = self._hessian = None
    if self._scales:
      flat = torch.cat([s.device_tensor.reshape(-1) for s, _ in self._scales]).cpu().numpy()
      pos = 0
      for s, params in self._scales:
        n = s.size
        s._host = flat[pos:pos + n].reshape(s.shape)   # pylint: disable=protected-access
        pos += n
        if params.scale is s:
          object.__setattr__(params, "scale", s._host)   # pylint: disable=protected-access
      self._scales = []
    if self._infos:
      infos, self._infos = self._infos, []
      _check_info(torch.cat([i.reshape(-1) for i in infos]))


def _apply_queue() -> Optional[_ApplyQueue]:
  host = requant_queue.active()
  if host is None:
    return None
  q = getattr(host, "gptq", None)
  if q is None:
    q = host.gptq = _ApplyQueue(host)
    host.defer(q.complete)
  return q


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: np.ndarray | None = None, tensor_qsv: Mapping[str, Any] | None = None,
) -> qtyping.UniformQuantParams:
  """ref :219-300."""
  cfg = tensor_quant_config
  act_qsv = tensor_qsv.get("activation_tensor_qsv") if tensor_qsv else None
  has_hessian = act_qsv is not None and "hessian" in act_qsv
  if (has_hessian and isinstance(tensor_content, (np.ndarray, rt.HbmArray))
      and tensor_content.dtype == np.float32 and tensor_content.ndim == 2):
    # one upload serves both the min / max below and the update (a Gemma-2B layer is 440 MB)
    rt.require_gpu()
    queue = _apply_queue()
    if not isinstance(tensor_content, rt.HbmArray):
      # (inside the model-level loop the GPU is busy with the previous ops' inverses and updates:
      # the upload must not wait for them, see runtime.upload_overlapped)
      big = queue is not None and tensor_content.nbytes >= (1 << 20)
      tensor_content = rt.HbmArray(rt.upload_overlapped(tensor_content) if big else rt.to_device(tensor_content))
    block_size = uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity)
    rows, d = tensor_content.shape
    if (queue is not None and cfg.symmetric and cfg.num_bits in (2, 4, 8)
        and (tensor_qsv is None or "min" not in tensor_qsv)
        and (block_size or cfg.granularity == qtyping.QuantGranularity.CHANNELWISE)
        and (not block_size or (d % block_size == 0 and block_size % 4 == 0))
        and common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity) == (1 if block_size else 0)):
      w_dev = tensor_content.device_tensor
      if not w_dev.is_contiguous():
        w_dev = w_dev.contiguous()
      scale, q = queue.submit(w_dev, act_qsv["hessian"], cfg, block_size, (rows, d))
      zp = np.zeros(scale.shape, dtype=np.int8)       # symmetric: zeros cast to the quantized type (ref a2)
      params = qtyping.UniformQuantParams(
          scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=True,
          quantized_dimension=1 if block_size else 0, block_size=block_size, quantized_data=q)
      queue.attach(scale, params)
      return params
  if tensor_qsv is None or "min" not in tensor_qsv:
    if tensor_content is None:
      raise ValueError(
          f"{op_info.op_name}(index: {op_info.subgraph_op_index}) not found in"
          " tensor_name_to_qsv. Check if the correct calibration results are passed into the"
          " ParamsGenerator.")
    tensor_min_max = common_quantize.init_tensor_min_max(tensor_content, op_info)
  else:
    tensor_min_max = tensor_qsv
  if "min" not in tensor_min_max or "max" not in tensor_min_max:
    raise ValueError(
        "min and max must be provided to produce tensor quantization parameters. Check if the"
        " correct calibration results are passed into the ParamsGenerator.")
  zp, scale = uniform_quantize_tensor.tensor_zp_scale_from_min_max(
      tensor_min_max["min"], tensor_min_max["max"], cfg.num_bits, cfg.symmetric, cfg.granularity,
      None)
  params = qtyping.UniformQuantParams(
      scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=cfg.symmetric,
      quantized_dimension=common_utils.get_weight_quantized_dim(op_info, tensor_content,
                                                               cfg.granularity),
      block_size=uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity))
  if tensor_content is None or not has_hessian:
    return params
  return _apply_gptq(tensor_content, params, act_qsv, cfg)
