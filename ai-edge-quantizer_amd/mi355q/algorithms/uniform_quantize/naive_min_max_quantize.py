"""Min/max uniform quantization, GPU backed ("min_max_uniform_quantize").

Mirror of ref: algorithms/uniform_quantize/naive_min_max_quantize.py. For a
symmetric weight whose scales run along rows (CHANNELWISE on dim 0) or along
blocks of the innermost dim (BLOCKWISE_*), min/max -> scale -> quantize is ONE
fused libmi355q launch that reads the FP32 buffer once
(mi355q_requant_sym_f32). Every other case (asymmetric, TENSORWISE, channel-last,
supplied QSV) uses the K1 / K3 kernels with the scalar parameter math on the host.
"""
from __future__ import annotations

import dataclasses
from collections.abc import MutableMapping, Sequence
from typing import Any, Optional

import numpy as np

from ... import ops
from ... import qtyping
from ... import requant_queue
from ... import runtime as rt
from ...utils import tfl_flatbuffer_utils
from ..utils import common_utils
from . import common_quantize
from . import uniform_quantize_tensor

ALGORITHM_KEY = "min_max_uniform_quantize"
_IntType = uniform_quantize_tensor.IntType


def fused_weight_layout(tensor_content: np.ndarray, granularity, quantized_dim):
  """(rows, cols, block) when the tensor fits the fused kernel's layout, else None."""
  if tensor_content is None or tensor_content.dtype != np.float32 or tensor_content.size == 0:
    return None
  shape = tensor_content.shape
  if uniform_quantize_tensor.is_blockwise(granularity):
    block = uniform_quantize_tensor.extract_block_size_from_granularity(granularity)
    if quantized_dim is None or any(d != 1 for d in shape[quantized_dim + 1:]):
      return None
    cols = shape[quantized_dim]
    if block == 0 or cols % block:
      return None
    return int(tensor_content.size // cols), int(cols), int(block)
  if granularity == qtyping.QuantGranularity.CHANNELWISE and quantized_dim is not None:
    if any(d != 1 for d in shape[:quantized_dim]):
      return None
    rows = shape[quantized_dim]
    return int(rows), int(tensor_content.size // rows), 0
  return None




def packs_in_kernel(layout, num_bits: int) -> bool:
  """The vectorized kernels pack sub-byte results; the generic fallback (odd widths) does not
  (include/mi355q.h)."""
  _, cols, block = layout
  return (num_bits in (2, 4) and cols % 4 == 0
          and (block in (32, 64, 128, 256) or (block == 0 and cols <= 16384)))


_MIN_BATCHED_BYTES = 64 << 10


def batchable(layout, tensor_content) -> bool:
  """Tensors worth queueing for the batched launch (requant_queue): tiny ones are quantized on
  the spot, their launch cost is the same either way."""
  rows, cols, _ = layout
  return rows * cols * 4 >= _MIN_BATCHED_BYTES


def fused_symmetric_requant(tensor_content: np.ndarray, layout, num_bits: int,
                            clip: Optional[np.ndarray] = None):
  """Runs mi355q_requant_sym_f32; returns (scale f32 [n_scales], q int8 like tensor).

  For sub-byte widths the same launch also emits the packed bytes the QUANTIZE_TENSOR
  transformation stores (ref transformation_utils.py:293-353); they ride along on the returned
  array's `packed` attribute so the model writer does not have to upload q again to pack it.
  """
  rows, cols, block = layout
  rt.require_gpu()
  x = rt.to_device(tensor_content.reshape(rows, cols))
  if clip is not None:                # host values, or the device tensor a clip search left behind
    clip = (rt.to_device(np.ascontiguousarray(clip, np.float32).reshape(-1))
            if isinstance(clip, np.ndarray) else clip.reshape(-1))
  c = clip
  sub_byte = packs_in_kernel(layout, num_bits)
  r = ops.requant_sym(x, block, num_bits, clip=c, want_q=True, want_packed=sub_byte)
  if tensor_content.nbytes >= rt.KEEP_IN_HBM_BYTES:
    # large weights stay in HBM until the model writer copies them (packed bytes for sub-byte
    # types) straight into the output file's mapping; NumPy consumers get a host copy on demand
    q = rt.HbmArray(r["q"].reshape(tensor_content.shape))
    if sub_byte:
      q.packed = rt.HbmArray(r["packed"])
    # ... and so do their scales: reading them back here would make every call wait for its own
    # kernels (25-30 us of a 50-80 us call) instead of letting the next tensor's be enqueued
    return rt.HbmArray(r["scale"].reshape(-1)), q
  q = rt.to_numpy(r["q"]).reshape(tensor_content.shape)
  if sub_byte:
    q = q.view(PackedCarrier)
    q.packed = rt.to_numpy(r["packed"])
  return rt.to_numpy(r["scale"]).reshape(-1), q


class PackedCarrier(np.ndarray):
  """int8 quantized values (a plain ndarray in every respect) + the packed bytes of the same
  values as produced by the kernel that quantized them. Views / copies drop the attribute."""
  packed: Optional[np.ndarray] = None

  def __array_finalize__(self, obj):
    self.packed = None


def scale_shape_for(tensor_content: np.ndarray, granularity, quantized_dim) -> tuple[int, ...]:
  """Shape the reference gives `scale` (keepdims channelwise; squeezed block axis)."""
  shape = tensor_content.shape
  if uniform_quantize_tensor.is_blockwise(granularity):
    block = uniform_quantize_tensor.extract_block_size_from_granularity(granularity)
    return tuple(d // block if i == quantized_dim else d for i, d in enumerate(shape))
  return tuple(d if i == quantized_dim else 1 for i, d in enumerate(shape))


_QUEUED_PLANS: dict = {}


def _queued_plan(op_info, cfg, tensor_content):
  """(layout, quantized_dim, block_size, scale shape, packs in kernel, zero points) of a weight
  that goes to the requant queue, or None when it does not (no fused layout, too small)."""
  if tensor_content is None:
    return None
  op_name = op_info.op_name
  adj = op_info.op.builtinOptions.adjY if op_name == qtyping.TFLOperationName.BATCH_MATMUL else None
  key = (op_name, adj, cfg.granularity, cfg.num_bits, tensor_content.shape, tensor_content.dtype)
  plan = _QUEUED_PLANS.get(key, _QUEUED_PLANS)
  if plan is _QUEUED_PLANS:
    quantized_dim = common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity)
    layout = fused_weight_layout(tensor_content, cfg.granularity, quantized_dim)
    plan = None
    if layout is not None and batchable(layout, tensor_content):
      scale_shape = scale_shape_for(tensor_content, cfg.granularity, quantized_dim)
      zero_point = np.zeros(scale_shape, np.int8)
      zero_point.flags.writeable = False          # shared by every tensor of this shape
      plan = (layout, quantized_dim,
              uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity), scale_shape,
              packs_in_kernel(layout, cfg.num_bits), zero_point)
    if len(_QUEUED_PLANS) > 4096:
      _QUEUED_PLANS.clear()
    _QUEUED_PLANS[key] = plan
  return plan


def get_tensor_quant_params(
    op_info: qtyping.OpInfo, tensor_quant_config: qtyping.TensorQuantizationConfig,
    tensor_content: Optional[np.ndarray] = None, tensor_qsv: Optional[dict[str, Any]] = None,
) -> qtyping.UniformQuantParams:
  """ref :34-110."""
  cfg = tensor_quant_config
  have_qsv = tensor_qsv is not None and "min" in tensor_qsv
  if not have_qsv and tensor_content is None:
    raise ValueError(
        f"{op_info.op_name}(index: {op_info.subgraph_op_index}) not found in"
        " tensor_name_to_qsv. Check if the correct calibration results are passed into the"
        " ParamsGenerator.")
  # ---- fused single-pass path (weights; min/max collected on the spot) ----
  weight_cfg = op_info.op_quant_config.weight_tensor_config
  if (not have_qsv and cfg.symmetric and cfg.num_bits in (2, 4, 8) and weight_cfg is not None
      and weight_cfg.granularity == cfg.granularity):
    queue = requant_queue.active()
    if queue is not None:
      # inside ParamsGenerator's loop: enqueue, equally shaped weights leave in one launch. What the
      # call derives from (op, granularity, bits, shape) alone is looked up, not recomputed: the
      # host side of a queued tensor has to stay below the 14 us the kernel takes for it
      plan = _queued_plan(op_info, cfg, tensor_content)
      if plan is not None:
        layout, quantized_dim, block_size, scale_shape, packs, zero_point = plan
        scale, q, slot = queue.submit(tensor_content, layout, cfg.num_bits, scale_shape, packs)
        params = qtyping.UniformQuantParams(
            scale=scale, zero_point=zero_point, num_bits=cfg.num_bits,
            symmetric=True, quantized_dimension=quantized_dim, block_size=block_size,
            quantized_data=q)
        queue.attach(slot, params)
        return params
  quantized_dim = common_utils.get_weight_quantized_dim(op_info, tensor_content, cfg.granularity)
  block_size = uniform_quantize_tensor.extract_block_size_from_granularity(cfg.granularity)
  if (not have_qsv and cfg.symmetric and cfg.num_bits in (2, 4, 8) and weight_cfg is not None
      and weight_cfg.granularity == cfg.granularity):
    layout = fused_weight_layout(tensor_content, cfg.granularity, quantized_dim)
    if layout is not None:
      scale, q = fused_symmetric_requant(tensor_content, layout, cfg.num_bits)
      scale = scale.reshape(scale_shape_for(tensor_content, cfg.granularity, quantized_dim))
      return qtyping.UniformQuantParams(
          scale=scale, zero_point=np.zeros(scale.shape, np.int8), num_bits=cfg.num_bits,
          symmetric=True, quantized_dimension=quantized_dim, block_size=block_size,
          quantized_data=q)

  # ---- general path: K1 (if needed) -> host zp/scale -> K3 ----
  if have_qsv:
    tensor_min_max = tensor_qsv
  else:
    tensor_min_max = common_quantize.init_tensor_min_max(tensor_content, op_info)
  if "min" not in tensor_min_max or "max" not in tensor_min_max:
    raise ValueError(
        "min and max must be provided to produce tensor quantization parameters. Check if the"
        " correct calibration results are passed into the ParamsGenerator.")
  zp, scale = uniform_quantize_tensor.tensor_zp_scale_from_min_max(
      tensor_min_max["min"], tensor_min_max["max"], cfg.num_bits, cfg.symmetric,
      cfg.granularity, None)
  quant_params = qtyping.UniformQuantParams(
      scale=scale, zero_point=zp, num_bits=cfg.num_bits, symmetric=cfg.symmetric,
      quantized_dimension=quantized_dim, block_size=block_size)
  if tensor_content is None:
    return quant_params
  q = uniform_quantize_tensor.uniform_quantize(
      tensor_content, quant_params, uniform_quantize_tensor.is_blockwise(cfg.granularity))
  return dataclasses.replace(quant_params, quantized_data=q)


def check_if_quantized(tensor: Any) -> bool:
  return common_quantize.check_if_quantized(tensor)


def init_qsvs(op_info: qtyping.OpInfo, graph_info: qtyping.GraphInfo,
              inputs_to_ignore: Sequence[int] | None = None,
              outputs_to_ignore: Sequence[int] | None = None, **kwargs) -> qtyping.QSV:
  """Initial QSVs: min/max of constant operands, {} for runtime tensors (ref :121-178)."""
  del kwargs
  skip_in = list(inputs_to_ignore or [])
  skip_in += [k for k, tid in enumerate(op_info.op.inputs)
              if tid != -1 and check_if_quantized(graph_info.subgraph_tensors[tid])]
  skip_out = list(outputs_to_ignore or [])
  qsvs = {}
  for ids, skip in ((op_info.op.inputs, skip_in), (op_info.op.outputs, skip_out)):
    for k, tid in enumerate(ids):
      if tid == -1 or k in skip:
        continue
      tensor = graph_info.subgraph_tensors[tid]
      data = tfl_flatbuffer_utils.get_tensor_data(tensor, graph_info.buffers)
      qsvs[tfl_flatbuffer_utils.get_tensor_name(tensor)] = common_quantize.init_tensor_min_max(
          data, op_info)
  return qsvs


def min_max_calibrate(tfl_op, graph_info: qtyping.GraphInfo,
                      tensor_content_map: MutableMapping[str, np.ndarray],
                      inputs_to_ignore: Sequence[int] | None = None,
                      outputs_to_ignore: Sequence[int] | None = None,
                      valid_range: tuple[float, float] = (-3e38, 3e38),
                      **kwargs) -> dict[str, qtyping.QSV]:
  """Per-op activation statistics for one calibration sample (ref :181-226)."""
  lo, hi = valid_range
  out = {}
  for tid in common_quantize.get_tensor_indices_requiring_calibration(
      tfl_op, graph_info, inputs_to_ignore, outputs_to_ignore):
    res = common_quantize.collect_activation_tensor_statistics(
        tid, graph_info, tensor_content_map, valid_float_range_min=lo, valid_float_range_max=hi)
    if res is not None:
      out[res[0]] = res[2]
  return out
