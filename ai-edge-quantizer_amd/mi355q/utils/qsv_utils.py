"""QSV (quantization statistics) merge rules -- host scalars, plus the GPTQ
Hessian running mean. ref: utils/qsv_utils.py:25-122.

These define what any multi-GPU exchange must reproduce:
`moving_average_update` is order dependent, so ranks all-gather per-sample
(min, max) and replay it in dataset order (mi355q.distributed).
"""
from __future__ import annotations

from typing import Any

import numpy as np

from .. import qtyping


def moving_average_update(qsv: qtyping.QSV, new_qsv: qtyping.QSV,
                          smoothing_factor: float = 0.95) -> qtyping.QSV:
  """q <- f*q + (1-f)*new for min and max; first sample is taken as is."""
  if not qsv:
    return new_qsv
  f = smoothing_factor
  return {key: f * qsv[key] + (1.0 - f) * new_qsv[key] for key in ("min", "max")}


def min_max_update(qsv: qtyping.QSV, new_qsv: qtyping.QSV) -> qtyping.QSV:
  if not qsv:
    return new_qsv
  return {"min": np.minimum(qsv["min"], new_qsv["min"]),
          "max": np.maximum(qsv["max"], new_qsv["max"])}


def _hessian_samples(qsv: qtyping.QSV):
  """The number of samples a QSV's Hessian is the mean over: its num_samples, unless samples WITHOUT a Hessian have been
  merged into the QSV since ("hessian_num_samples", see gptq_and_moving_average_update)."""
  return qsv.get("hessian_num_samples", qsv["num_samples"])


def _gptq_merge_hessian(qsv: qtyping.QSV, new_qsv: qtyping.QSV) -> tuple[Any, int]:
  """(H_cur*n_cur + H_new*n_new) / (n_cur + n_new), num_samples summed (ref :71-88).

  float64 Hessians (what calibrate() produces) are merged by
  mi355q_gptq_hessian_merge_f64 -- the same three IEEE operations per element.
  """
  n0, n1 = _hessian_samples(qsv), _hessian_samples(new_qsv)
  total = n0 + n1
  if total == 0:
    return new_qsv["hessian"], 0
  h0, h1 = qsv["hessian"], new_qsv["hessian"]
  if hasattr(h0, "absorb") and hasattr(h1, "absorb"):
    # both still collect tokens in HBM (gptq.HessianAccumulator): the weighted mean of the two is
    # the accumulator over both sets of samples; h1 is this sample's own, made for this merge
    h0.absorb(h1)
    return h0, total
  if not (hasattr(h0, "dtype") and hasattr(h1, "dtype")):
    h0, h1 = np.asarray(h0), np.asarray(h1)
  if h0.dtype == np.float64 and h1.dtype == np.float64 and h0.ndim == 2 and h0.shape == h1.shape \
      and h0.shape[0] == h0.shape[1]:
    from .. import ops
    from .. import runtime as rt
    rt.require_gpu()
    merged = ops.gptq_hessian_merge(rt.on_device(h0), float(n0), rt.on_device(h1), float(n1))
    # Hessians that were produced on the GPU stay there (see runtime.HbmArray)
    resident = isinstance(h0, rt.HbmArray) or isinstance(h1, rt.HbmArray)
    return (rt.HbmArray(merged) if resident else rt.to_numpy(merged)), total
  return (np.asarray(h0) * n0 + np.asarray(h1) * n1) / total, total


def _oscar_merge_mu2(qsv1: qtyping.QSV, qsv2: qtyping.QSV):
  """Sample-weighted mean of the second-moment vectors (ref :125-158)."""
  if "mu2" not in qsv1 and "mu2" not in qsv2:
    return None, 0
  if "mu2" not in qsv1:
    return qsv2.get("mu2"), qsv2.get("num_samples", 0)
  if "mu2" not in qsv2:
    return qsv1.get("mu2"), qsv1.get("num_samples", 0)
  n0, n1 = qsv1.get("num_samples", 0), qsv2.get("num_samples", 0)
  total = n0 + n1
  if total == 0:
    return qsv2["mu2"], 0
  return (qsv1["mu2"] * n0 + qsv2["mu2"] * n1) / total, total


def oscar_and_moving_average_update(qsv: qtyping.QSV, new_qsv: qtyping.QSV) -> qtyping.QSV:
  """EMA for min/max + sample-weighted mean for mu2 (ref :161-171). O(channels) host math."""
  if not qsv:
    return new_qsv
  out = moving_average_update(qsv, new_qsv)
  out["mu2"], out["num_samples"] = _oscar_merge_mu2(qsv, new_qsv)
  return out


def gptq_and_moving_average_update(qsv: qtyping.QSV, new_qsv: qtyping.QSV) -> qtyping.QSV:
  """EMA for min/max + sample-weighted mean for the Hessian (ref :71-102)."""
  if not qsv:
    return new_qsv
  out = moving_average_update(qsv, new_qsv)
  if "hessian" not in qsv or "hessian" not in new_qsv:
    # Neither side: a tensor no op reads a Hessian from (Calibrator(hessians="consumed")): min / max /
    # count only. One side only (a resumed calibration whose earlier result came from hessians="all",
    # an earlier build or the reference, where every runtime tensor carries one; or a tensor read by a
    # GPTQ op under one signature's plan only): the samples of the side without a Hessian cannot
    # weigh in, so the existing Hessian is kept as the mean over ITS samples, whose number travels with it
    # ("hessian_num_samples"), and the counts add.
    out["num_samples"] = qsv.get("num_samples", 0) + new_qsv.get("num_samples", 0)
    side = qsv if "hessian" in qsv else new_qsv if "hessian" in new_qsv else None
    if side is not None:
      # the Hessian stays the mean over ITS samples, and says so: a later merge weighs it by that count, not by the
      # QSV's (which now includes samples that never contributed to it). The reference raises KeyError here.
      out["hessian"] = side["hessian"]
      out["hessian_num_samples"] = _hessian_samples(side) if "num_samples" in side else 0
    return out
  out["hessian"], with_hessian = _gptq_merge_hessian(qsv, new_qsv)
  out["num_samples"] = qsv["num_samples"] + new_qsv["num_samples"]
  if "hessian_num_samples" in qsv or "hessian_num_samples" in new_qsv:
    out["hessian_num_samples"] = with_hessian
  return out


# Rules Calibrator.replay may advance for a whole block of samples at once (calibrator.StepBlock): "ema" = min / max
# only, "count" = min / max plus the summed num_samples beside an untouched "hessian" entry. Anything without the
# attribute -- a partial with another smoothing factor, a user's rule, OSCAR's -- is called sample by sample.
moving_average_update.block_mode = "ema"
gptq_and_moving_average_update.block_mode = "count"
