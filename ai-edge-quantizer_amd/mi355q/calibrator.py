"""Calibration driver: per-sample tensor contents -> model QSVs (ref: calibrator.py:191-582).

The reference obtains every intermediate tensor of a sample by running the float model in the
LiteRT interpreter with `preserve_all_tensors` (a third-party runtime that is absent here) and
then walks the graph calling the registered calibration functions. This class is that walk:
the per-sample *tensor name -> content* map comes from the caller (an interpreter the caller
owns, recorded activations, or synthetic tensors) either directly as the dataset elements or
through a `tensor_provider(signature_key, sample) -> map` callback. Statistics run on the GPU
through the registered calibration functions (`min_max_calibrate`, `gptq.calibrate`).
"""
from __future__ import annotations

import contextlib
import copy
import json
import os
from typing import Any, Callable, Iterable, Mapping, Optional

import numpy as np

from . import algorithm_manager
from . import default_policy
from . import ops
from . import qtyping
from . import recipe_manager
from . import runtime as rt
from .utils import qsv_utils
from .utils import tfl_flatbuffer_utils

_MISSING = object()
TensorProvider = Callable[[Optional[str], Any], Mapping[str, np.ndarray]]


class _QsvEncoder(json.JSONEncoder):
  def default(self, o):
    if isinstance(o, np.ndarray):
      return o.tolist()
    if isinstance(o, np.generic):
      return o.item()
    if hasattr(o, "__array__"):          # results kept in HBM (runtime.HbmArray)
      return np.asarray(o).tolist()
    return super().default(o)


class Calibrator:
  def __init__(self, float_tflite: Any, tensor_provider: Optional[TensorProvider] = None,
               qsv_update_func: Any = _MISSING, hessians: str = "consumed"):
    """hessians="consumed": a GPTQ Hessian (d x d per tensor, the dominant cost of calibration) is
    collected only for tensors some GPTQ op reads one from -- the first input of the op, ref
    algorithms/utils/common_utils.py:182-216; "all": for every runtime tensor of every GPTQ op,
    outputs included, as ref algorithms/uniform_quantize/gptq.py:84-107 does (nothing reads them)."""
    if hessians not in ("consumed", "all"):
      raise ValueError("hessians must be 'consumed' or 'all'")
    self._hessians = hessians
    self._flatbuffer_model = (float_tflite if hasattr(float_tflite, "subgraphs")
                              else tfl_flatbuffer_utils.read_model(float_tflite))
    self._tensor_provider = tensor_provider
    self._is_custom_qsv_update_func = qsv_update_func is not _MISSING
    self._qsv_update_func = (qsv_update_func if self._is_custom_qsv_update_func
                             else qsv_utils.moving_average_update)
    self._tensor_content_map: dict[str, Any] = {}
    self._model_qsvs: dict[str, qtyping.QSV] = {}
    self._metadata: dict[str, Any] = {"num_samples_calibrated": 0}
    self._recording: Optional[list] = None     # record_step(): events instead of merges
    self._plans: Optional[dict] = None         # plan_once(): per-signature op lists

  # ---- signatures ---------------------------------------------------------------------------
  def get_signature_list(self) -> list[str]:
    return [s.signatureKey.decode("utf-8") if isinstance(s.signatureKey, (bytes, bytearray))
            else s.signatureKey for s in (self._flatbuffer_model.signatureDefs or [])]

  def _main_subgraph(self, signature_key: Optional[str]) -> int:
    sigs = self._flatbuffer_model.signatureDefs or []
    if signature_key is None:
      if len(sigs) > 1:
        raise ValueError("signature_key is required for a model with several signatures")
      return sigs[0].subgraphIndex if sigs else 0
    for s in sigs:
      key = s.signatureKey.decode("utf-8") if isinstance(s.signatureKey, (bytes, bytearray)) else s.signatureKey
      if key == signature_key:
        return s.subgraphIndex
    raise ValueError(f"signature {signature_key!r} not found in the model")

  # ---- one sample (ref :501-582) ---------------------------------------------------------------
  def _update_qsvs(self, op_qsvs: dict[str, qtyping.QSV], ignore: set[str], update_func) -> set[str]:
    updated = set()
    for name, qsv in op_qsvs.items():
      if name in ignore:
        continue
      if name not in self._model_qsvs:
        self._model_qsvs[name] = qsv
      else:
        self._model_qsvs[name] = update_func(self._model_qsvs[name], qsv)
      updated.add(name)
    return updated

  def _prepare_step(self, signature_key: Optional[str], data: Any,
                    model_recipe_manager: recipe_manager.RecipeManager) -> dict:
    """First half of a calibration step: the sample's activations go to HBM and their (min, max)
    reduction is launched; nothing waits for the GPU. Steps are prepared ONE AHEAD of the walk
    (calibrate / record_steps), so the small copy that brings a sample's statistics back is
    queued in front of the previous sample's Hessian products instead of behind them, and the host
    walks sample k while the GPU still multiplies sample k - 1."""
    contents = self._tensor_provider(signature_key, data) if self._tensor_provider else data
    if not isinstance(contents, Mapping):
      raise TypeError("a calibration sample must be a {tensor name: ndarray} map (or pass a"
                      " tensor_provider that turns samples into one)")
    # activations that already live in HBM (torch tensors on the device, e.g. the outputs of a
    # float run on this GPU) stay there: statistics are taken where they are
    contents = {k: rt.resident_sample(v) for k, v in contents.items()}
    return {"contents": contents, "stage": self._stage_sample(signature_key, contents, model_recipe_manager)}

  def _finish_step(self, signature_key: Optional[str], prepared: dict,
                   model_recipe_manager: recipe_manager.RecipeManager, lazy: bool = False) -> None:
    """Second half: the ops are walked. The sample's (min, max) are on the host by now -- or,
    with `lazy` (record_steps: nothing reads a recorded statistic before wait_for_statistics()),
    are views of the pinned buffer their copy is still on its way into (NaN until it lands)."""
    self._tensor_content_map.update(prepared["contents"])
    stage = prepared["stage"]
    if stage is not None:
      arrays, dev, pinned, event, lo, hi = stage
      if lazy:
        mm = pinned.numpy()
        self._last_stage_event = event
      else:
        event.synchronize()
        mm = pinned.numpy().astype(np.float32, copy=True)
      rt.stage_calibration_step({
          id(a): {"host": a, "dev": d, "lo": lo, "hi": hi, "minmax": (mm[i, 0:1], mm[i, 1:2])}
          for i, (a, d) in enumerate(zip(arrays, dev))})
    from .algorithms.uniform_quantize import gptq
    readers = (self._plan(signature_key, model_recipe_manager)["hessian_readers"]
               if self._hessians == "consumed" else None)
    try:
      with gptq.hessians_only_for(readers):
        if self._recording is None:
          # every per-sample Hessian statistic is merged inside the walk: its tokens are copied once, where
          # they are merged (record_steps keeps the same window open until its consumer has merged)
          with gptq.borrowing():
            self._walk(signature_key, model_recipe_manager)
        else:
          self._walk(signature_key, model_recipe_manager)
    finally:
      rt.clear_calibration_step()

  def _calibrate_step(self, signature_key: Optional[str], data: Any,
                      model_recipe_manager: recipe_manager.RecipeManager) -> None:
    self._finish_step(signature_key, self._prepare_step(signature_key, data, model_recipe_manager),
                      model_recipe_manager)

  def _steps_one_ahead(self, signature_key, dataset, model_recipe_manager):
    """Prepared steps of `dataset`, each yielded after the NEXT one has been prepared."""
    waiting = None
    for data in dataset:
      nxt = self._prepare_step(signature_key, data, model_recipe_manager)
      if waiting is not None:
        yield waiting
      waiting = nxt
    if waiting is not None:
      yield waiting

  def finalize_statistics(self) -> None:
    """Statistics that still hold unprocessed samples in HBM (GPTQ Hessians collect tokens in a
    slab) are brought up to date and their staging memory is returned."""
    for qsv in self._model_qsvs.values():
      h = qsv.get("hessian") if isinstance(qsv, dict) else None
      if hasattr(h, "finalize"):
        h.finalize()
    ops.release_scratch()

  @contextlib.contextmanager
  def plan_once(self):
    """Inside this block the ops a signature calibrates (a function of the model, the recipe and
    the registry only) and the tensors to stage for them are worked out once per signature
    instead of once per sample: the scope matching and config checks behind them cost more host
    time per sample than the statistics kernels take."""
    outermost = self._plans is None
    if outermost:
      self._plans = {}
    try:
      yield self
    finally:
      if outermost:
        self._plans = None

  def _plan(self, signature_key, model_recipe_manager) -> dict:
    key = (signature_key, id(model_recipe_manager))
    plan = None if self._plans is None else self._plans.get(key)
    if plan is None:
      from .algorithms.uniform_quantize import common_quantize, naive_min_max_quantize
      from .algorithms.uniform_quantize import gptq as gptq_module
      names: dict[str, None] = {}
      ops_ = []
      readers: set[str] = set()
      for sg, graph_info, op, op_key, alg in self._scan_ops_to_calibrate(signature_key, model_recipe_manager):
        if alg == algorithm_manager.AlgorithmName.GPTQ and not isinstance(op, qtyping.IOOperator) and len(op.inputs):
          readers.add(tfl_flatbuffer_utils.get_tensor_name(sg.tensors[op.inputs[0]]))
        calibrate = algorithm_manager.get_quantization_func(alg, op_key, qtyping.QuantizeMode.CALIBRATE)
        mine = []
        for tid in common_quantize.get_tensor_indices_requiring_calibration(op, graph_info):
          tensor = sg.tensors[tid]
          if self._flatbuffer_model.buffers[tensor.buffer].data is None:
            name = tfl_flatbuffer_utils.get_tensor_name(tensor)
            names[name] = None
            mine.append(name)
        # the stock per-op function (min / max of every runtime tensor of the op) is a function of
        # these names alone: the walk takes them from here instead of deriving them per sample
        # (gptq.calibrate is the same walk plus a Hessian for the tensors its readers need: ("gptq", names))
        stock = (mine if calibrate is naive_min_max_quantize.min_max_calibrate
                 else ("gptq", mine) if calibrate is gptq_module.calibrate else None)
        ops_.append((sg, graph_info, op, op_key, alg, calibrate, stock))
      plan = {"ops": ops_, "runtime_tensors": list(names), "hessian_readers": readers}
      if self._plans is not None:
        self._plans[key] = plan
    return plan

  def _ops_to_calibrate(self, signature_key, model_recipe_manager):
    return self._plan(signature_key, model_recipe_manager)["ops"]

  def _scan_ops_to_calibrate(self, signature_key, model_recipe_manager):
    """(subgraph, graph_info, op, op_key, algorithm) of every op the recipe calibrates, in the
    reference's visiting order (main subgraph first, then subgraphs its ops invoke)."""
    codes = self._flatbuffer_model.operatorCodes
    todo = [self._main_subgraph(signature_key)]
    while todo:
      sg = self._flatbuffer_model.subgraphs[todo.pop()]
      graph_info = qtyping.GraphInfo(sg.tensors, self._flatbuffer_model.buffers)
      ops_ = list(sg.operators) + tfl_flatbuffer_utils.get_subgraph_input_output_operators(sg)
      for op in ops_:
        if isinstance(op, qtyping.IOOperator):
          op_key = op.op_key
        else:
          op_key = tfl_flatbuffer_utils.TFL_OP_CODE_TO_NAME.get(codes[op.opcodeIndex].builtinCode)
          if op_key is None:
            continue
        scope = tfl_flatbuffer_utils.get_op_scope(op, sg.tensors)
        alg, _ = model_recipe_manager.get_quantization_configs(op_key, scope)
        if alg == algorithm_manager.AlgorithmName.NO_QUANTIZE:
          continue
        if default_policy.is_non_quantizable_composite_op(op):
          continue
        yield sg, graph_info, op, op_key, alg
        todo.extend(tfl_flatbuffer_utils.get_op_side_effect_subgraphs(op))

  def _stage_sample(self, signature_key, contents, model_recipe_manager):
    """Every float32 activation the walk will read goes to HBM once and gets its (min, max)
    from one batched launch (the per-op calibration functions then find it staged). Returns
    (arrays, device tensors, pinned result, event, lo, hi) -- the result is on its way -- or None."""
    import torch
    lo, hi = -3e38, 3e38                      # the calibration functions' default valid_range
    wanted: dict[int, np.ndarray] = {}
    for name in self._plan(signature_key, model_recipe_manager)["runtime_tensors"]:
      arr = contents.get(name)
      if arr is None:
        arr = self._tensor_content_map.get(name)
      if isinstance(arr, (np.ndarray, rt.HbmArray)) and arr.dtype == np.float32 and arr.size:
        wanted[id(arr)] = arr
    if not wanted:
      return None
    rt.require_gpu()
    arrays = list(wanted.values())
    dev = [a.device_tensor.contiguous().reshape(-1) if isinstance(a, rt.HbmArray)
           else rt.to_device(np.ascontiguousarray(a).reshape(-1)) for a in arrays]
    mm = ops.act_minmax(dev, lo, hi)
    pinned = self._pinned_results(tuple(mm.shape), mm.dtype)
    pinned.fill_(float("nan"))         # (a statistic read before its copy has landed must not look like one)
    pinned.copy_(mm, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return arrays, dev, pinned, event, lo, hi

  def _pinned_results(self, shape, dtype):
    """A sample's [tensors, 2] results in page-locked memory, cut from an arena of 64 samples' worth: every sample's block
    stays alive until the statistics are read (record_steps), and a page-locked allocation of its own per sample was a
    hipHostMalloc each -- 512 of them for BASELINE config 4."""
    import torch
    n = 1
    for d in shape:
      n *= d
    arena = getattr(self, "_pinned_arena", None)
    if arena is None or arena[0].dtype != dtype or arena[1] + n > arena[0].numel():
      arena = self._pinned_arena = [torch.empty((max(64 * n, n),), dtype=dtype, pin_memory=True), 0]
    out = arena[0][arena[1]:arena[1] + n].view(shape)
    arena[1] += n
    return out

  def _walk(self, signature_key, model_recipe_manager) -> None:
    from .algorithms.uniform_quantize import common_quantize
    updated: set[str] = set()
    for _, graph_info, op, op_key, alg, calibrate, stock in self._ops_to_calibrate(signature_key, model_recipe_manager):
      if stock is None:
        op_qsvs = calibrate(op, graph_info, self._tensor_content_map)
      else:
        # min_max_calibrate with its default valid_range, minus what a tensor seen earlier in this
        # sample (the previous op's output) would compute again only to be ignored below
        op_qsvs = {}
        with_hessian = isinstance(stock, tuple)
        if with_hessian:
          from .algorithms.uniform_quantize import gptq
          readers = gptq._HESSIAN_READERS   # pylint: disable=protected-access
        for name in (stock[1] if with_hessian else stock):
          if name in updated:
            continue
          content = self._tensor_content_map[name]
          qsv = common_quantize.get_activation_min_max(content, -3e38, 3e38)
          qsv["num_samples"] = np.array(content.shape[0] if content.ndim > 0 else 1)
          if with_hessian and (readers is None or name in readers):
            qsv["hessian"] = gptq.hessian_of(content, qsv["num_samples"])      # (ref gptq.py:55-108)
          op_qsvs[name] = qsv
      if self._recording is not None:
        # sample-sharded calibration: keep this sample's statistics as events; another process
        # replays all samples' events in dataset order (distributed.calibrate_sharded)
        for name, qsv in op_qsvs.items():
          if name not in updated:
            self._recording.append((name, str(getattr(alg, "value", alg)), str(getattr(op_key, "value", op_key)), qsv))
            updated.add(name)
        continue
      update = (self._qsv_update_func if self._is_custom_qsv_update_func
                else algorithm_manager.get_update_qsv_func(alg, op_key))
      updated |= self._update_qsvs(op_qsvs, updated, update)

  def record_steps(self, signature_key: Optional[str], dataset: Iterable[Any],
                   model_recipe_manager: recipe_manager.RecipeManager):
    """record_step over a sequence of samples, the next one's statistics already on their way
    while this one is walked (yields one event list per sample, in order)."""
    from .algorithms.uniform_quantize import gptq
    for prepared in self._steps_one_ahead(signature_key, dataset, model_recipe_manager):
      self._recording = []
      # (the consumer merges this sample's Hessian statistics before it asks for the next sample: until then
      # they only refer to the sample's tokens, gptq.borrowing())
      with gptq.borrowing():
        try:
          self._finish_step(signature_key, prepared, model_recipe_manager, lazy=os.environ.get("MI355Q_CALIBRATION_LAZY", "1") != "0")
          yield self._recording
        finally:
          self._recording = None

  def wait_for_statistics(self) -> None:
    """The (min, max) of every step recorded by record_steps are on the host (their events may be
    read, pickled or replayed after this)."""
    event = getattr(self, "_last_stage_event", None)
    if event is not None:
      event.synchronize()
      self._last_stage_event = None

  def record_step(self, signature_key: Optional[str], data: Any,
                  model_recipe_manager: recipe_manager.RecipeManager) -> list[tuple]:
    """One calibration step whose per-tensor statistics are returned as
    (tensor name, algorithm, op, qsv) events instead of being merged into the model QSVs."""
    self._recording = []
    try:
      self._calibrate_step(signature_key, data, model_recipe_manager)
      return self._recording
    finally:
      self._recording = None

  def replay(self, steps: Iterable[list[tuple]],
             update_overrides: Optional[Mapping[str, Callable]] = None) -> None:
    """Merges recorded steps, in the order given, exactly as calibrating those samples here would
    have (first sighting of a tensor sets its QSV, later ones go through the op's update rule).
    `update_overrides` maps an event's algorithm tag to the rule to use instead (events whose
    large statistics travel separately, distributed.calibrate_sharded)."""
    update_overrides = update_overrides or {}
    for events in steps:
      self._metadata["num_samples_calibrated"] += 1
      for name, alg, op_key, qsv in events:
        if name not in self._model_qsvs:
          self._model_qsvs[name] = qsv
          continue
        if alg in update_overrides:
          self._model_qsvs[name] = update_overrides[alg](self._model_qsvs[name], qsv)
          continue
        update = (self._qsv_update_func if self._is_custom_qsv_update_func
                  else algorithm_manager.get_update_qsv_func(alg, qtyping.TFLOperationName(op_key)))
        self._model_qsvs[name] = update(self._model_qsvs[name], qsv)

  # ---- public API (ref :312-392) -----------------------------------------------------------------
  def calibrate(self, calibration_dataset: Mapping[Optional[str], Iterable[Any]],
                model_recipe_manager: recipe_manager.RecipeManager, cache_output: bool = False) -> None:
    del cache_output   # model outputs are the caller's: nothing is executed here
    with self.plan_once():
      for signature_key, dataset in calibration_dataset.items():
        def counted(samples):        # (a sample counts when it is taken up, before its step can fail: ref :325-330)
          for data in samples:
            self._metadata["num_samples_calibrated"] += 1
            yield data
        for prepared in self._steps_one_ahead(signature_key, counted(dataset), model_recipe_manager):
          self._finish_step(signature_key, prepared, model_recipe_manager)
    self.finalize_statistics()

  def get_model_qsvs(self) -> dict[str, qtyping.QSV]:
    return self._model_qsvs

  def reset_model_qsvs(self) -> None:
    self._model_qsvs = {}
    self._metadata = {"num_samples_calibrated": 0}

  def load_model_qsvs(self, model_qsvs: Any) -> None:
    if isinstance(model_qsvs, str):
      with open(model_qsvs, "r", encoding="utf-8") as f:
        blob = json.load(f)
      raw = blob.get("model_qsvs", blob)
      self._model_qsvs = {name: {k: np.asarray(v, np.float32 if k in ("min", "max") else None)
                                 for k, v in qsv.items()} for name, qsv in raw.items()}
      self._metadata = dict(blob.get("metadata", {}))
      self._metadata["num_samples_calibrated"] = self._metadata.get("num_samples_calibrated", 0)
    else:
      self._model_qsvs = copy.deepcopy(model_qsvs)

  def save_calibration_result(self, file_path: str, extra_metadata: Optional[dict[str, Any]] = None) -> None:
    with open(file_path, "w", encoding="utf-8") as f:
      json.dump({"model_qsvs": self._model_qsvs, "metadata": {**self._metadata, **(extra_metadata or {})}},
                f, cls=_QsvEncoder)
