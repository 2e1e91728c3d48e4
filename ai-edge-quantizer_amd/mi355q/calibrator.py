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


def _plain_min_max(qsv) -> bool:
  """A QSV whose min / max are the one-element float32 arrays calibration itself produces (a loaded result may hold
  anything: such tensors go through the update rule itself)."""
  if not isinstance(qsv, dict):
    return False
  for key in ("min", "max"):
    v = qsv.get(key)
    if not (isinstance(v, np.ndarray) and v.dtype == np.float32 and v.size == 1):
      return False
  return True


def _describe(v) -> Optional[list]:
  """[entry, flat float32 device tensor, device pointer, element count, leading dimension (num_samples, ref
  common_quantize.py:1447-1449), rank, tokens view (filled on demand), bytes uploaded for it] of one sample entry that
  the block path covers: a float32 (or bfloat16) array with elements, in HBM or on the host. None for anything else."""
  import torch
  uploaded = 0
  if isinstance(v, rt.HbmArray):
    shape, t = v.shape, v.device_tensor
  elif isinstance(v, torch.Tensor):
    if v.is_cuda and v.dtype == torch.float32 and v.is_contiguous():
      # the common entry -- an activation a float run left in HBM -- by the shortest way: three calls, no new tensor
      n = v.numel()
      if not n:
        return None
      shape = v.shape
      return [v, v, v.data_ptr(), n, shape[0] if len(shape) else 1, len(shape), None, 0]
    shape, t = tuple(v.shape), v.detach()
    if t.dtype == torch.bfloat16:         # (as runtime.resident_sample: widening is exact)
      t = t.float()
    if not t.is_cuda:
      if t.dtype != torch.float32 or not t.numel():
        return None
      t = rt.to_device(np.ascontiguousarray(t.numpy()).reshape(-1))
      uploaded = t.numel() * 4
  elif isinstance(v, np.ndarray):
    if v.dtype != np.float32 or not v.size:
      return None
    shape, t = v.shape, rt.to_device(np.ascontiguousarray(v).reshape(-1))
    uploaded = t.numel() * 4
  else:
    return None
  if t.dtype != torch.float32 or not t.numel():
    return None
  t = t.contiguous()
  if not uploaded and isinstance(v, torch.Tensor) and (v.dtype != torch.float32 or t.data_ptr() != v.data_ptr()):
    # a float32 copy made in HBM (bfloat16 widened, strides gathered): given back with its block, like an upload --
    # kept until the dataset is through it would be a second copy of every such sample
    uploaded = t.numel() * 4
  flat = t.reshape(-1)
  return [v, t.reshape(shape) if uploaded else t, flat.data_ptr(), flat.numel(), shape[0] if len(shape) else 1,
          len(shape), None, uploaded]


class StepBlock:
  """The statistics of K consecutive samples of one signature, taken by ONE launch (Calibrator.record_blocks).

  What K calls of record_step would have returned as K lists of (name, algorithm, op, qsv) events, kept as arrays:
    slots        ((tensor name, algorithm, op), ...) in the order the op walk first meets each runtime tensor
                 (ref calibrator.py:567-582: a tensor updated once per sample, by the first op that lists it)
    stats        float32 [K, T, 2]  (min, max) per sample and slot (a view of page-locked memory the copy is still
                 writing until Calibrator.wait_for_statistics(); pickles as a plain array)
    num_samples  int64 [K, T]       each content's leading dimension (ref common_quantize.py:1447-1449)
    ndims        (T,)               rank of each content: min / max are shaped (1,) * ndim (ref :1380)
    hessian_dims {slot index: d}    slots whose GPTQ Hessian was set aside into a running accumulator in HBM
  `first` is the dataset index of the block's first sample (the replay order key)."""
  __slots__ = ("slots", "stats", "num_samples", "ndims", "hessian_dims", "first", "with_hessian")

  def __init__(self, slots, stats, num_samples, ndims, hessian_dims=None, first=0, with_hessian=()):
    self.slots, self.stats, self.num_samples, self.ndims = slots, stats, num_samples, ndims
    self.hessian_dims, self.first = dict(hessian_dims or {}), first
    self.with_hessian = frozenset(with_hessian)      # slots whose samples went into a Hessian statistic (tagged or not)

  def __len__(self) -> int:
    return int(self.stats.shape[0])

  def __reduce__(self):
    return (StepBlock, (self.slots, np.array(self.stats, np.float32), np.array(self.num_samples), self.ndims,
                        self.hessian_dims, self.first, tuple(sorted(self.with_hessian))))

  def qsv(self, k: int, t: int) -> dict:
    """The event record of sample k, slot t, as the per-sample walk builds it."""
    shape = (1,) * self.ndims[t]
    out = {"min": self.stats[k, t, 0:1].reshape(shape), "max": self.stats[k, t, 1:2].reshape(shape),
           "num_samples": np.array(self.num_samples[k, t])}
    if t in self.hessian_dims:
      out["hessian_dim"] = self.hessian_dims[t]
    return out

  def events(self, k: int) -> list[tuple]:
    return [(name, alg, op_key, self.qsv(k, t)) for t, (name, alg, op_key) in enumerate(self.slots)]


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
    self._raw_carry: dict[str, Any] = {}       # record_blocks(): the samples' entries as the caller handed them over
    self._described: dict[int, list] = {}      # id(entry) -> [entry, device tensor, pointer, numel, leading dim, ndim, tokens]
    self._new_hessians: dict[str, Any] = {}    # calibrate() over blocks: accumulators of tensors not merged yet
    self._blocks_off = False                   # calibrate() over blocks: the rest of the dataset takes the per-sample walk

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

  def _steps_one_ahead(self, signature_key, dataset, model_recipe_manager, taken_up: Optional[Callable[[], None]] = None):
    """Prepared steps of `dataset`, each yielded after the NEXT one has been prepared. Being one ahead does not show:
    when the dataset or the next sample's preparation raises, the step that waits is handed out (and merged by the
    caller) first, and `taken_up()` is called once per sample right before its step is handed out or its preparation's
    exception surfaces -- a sample counts when it is taken up, every sample before it is merged (ref :325-330)."""
    waiting = None
    it = iter(dataset)
    while True:
      failed, counts = None, False
      try:
        data = next(it)
      except StopIteration:
        break
      except Exception as e:  # pylint: disable=broad-exception-caught
        failed = e
      if failed is None:
        try:
          nxt = self._prepare_step(signature_key, data, model_recipe_manager)
        except Exception as e:  # pylint: disable=broad-exception-caught
          failed, counts = e, True
      if waiting is not None:
        if taken_up is not None:
          taken_up()
        yield waiting
        waiting = None
      if failed is not None:
        if counts and taken_up is not None:
          taken_up()
        raise failed
      waiting = nxt
    if waiting is not None:
      if taken_up is not None:
        taken_up()
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
      plan = {"ops": ops_, "runtime_tensors": list(names), "hessian_readers": readers,
              "slots": self._slots_of(ops_, readers)}
      if self._plans is not None:
        self._plans[key] = plan
    return plan

  def _slots_of(self, ops_, readers) -> Optional[list]:
    """What one sample's op walk records when every op uses a stock calibration function, as a static list: (tensor
    name, algorithm, op, Hessian wanted, algorithm key, op key) per runtime tensor, in the order the walk first meets
    it -- a tensor is updated once per sample, by the first op that lists it (ref calibrator.py:567-582), and a GPTQ
    op adds a Hessian for the tensors some op reads one from (_walk). None when an op brings its own function."""
    wanted = readers if self._hessians == "consumed" else None
    slots, seen = [], set()
    for _, _, _, op_key, alg, _, stock in ops_:
      if stock is None:
        return None
      with_hessian = isinstance(stock, tuple)
      for name in (stock[1] if with_hessian else stock):
        if name not in seen:
          seen.add(name)
          slots.append((name, str(getattr(alg, "value", alg)), str(getattr(op_key, "value", op_key)),
                        with_hessian and (wanted is None or name in wanted), alg, op_key))
    return slots

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

  # ---- K samples per launch -------------------------------------------------------------------------
  # The per-sample walk above costs ~10 us of interpreter per (sample, tensor) -- plan lookup, staging records, one small
  # dict and three small arrays per statistic -- against 0.7 us of kernel for a 4 MiB activation: BASELINE config 4 ran at
  # 0.05 of HBM end to end with a 0.78 kernel. When every op of the signature uses a stock calibration function the
  # walk's outcome is a static list of slots (_slots_of), so K samples become ONE pointer table, ONE launch and ONE copy
  # back, and their K x T statistics are replayed as array operations in dataset order (replay).
  BLOCK_SAMPLES = int(os.environ.get("MI355Q_CALIBRATION_BLOCK", 64))
  BLOCK_ENTRIES = 8192           # (sample, tensor) pairs per launch: one pinned pointer-table slot (ops._TABLE_CAPACITY)
  BLOCK_UPLOAD_BYTES = 2 << 30   # host arrays of one block that sit in HBM at a time

  def samples_per_launch(self, signature_key, dataset, model_recipe_manager, asked: Optional[int] = None) -> int:
    """How many samples one launch may cover. 1 = the per-sample walk: always for signatures with an op that brings its
    own calibration function or a resumed non-accumulating Hessian, and -- unless the caller asks for K > 1 and thereby
    asserts that the samples do not share buffers that are overwritten between them -- for datasets that are produced
    while they are walked (a generator, a tensor_provider): K samples are only read after the K-th has been pulled."""
    if asked is not None and asked <= 1:
      return 1
    slots = self._plan(signature_key, model_recipe_manager)["slots"]
    if not slots:
      return 1
    if asked is None and (self._tensor_provider is not None
                          or not (hasattr(dataset, "__len__") and hasattr(dataset, "__getitem__"))):
      return 1
    for name, _, _, with_hessian, _, _ in slots:
      cur = self._model_qsvs.get(name)
      if with_hessian and isinstance(cur, dict) and "hessian" in cur and not hasattr(cur["hessian"], "add_block"):
        return 1
    return max(1, min(asked or self.BLOCK_SAMPLES, self.BLOCK_ENTRIES // len(slots)))

  def _blocks_keep_hessians(self, signature_key, model_recipe_manager) -> bool:
    """Whether calibrate()'s own block merge (_replay_block with no overrides) hands every Hessian slot's accumulator
    to the QSV. It does so on the array path only: the slot's update rule must be one that advances over whole blocks
    (`block_mode`: the stock rules of utils/qsv_utils.py) and the stored QSV must be calibration's own float32 pair.
    Any other rule -- a custom qsv_update_func, a wrapper or functools.partial around the stock one -- is fed the
    block's samples as events WITHOUT a 'hessian' (the tokens went into the accumulator), so such signatures take the
    per-sample walk, whose events carry the Hessian (ref calibrator.py:567-582, utils/qsv_utils.py:90-122)."""
    for name, _, _, with_hessian, alg, op_key in self._plan(signature_key, model_recipe_manager)["slots"] or ():
      if not with_hessian:
        continue
      update = (self._qsv_update_func if self._is_custom_qsv_update_func
                else algorithm_manager.get_update_qsv_func(alg, op_key))
      if getattr(update, "block_mode", None) is None:
        return False
      cur = self._model_qsvs.get(name)
      if cur is not None and not _plain_min_max(cur):
        return False
    return True

  def _sync_carry(self) -> None:
    """What record_blocks saw last of every tensor joins the content map of the per-sample walk."""
    if self._raw_carry:
      self._tensor_content_map.update({k: rt.resident_sample(v) for k, v in self._raw_carry.items()})
      self._raw_carry.clear()

  def _seed_carry(self) -> None:
    """The block path's view of "the last content of every tensor" starts from the per-sample walk's: the two are ONE
    store in the reference (ref calibrator.py:529: a sample that omits a tensor reuses what an earlier sample left)."""
    self._raw_carry.clear()
    self._raw_carry.update(self._tensor_content_map)

  def _gather_block(self, slots, samples, limit: int, first: int, hessian_sink, hessian_tag):
    """(StepBlock, samples taken, whether the sample after them does not qualify) from the head of `samples` (a list).
    A sample does not qualify when one of its tensors is missing, not float32, empty or not an array: the per-sample
    walk deals with it and raises what the reference raises. A block also ends when its host arrays fill the upload
    budget."""
    import torch
    n_slots = len(slots)
    names = [s[0] for s in slots]
    hslots = [t for t, s in enumerate(slots) if s[3]]
    if (hslots and hessian_sink is None) or self._blocks_off:
      # nowhere to put the Hessians' samples (the per-sample walk returns them), or calibrate() has seen the walk leave
      # a QSV its block merge would not take
      return None, 0, True
    described, carry = self._described, self._raw_carry
    pointers, lengths, leading = [], [], []
    tokens = {t: ([], []) for t in hslots}
    ndims = None
    uploaded, uploaded_keys = 0, []
    taken, unfit = 0, False
    for sample in samples[:limit]:
      unfit = True                    # (cleared at the end of the loop body: every early exit below is a misfit)
      if not isinstance(sample, Mapping):
        break
      carry.update(sample)
      row = []
      for name in names:
        v = carry.get(name)
        rec = described.get(id(v))
        if rec is None or rec[0] is not v:
          rec = _describe(v)
          if rec is None:
            break
          described[id(v)] = rec
          if rec[7]:
            uploaded += rec[7]
            uploaded_keys.append(id(v))
        row.append(rec)
      if len(row) != n_slots:
        break
      dims = tuple([r[5] for r in row])
      if ndims is None:
        ndims = dims
      elif dims != ndims:
        break
      if hslots and any(row[t][5] < 1 for t in hslots):
        break
      pointers.extend([r[2] for r in row])
      lengths.extend([r[3] for r in row])
      leading.extend([r[4] for r in row])
      for t in hslots:
        r = row[t]
        if r[6] is None:
          r[6] = r[1].reshape(-1, r[1].shape[-1])
        tokens[t][0].append(r[6])
        tokens[t][1].append(r[4])
      taken += 1
      unfit = False
      if uploaded > self.BLOCK_UPLOAD_BYTES:
        break
    if not taken:
      return None, 0, unfit
    rt.require_gpu()
    mm = ops.act_minmax_entries(pointers, lengths, -3e38, 3e38)     # the calibration functions' default valid_range
    pinned = self._pinned_results((taken, n_slots, 2), mm.dtype)
    pinned.fill_(float("nan"))         # (a statistic read before its copy has landed must not look like one)
    pinned.copy_(mm.view(taken, n_slots, 2), non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    self._last_stage_event = event
    hessian_dims = {}
    for t in hslots:
      xs, ns = tokens[t]
      hessian_dims[t] = int(xs[0].shape[1])
      hessian_sink(names[t], xs, ns)
    block = StepBlock(tuple((s[0], hessian_tag if (t in hessian_dims and hessian_tag) else s[1], s[2])
                            for t, s in enumerate(slots)),
                      pinned.numpy(), np.array(leading, np.int64).reshape(taken, n_slots), ndims,
                      hessian_dims if hessian_tag else None, first, tuple(hessian_dims))
    for key in uploaded_keys:          # the uploads of this block: their launch is queued, the allocator orders the reuse
      described.pop(key, None)
    return block, taken, unfit

  def record_blocks(self, signature_key: Optional[str], dataset: Iterable[Any],
                    model_recipe_manager: recipe_manager.RecipeManager, limit: int, first: int = 0,
                    hessian_sink: Optional[Callable] = None, hessian_tag: Optional[str] = None,
                    fallback: Optional[Callable] = None, taken_up: Optional[Callable[[int], None]] = None):
    """record_step over a dataset, up to `limit` samples per launch: yields (dataset index of the first sample covered,
    StepBlock) -- or (index, what `fallback(sample)` returned; default: record_step's event list) for a sample the
    block path does not cover. `hessian_sink(name, [tokens float32 [t, d] per sample], [num_samples per sample])`
    receives the samples of every tensor a GPTQ op reads a Hessian from, behind the block's launch; such slots are
    tagged `hessian_tag` in the block when one is given. `taken_up(n)` is told about samples before their step can
    fail (a sample counts when it is taken up: ref calibrator.py:325-330). An exception of the dataset's iterator
    surfaces after the samples pulled before it have been processed."""
    import collections
    slots = self._plan(signature_key, model_recipe_manager)["slots"]
    if fallback is None:
      fallback = lambda data: self.record_step(signature_key, data, model_recipe_manager)
    it = iter(dataset)
    buf: collections.deque = collections.deque()
    failed = None
    index = first
    self._seed_carry()
    try:
      while True:
        while failed is None and len(buf) < limit:
          try:
            buf.append(next(it))
          except StopIteration:
            failed = StopIteration
          except Exception as e:  # pylint: disable=broad-exception-caught
            failed = e
        if not buf:
          break
        head = list(buf)
        block, taken, unfit = self._gather_block(slots, head, limit, index, hessian_sink, hessian_tag)
        if block is not None:
          if taken_up is not None:
            taken_up(taken)
          for _ in range(taken):
            buf.popleft()
          yield index, block
          index += taken
        if unfit:
          # the sample at the head did not qualify: the per-sample walk handles it (and raises what it raises)
          self._sync_carry()
          data = buf.popleft()
          if taken_up is not None:
            taken_up(1)
          yield index, fallback(data)
          index += 1
          self._seed_carry()
      if failed is not None and failed is not StopIteration:
        raise failed
    finally:
      self._sync_carry()       # (a later per-sample step, or the next call, starts from what these samples left)
      self._described.clear()

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

  def replay(self, steps: Iterable[Any],
             update_overrides: Optional[Mapping[str, Callable]] = None) -> None:
    """Merges recorded steps, in the order given, exactly as calibrating those samples here would
    have (first sighting of a tensor sets its QSV, later ones go through the op's update rule).
    A step is one sample's event list (record_step) or a StepBlock (record_blocks: its samples in order).
    `update_overrides` maps an event's algorithm tag to the rule to use instead (events whose
    large statistics travel separately, distributed.calibrate_sharded)."""
    update_overrides = update_overrides or {}
    for events in steps:
      if isinstance(events, StepBlock):
        self._replay_block(events, update_overrides)
        continue
      self._metadata["num_samples_calibrated"] += 1
      self._replay_events(events, update_overrides)

  def _replay_events(self, events, update_overrides) -> None:
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

  def _replay_block(self, block: StepBlock, update_overrides, count: bool = True) -> None:
    """The K samples of a block through the update rules, sample by sample in effect. Slots whose rule is one of the
    stock ones -- the moving average of min / max (ref utils/qsv_utils.py:43-68), with or without the sample count
    GPTQ keeps beside it (:90-102) -- advance together: one array expression per sample over all such slots, the same
    three float32 operations per element as the rule's own `f * q + (1 - f) * new`. Any other rule (a custom
    qsv_update_func, OSCAR's) sees the block's samples as the events they stand for."""
    k_samples, n_slots = block.stats.shape[0], len(block.slots)
    if count:
      self._metadata["num_samples_calibrated"] += k_samples
    fast, modes, slow = [], [], []
    for t, (name, alg, op_key) in enumerate(block.slots):
      if alg in update_overrides:
        update = update_overrides[alg]
      elif self._is_custom_qsv_update_func:
        update = self._qsv_update_func
      else:
        update = algorithm_manager.get_update_qsv_func(alg, qtyping.TFLOperationName(op_key))
      mode = getattr(update, "block_mode", None)
      cur = self._model_qsvs.get(name)
      if mode is None or not (cur is None or _plain_min_max(cur)):
        slow.append((t, update))
      else:
        fast.append(t)
        modes.append(mode)
    if fast:
      idx = np.array(fast, np.intp)
      stats = block.stats[:, idx, :]                       # [K, F, 2] float32
      state = np.empty((len(fast), 2), np.float32)
      has = np.zeros(len(fast), bool)
      for j, t in enumerate(fast):
        cur = self._model_qsvs.get(block.slots[t][0])
        if cur is not None:
          has[j] = True
          state[j, 0], state[j, 1] = cur["min"].reshape(-1)[0], cur["max"].reshape(-1)[0]
      f = 0.95                                             # moving_average_update's smoothing_factor
      if has.all():
        state = f * state + (1.0 - f) * stats[0]
      elif has.any():
        state[has] = f * state[has] + (1.0 - f) * stats[0][has]
        state[~has] = stats[0][~has]
      else:
        state = stats[0].copy()
      for k in range(1, k_samples):
        state = f * state + (1.0 - f) * stats[k]
      state = np.ascontiguousarray(state, np.float32)
      for j, t in enumerate(fast):
        name = block.slots[t][0]
        shape = (1,) * block.ndims[t]
        cur = self._model_qsvs.get(name)
        out = {"min": state[j, 0:1].reshape(shape), "max": state[j, 1:2].reshape(shape)}
        merged = cur is not None or k_samples > 1          # (a lone first sighting is stored as the event itself)
        if modes[j] == "ema" and merged:
          self._model_qsvs[name] = out
          continue
        total = block.num_samples[:, t].sum()
        if cur is not None:
          total = cur.get("num_samples", 0) + total
        out["num_samples"] = total if merged else np.array(block.num_samples[0, t])
        if t in block.hessian_dims:
          out["hessian_dim"] = block.hessian_dims[t]
        if cur is not None:
          for key in ("hessian", "hessian_dim"):
            if key in cur:
              out[key] = cur[key]
        fresh = self._new_hessians.pop(name, None)         # calibrate(): this block's samples went into it already
        if fresh is not None and "hessian" not in out:
          out["hessian"] = fresh
        if cur is not None and "hessian" in out and t not in block.hessian_dims:
          # the Hessian's own sample count, where it is not the QSV's (one side of a merge had no Hessian:
          # utils/qsv_utils.gptq_and_moving_average_update keeps the two counts apart)
          mine = block.num_samples[:, t].sum()
          if t not in block.with_hessian:                      # these samples brought none: the Hessian stays what it was
            out["hessian_num_samples"] = cur.get("hessian_num_samples", cur.get("num_samples", 0))
          elif "hessian" not in cur:                           # the QSV had none so far
            out["hessian_num_samples"] = mine
          elif "hessian_num_samples" in cur:
            out["hessian_num_samples"] = cur["hessian_num_samples"] + mine
        self._model_qsvs[name] = out
    if slow:
      for k in range(k_samples):
        self._replay_events([(block.slots[t][0], block.slots[t][1], block.slots[t][2], block.qsv(k, t))
                             for t, _ in slow], update_overrides)

  # ---- public API (ref :312-392) -----------------------------------------------------------------
  def calibrate(self, calibration_dataset: Mapping[Optional[str], Iterable[Any]],
                model_recipe_manager: recipe_manager.RecipeManager, cache_output: bool = False,
                samples_per_launch: Optional[int] = None) -> None:
    """`samples_per_launch`: None = up to BLOCK_SAMPLES samples share a launch when the dataset is a sequence that
    exists before the call (see samples_per_launch()); 1 = the per-sample walk; K > 1 = blocks of K also for a
    dataset that is generated while it is read (the caller vouches that K consecutive samples do not alias)."""
    del cache_output   # model outputs are the caller's: nothing is executed here
    with self.plan_once():
      for signature_key, dataset in calibration_dataset.items():
        limit = self.samples_per_launch(signature_key, dataset, model_recipe_manager, samples_per_launch)
        if limit > 1 and not self._blocks_keep_hessians(signature_key, model_recipe_manager):
          limit = 1
        if limit > 1:
          self._calibrate_in_blocks(signature_key, dataset, model_recipe_manager, limit)
          continue
        def taken_up():              # (a sample counts when it is taken up, before its step can fail: ref :325-330)
          self._metadata["num_samples_calibrated"] += 1
        for prepared in self._steps_one_ahead(signature_key, dataset, model_recipe_manager, taken_up):
          self._finish_step(signature_key, prepared, model_recipe_manager)
    self.finalize_statistics()

  def _calibrate_in_blocks(self, signature_key, dataset, model_recipe_manager, limit: int) -> None:
    """calibrate() of one signature, K samples per launch. The blocks' statistics are merged when the dataset is
    through -- or earlier when a sample needs the per-sample walk, which reads and writes the model QSVs itself, or
    the dataset raises -- so the host gathers block b + 1 while the GPU reduces block b."""
    from .algorithms.uniform_quantize import gptq
    waiting: list[StepBlock] = []

    def merge_waiting() -> None:
      if waiting:
        self.wait_for_statistics()
        for block in waiting:
          self._replay_block(block, {}, count=False)
        del waiting[:]

    def taken_up(n: int) -> None:
      self._metadata["num_samples_calibrated"] += n

    def hessian_sink(name, xs, ns) -> None:
      acc = self._new_hessians.get(name)
      if acc is None:
        cur = self._model_qsvs.get(name)
        acc = cur.get("hessian") if isinstance(cur, dict) else None
        if acc is not None and not hasattr(acc, "add_block"):
          # a finished float64 Hessian: a sample the blocks do not cover (float64 tokens, say) went through the
          # per-sample walk, whose merge left a plain array. Nothing waits to be merged at this point (per_sample
          # merges first), so the QSV's count is the Hessian's: it becomes the float64 share of an accumulator
          # (the reference's weighted mean, utils/qsv_utils.py:71-88) and the blocks carry on.
          acc = cur["hessian"] = gptq.HessianAccumulator.resumed(acc, float(qsv_utils._hessian_samples(cur)))  # pylint: disable=protected-access
        if acc is None:
          acc = self._new_hessians[name] = gptq.HessianAccumulator(int(xs[0].shape[1]))
      acc.add_block(xs, ns)

    def per_sample(data) -> None:
      merge_waiting()
      self._calibrate_step(signature_key, data, model_recipe_manager)
      # what the walk left may be a QSV the array path of _replay_block does not take (see _blocks_keep_hessians):
      # the rest of the dataset then goes sample by sample too
      self._blocks_off = not self._blocks_keep_hessians(signature_key, model_recipe_manager)

    try:
      for _, item in self.record_blocks(signature_key, dataset, model_recipe_manager, limit, hessian_sink=hessian_sink,
                                        fallback=per_sample, taken_up=taken_up):
        if isinstance(item, StepBlock):
          waiting.append(item)
    finally:
      self._blocks_off = False
      merge_waiting()
      self._new_hessians.clear()

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
