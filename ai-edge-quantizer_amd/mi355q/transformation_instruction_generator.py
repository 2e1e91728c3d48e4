"""Per-tensor quantization parameters -> ordered transformation instructions.

Restatement of ref: transformation_instruction_generator.py:36-862. A tensor's parameters say,
per op touching it, which transformations that op wants (`ADD_QUANTIZE` before a consumer,
`ADD_DEQUANTIZE` after a producer, `QUANTIZE_TENSOR` for constants, ...). Two reductions turn
them into the instruction list the performer executes:

* horizontal: consumers asking for the same transformation with equal parameters at the same
  depth share one instruction (ref :247-318);
* vertical: the producer's last transformation meets each consumer group's first one -
  DQ + Q with equal parameters cancel into QUANTIZE_TENSOR (the tensor itself becomes integer),
  DQ + Q with different parameters become QUANTIZE_TENSOR + a requantizing ADD_QUANTIZE,
  DQ + NO_QUANTIZE keeps a DEQUANTIZE for just those consumers (ref :407-488).

Constant tensors that feed both quantized and unquantized consumers were marked by the
parameter generator with DUPLICATE_TENSOR / DUPLICATE_BUFFER; the trailing duplication is
dropped so the original tensor serves the last group (ref :490-535).
"""
from __future__ import annotations

import collections
import dataclasses
from typing import Any, Optional

from . import qtyping
from .utils import tfl_flatbuffer_utils

_T = qtyping.QuantTransformation
_Op = qtyping.TFLOperationName

# Ops whose output scale is tied to their input / inputs to their output / fixed by the kernel
# (what the reference derives by probing its materializers, utils/constrained_ops_utils.py).
_SCALE_CONSTRAINED_OPS = frozenset(_Op[n] for n in (
    # same as input scale
    "AVERAGE_POOL_2D", "BROADCAST_TO", "GATHER", "GATHER_ND", "MAX_POOL_2D", "MIRROR_PAD", "PAD",
    "REDUCE_MIN", "RESHAPE", "RESIZE_BILINEAR", "RESIZE_NEAREST_NEIGHBOR", "SLICE", "SPACE_TO_DEPTH",
    "SPLIT", "STRIDED_SLICE", "TRANSPOSE", "UNPACK",
    # same as output scale
    "CONCATENATION", "DYNAMIC_UPDATE_SLICE", "MAXIMUM", "PACK", "PADV2", "SELECT", "SELECT_V2",
    # fixed output scale
    "LOGISTIC", "SOFTMAX", "TANH"))


@dataclasses.dataclass(frozen=True)
class TensorGraphInfo:
  tensor_id: int
  subgraph_id: int
  producer: int            # op index, -1 for graph inputs / constants
  consumers: list[int]     # op indices, -1 for "is a graph output"


def tensor_graph_info(model: Any) -> dict[str, TensorGraphInfo]:
  """name -> where the tensor sits (a name seen again later in the model wins, ref :237-245)."""
  out: dict[str, TensorGraphInfo] = {}
  for sg_id, sg in enumerate(model.subgraphs):
    consumers = collections.defaultdict(list)
    producer = {}
    for tid in sg.outputs:
      consumers[tid].append(-1)
    for op_id, op in enumerate(sg.operators):
      for tid in op.inputs:
        consumers[tid].append(op_id)
      for tid in op.outputs:
        producer[tid] = op_id
    for tid, tensor in enumerate(sg.tensors):
      out[tfl_flatbuffer_utils.get_tensor_name(tensor)] = TensorGraphInfo(
          tid, sg_id, producer.get(tid, -1), consumers[tid])
  return out


def _mergeable(a: qtyping.OpToTensorParams, b: qtyping.OpToTensorParams, depth: int) -> bool:
  """May consumers a and b share their depth-th transformation? (ref :36-72)"""
  both_rotated = all(isinstance(p.parameters, qtyping.UniformQuantParams)
                     and p.parameters.hadamard is not None for p in (a, b))
  if both_rotated:
    return True
  return (a.parameters == b.parameters and len(a.transformations) > depth
          and len(b.transformations) > depth and a.transformations[depth] == b.transformations[depth])


def group_consumers(consumers: list[qtyping.OpToTensorParams]) -> list[list[list[int]]]:
  """levels[d] = groups (lists of consumer positions, first member first) that still share
  everything up to transformation depth d-1; levels[0] is everyone (ref :247-318)."""
  if not consumers:
    return []
  levels = [[list(range(len(consumers)))]]
  depth_max = max(len(c.transformations) for c in consumers)
  for depth in range(depth_max):
    nxt: list[list[int]] = []
    for pos, c in enumerate(consumers):
      if len(c.transformations) <= depth:
        continue
      parent = next((g for g in levels[depth] if pos in g), None)
      if parent is None:
        continue
      for g in nxt:
        if g[0] in parent and _mergeable(consumers[g[0]], c, depth):
          g.append(pos)
          break
      else:
        nxt.append([pos])
    levels.append(nxt)
  return levels


def _inst(kind, info_or_inst, consumers, parameters) -> qtyping.TransformationInst:
  return qtyping.TransformationInst(kind, info_or_inst.tensor_id, info_or_inst.producer, consumers,
                                    parameters)


def _meet(producer_rule: qtyping.TransformationInst,
          first_rules: list[qtyping.TransformationInst]) -> list[qtyping.TransformationInst]:
  """Vertical step between the producer's last rule and each consumer group's first rule."""
  out = []
  for rule in first_rules:
    from_dq = producer_rule.transformation == _T.ADD_DEQUANTIZE
    if from_dq and rule.transformation == _T.ADD_QUANTIZE:
      same = producer_rule.parameters == rule.parameters
      for c in rule.consumers:
        if c in producer_rule.consumers or not same:
          producer_rule.consumers.remove(c)
      if same:      # DQ . Q == identity on an integer tensor
        out.append(_inst(_T.QUANTIZE_TENSOR, rule, rule.consumers, rule.parameters))
      else:         # integer tensor, then requantize for these consumers
        out.append(_inst(_T.QUANTIZE_TENSOR, rule, rule.consumers, producer_rule.parameters))
        out.append(_inst(_T.ADD_QUANTIZE, rule, rule.consumers, rule.parameters))
    elif from_dq and rule.transformation == _T.NO_QUANTIZE:
      for c in rule.consumers:
        if c in producer_rule.consumers:
          producer_rule.consumers.remove(c)
      out.append(_inst(_T.ADD_DEQUANTIZE, rule, rule.consumers, producer_rule.parameters))
    else:
      out.append(rule)
  if producer_rule.consumers:
    out.insert(0, producer_rule)
  return out


def split_by_tensor_duplication(insts: qtyping.TensorTransformationInsts) -> list[list[qtyping.TransformationInst]]:
  """[instructions for the original tensor, for duplicate 1, for duplicate 2, ...] (ref :701-777)."""
  subsets: list[list[qtyping.TransformationInst]] = [[]]
  home: dict[int, int] = {}
  for inst in insts.instructions or []:
    if inst.transformation == _T.DUPLICATE_TENSOR:
      subsets.append([inst])
      for c in inst.consumers:
        if home.setdefault(c, len(subsets) - 1) != len(subsets) - 1:
          raise ValueError(f"Tensor {insts.tensor_name} : duplicate tensor should be the first"
                           " instruction for its consumers.")
    else:
      subsets[home.setdefault(inst.consumers[0], 0)].append(inst)
  return subsets


class TransformationInstructionsGenerator:
  def __init__(self, model: Optional[Any] = None):
    self.flatbuffer_model = model
    self._where = tensor_graph_info(model) if model is not None else {}

  # ---- clean-ups after the two reductions -------------------------------------------------
  @staticmethod
  def _drop_redundant_duplications(insts: list[qtyping.TransformationInst]) -> None:
    for i in range(len(insts) - 1, -1, -1):          # the last duplicate reuses the original
      if insts[i].transformation == _T.DUPLICATE_TENSOR:
        insts.pop(i)
        break
    dup_consumers = {c for inst in insts if inst.transformation == _T.DUPLICATE_TENSOR
                     for c in inst.consumers}
    if dup_consumers:                                   # a duplicated tensor owns a fresh buffer
      insts[:] = [inst for inst in insts
                  if not (inst.transformation == _T.DUPLICATE_BUFFER
                          and dup_consumers.issuperset(inst.consumers))]

  def _check_valid(self, tti: qtyping.TensorTransformationInsts) -> None:
    for subset in split_by_tensor_duplication(tti):
      kinds = {inst.transformation for inst in subset}
      if _T.NO_QUANTIZE in kinds and kinds & {_T.QUANTIZE_TENSOR, _T.ADD_DEQUANTIZE}:
        raise ValueError("Tensor %s can not be both quantized and unquantized" % tti.tensor_name)

  def _producer_is_scale_constrained(self, subgraph_id: int, op_index: int) -> bool:
    sg = self.flatbuffer_model.subgraphs[subgraph_id]
    code = self.flatbuffer_model.operatorCodes[sg.operators[op_index].opcodeIndex].builtinCode
    return tfl_flatbuffer_utils.TFL_OP_CODE_TO_NAME.get(code) in _SCALE_CONSTRAINED_OPS

  def _fold_requantize_into_free_producer(self, tti: qtyping.TensorTransformationInsts) -> None:
    """[QUANTIZE_TENSOR(p0), ADD_QUANTIZE(p1)] on the same consumers behind a producer that may
    emit any scale: quantize the tensor with p1's scale / zero point right away (ref :577-619)."""
    insts = tti.instructions
    if insts is None or len(insts) != 2:
      return
    a, b = insts
    pa, pb = a.parameters, b.parameters
    if not (isinstance(pa, qtyping.UniformQuantParams) and isinstance(pb, qtyping.UniformQuantParams)):
      return
    if not (a.transformation == _T.QUANTIZE_TENSOR and b.transformation == _T.ADD_QUANTIZE
            and a.consumers == b.consumers):
      return
    for f in dataclasses.fields(qtyping.UniformQuantParams):
      if f.name not in ("scale", "zero_point") and getattr(pa, f.name) != getattr(pb, f.name):
        return
    if a.producer == -1 or self._producer_is_scale_constrained(tti.subgraph_id, a.producer):
      return
    a.parameters = dataclasses.replace(pa, scale=pb.scale, zero_point=pb.zero_point)
    insts.pop(1)

  # ---- one tensor -------------------------------------------------------------------------
  def tensor_instructions(self, param: qtyping.TensorTransformationParams) -> qtyping.TensorTransformationInsts:
    info = self._where[param.tensor_name]
    out: list[qtyping.TransformationInst] = []
    if param.producer:
      for kind in param.producer.transformations:
        out.append(_inst(kind, info, info.consumers, param.producer.parameters))
    consumers = param.consumers or []
    levels = group_consumers(consumers)

    def rules_at(level: int) -> list[qtyping.TransformationInst]:
      rules = []
      for g in levels[level]:
        lead = consumers[g[0]]
        if len(lead.transformations) <= level - 1:
          continue
        rules.append(_inst(lead.transformations[level - 1], info,
                           [consumers[i].subgraph_op_id for i in g], lead.parameters))
      return rules

    first = rules_at(1) if len(levels) > 1 else []
    later = [r for level in range(2, len(levels)) for r in rules_at(level)]
    if out:
      out += _meet(out.pop(), first)
    else:
      out += first
    out += later
    tti = qtyping.TensorTransformationInsts(param.tensor_name, info.subgraph_id, out)
    self._drop_redundant_duplications(out)
    self._check_valid(tti)
    self._fold_requantize_into_free_producer(tti)
    return tti

  def quant_params_to_transformation_insts(self, params: dict[str, qtyping.TensorTransformationParams],
                                           flatbuffer_model: Optional[Any] = None,
                                           enable_progress_bar: Optional[bool] = None):
    del enable_progress_bar
    if flatbuffer_model is not None:
      self.flatbuffer_model = flatbuffer_model
      self._where = tensor_graph_info(flatbuffer_model)
    return {name: self.tensor_instructions(p) for name, p in params.items()}
