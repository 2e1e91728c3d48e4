"""Multi-GPU layer: one process per GPU, torch.distributed (RCCL over xGMI).

The reference has no distributed code; this module defines what the build adds
and what it must reproduce:

  * weight requantization shards WHOLE tensor-buffers across ranks -- no
    collective on the data path, results are gathered to rank 0;
  * activation calibration shards SAMPLES contiguously; every rank computes
    per-sample (min, max) pairs on its GPU, ranks all-gather the pairs and every
    rank replays the reference's order-dependent update
    (qsv_utils.moving_average_update, ref: utils/qsv_utils.py:43-68, applied per
    sample in dataset order by calibrator.py:395-421) on the host. A plain
    all-reduce(min/max) is offered only for `min_max_update`
    (ref: utils/qsv_utils.py:105-122), for which it is exact;
  * the GPTQ Hessian is a sample-weighted mean (ref: utils/qsv_utils.py:71-88):
    ranks all-reduce(sum) num_samples-weighted partial Hessians and divide.

Transport: with one process per GPU (process group backend "nccl") every data collective goes
through libmi355q's own RCCL entry points (include/mi355q.h: mi355q_allgather_minmax,
mi355q_allreduce_minmax_f32, mi355q_allreduce_sum_f64, mi355q_allreduce_hessian_f64) on a
communicator created from a ncclUniqueId broadcast over the existing rendezvous; torch.distributed
is the rendezvous, the barrier and the object gather of results. With the "gloo" backend (CPU
tests, or several test ranks sharing one GPU -- RCCL refuses duplicate devices) the same
functions move host tensors through torch.distributed instead.

N > 1 over RCCL on a box that shows ONE GPU (`one_gpu_ranks_env`): RCCL's refusal compares host hash and bus id, and the host
hash can be set per process (NCCL_HOSTID). Ranks that claim different hosts are peers over RCCL's NET transport (sockets on
the loopback interface, staged through host memory) and may all sit on cuda:0: not xGMI, but every "nccl" branch of this
module and every RCCL entry point of libmi355q then runs with real peers (tests/test_gpu_distributed.py, bench.py
MI355Q_BENCH_ONE_GPU_HOSTS=1).
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Callable, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .utils import qsv_utils


# ------------------------------------------------------------------ setup ---
def one_gpu_ranks_env(rank: int) -> dict[str, str]:
  """Environment under which rank `rank` of an N > 1 "nccl" process group may share cuda:0 with its peers: every rank names
  a host of its own (RCCL then sees no duplicate device), the peers meet over sockets on the loopback interface. Set it
  before the first RCCL call of the process (LOCAL_RANK = 0: the device index is not the rank)."""
  return dict(NCCL_HOSTID=f"mi355q-one-gpu-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket",
              NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", LOCAL_RANK="0")


def init(backend: Optional[str] = None) -> tuple[int, int]:
  """Initialises the default process group from torchrun's environment.

  Returns (rank, world_size). Single-process runs need no group.
  """
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  if world == 1:
    return 0, 1
  if not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    use_gpu = torch.cuda.is_available()
    # MI355Q_DIST_BACKEND=gloo: several ranks on a box with fewer GPUs (tests): they share devices
    backend = backend or os.environ.get("MI355Q_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if use_gpu:
      local = int(os.environ.get("LOCAL_RANK", "0"))
      torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
    dist.init_process_group(backend)
  return rank, world


def _world(group=None) -> tuple[int, int]:
  if not dist.is_available() or not dist.is_initialized():
    return 0, 1
  return dist.get_rank(group), dist.get_world_size(group)


def _comm_device(group=None) -> torch.device:
  if dist.is_initialized() and dist.get_backend(group) == "nccl":
    return torch.device("cuda", torch.cuda.current_device())
  return torch.device("cpu")


# --------------------------------------------------- RCCL via the C ABI ---
# (group object or None for the default group, communicator): identity of the group object is
# checked, not its id() -- Python reuses ids of destroyed groups, and a stale communicator would
# talk to the wrong peers
_COMMS: list[tuple[Any, ctypes.c_void_p]] = []
_ATEXIT = [False]


def rccl_comm(group=None) -> Optional[ctypes.c_void_p]:
  """libmi355q's RCCL communicator for `group` (created on first use), or None when the ranks
  of the group cannot have one: no process group, or a host transport ("gloo")."""
  if not dist.is_available() or not dist.is_initialized() or dist.get_backend(group) != "nccl":
    return None
  for g, comm in _COMMS:
    if g is group:
      return comm
  comm = new_rccl_comm(dist.get_rank(group), dist.get_world_size(group),
                       lambda uid: _broadcast_bytes(uid, group))
  _COMMS.append((group, comm))     # the entry keeps `group` alive, so its identity stays unique
  if not _ATEXIT[0]:
    import atexit
    atexit.register(_destroy_rccl_comms_at_exit)
    _ATEXIT[0] = True
  return comm


def _broadcast_bytes(payload: Optional[bytes], group=None) -> bytes:
  box = [payload]
  src = 0 if group is None else dist.get_global_rank(group, 0)
  dist.broadcast_object_list(box, src=src, group=group)
  return box[0]


def new_rccl_comm(rank: int, world: int, broadcast: Callable[[Optional[bytes]], bytes]) -> ctypes.c_void_p:
  """ncclUniqueId from rank 0 -> `broadcast` (any rendezvous) -> ncclCommInitRank on every rank,
  all through the C ABI. The calling process's current GPU becomes the communicator's device."""
  from . import _ffi
  from . import runtime as rt
  rt.require_gpu()
  L = _ffi.lib()
  uid = None
  if rank == 0:
    buf = ctypes.create_string_buffer(128)
    _ffi.check(L.mi355q_comm_unique_id(buf))
    uid = buf.raw
  uid = broadcast(uid)
  comm = ctypes.c_void_p()
  _ffi.check(L.mi355q_comm_init_rank(ctypes.byref(comm), world, uid, rank))
  return comm


def destroy_rccl_comms(group=_ATEXIT) -> None:
  """Destroys the cached communicator of `group` (default: every cached communicator). Call it
  before dist.destroy_process_group(): ncclCommDestroy talks to the peers, and the cache entry
  keeps the group object alive. Raises when RCCL reports a failure."""
  from . import _ffi
  doomed = [e for e in _COMMS if group is _ATEXIT or e[0] is group]
  _COMMS[:] = [e for e in _COMMS if not any(e is d for d in doomed)]
  for _, comm in doomed:
    _ffi.check(_ffi.lib().mi355q_comm_destroy(comm))


def _destroy_rccl_comms_at_exit() -> None:
  """Interpreter exit: the process group or the HIP context may already be gone, and a peer may
  have left -- ncclCommDestroy could fail or wait for it. Communicators are destroyed only while
  the process group still stands; otherwise they are abandoned to process teardown. Never raises."""
  comms, _COMMS[:] = list(_COMMS), []
  try:
    if not comms or not dist.is_available() or not dist.is_initialized():
      return
    from . import _ffi
    L = _ffi.lib()
    for _, comm in comms:
      if L.mi355q_comm_destroy(comm) != 0:
        import sys
        print("mi355q: RCCL communicator not destroyed cleanly at exit: "
              + L.mi355q_last_error().decode("utf-8", "replace"), file=sys.stderr)
  except Exception:  # noqa: BLE001 - exit path
    pass


# ------------------------------------------------- weight requantization ---
def plan_tensor_shards(nbytes: Sequence[int], world_size: int) -> list[int]:
  """Greedy longest-processing-time bin packing of tensor-buffers onto ranks.

  Deterministic (ties -> lower index / lower rank) so every rank derives the
  same plan without communication. Returns the owning rank per tensor.
  """
  order = sorted(range(len(nbytes)), key=lambda i: (-int(nbytes[i]), i))
  load = [0] * world_size
  owner = [0] * len(nbytes)
  for i in order:
    r = min(range(world_size), key=lambda k: (load[k], k))
    owner[i] = r
    load[r] += int(nbytes[i])
  return owner


def quantize_sharded(tensors: dict[str, np.ndarray],
                     quantize_fn: Callable[[str, np.ndarray], Any], group=None) -> Optional[dict]:
  """Requantizes a model's weight buffers with tensor-level sharding.

  Every rank runs `quantize_fn(name, array)` (a get_tensor_quant_params closure)
  on the buffers it owns; rank 0 receives {name: result} for all of them (other
  ranks return None). No collective touches the weights themselves.
  """
  rank, world = _world(group)
  names = sorted(tensors)
  owner = plan_tensor_shards([tensors[n].nbytes for n in names], world)
  mine = {n: quantize_fn(n, tensors[n]) for n, o in zip(names, owner) if o == rank}
  if world == 1:
    return mine
  gathered = [None] * world if rank == 0 else None
  dist.gather_object(mine, gathered, dst=0, group=group)
  if rank != 0:
    return None
  merged: dict[str, Any] = {}
  for part in gathered:
    merged.update(part)
  return merged


# Seconds on one MI355X, fitted to the measured kernels (profiles/r03_c5_model.txt, bench.py
# extras.c5_gptq): what an op costs decides where it runs, not how many bytes it has -- under GPTQ
# a [2048, 16384] weight (128 MiB) costs ~50x a [16384, 2048] one (also 128 MiB) because its
# Hessian is 16384 x 16384.
COST_MODEL = {
    # damped Hessian inverse (mi355q_gptq_hinv_f64): d^3 flop at the rate the blocked factorization
    # sustains + one serial step per 64 columns
    "hinv_flops_per_s": float(os.environ.get("MI355Q_COST_HINV_FLOPS", 97e12)),
    "hinv_step_s": 40e-6,
    # OBS apply (mi355q_gptq_apply_f32): one dependent quantize -> divide -> update step per
    # column + 2 rows d^2 flop of lazy updates (bf16 split path for wide / tall layers)
    "apply_column_s": 0.2e-6,
    "apply_flops_per_s": 123e12,
    "apply_flops_per_s_wide": 222e12,
    # one pass over the FP32 weight (min/max + quantize, HBM bound)
    "stream_bytes_per_s": 4.0e12,
    # per-element cost of the scale searches relative to one pass
    "passes": {"min_max_uniform_quantize": 1.0, "MSE": 2.0, "OCTAV": 14.0, "HADAMARD_ROTATION": 16.0,
               "DECOMPOSED_HADAMARD_ROTATION": 16.0, "OSCAR": 100.0, "GPTQ": 1.0, "float_casting": 1.0,
               "dequantized_weight_recovery": 30.0},
    "launch_s": 20e-6,
    # X2: one ring reduce per distinct Hessian over xGMI (7 links x ~153 GB/s per GPU, point to point: a ring step is bound by
    # ONE link). ASSUMED until a node with more than one GPU has run it: 120 GB/s sustained per link, 30 us per collective.
    "xgmi_link_bytes_per_s": float(os.environ.get("MI355Q_COST_XGMI_LINK", 120e9)),
    "collective_s": 30e-6,
}


def hinv_seconds(d: int) -> float:
  c = COST_MODEL
  return d ** 3 / c["hinv_flops_per_s"] + (d / 64.0) * c["hinv_step_s"]


def gptq_apply_seconds(rows: int, d: int) -> float:
  c = COST_MODEL
  rate = c["apply_flops_per_s_wide"] if (d >= 4096 or rows >= 8192) else c["apply_flops_per_s"]
  return d * c["apply_column_s"] + 2.0 * rows * d * d / rate


def op_cost(item: tuple, model_qsvs: Optional[dict] = None) -> tuple[float, Optional[tuple], float]:
  """(seconds of the op itself, key of what it shares with other ops, seconds of that shared part).

  GPTQ ops that read the same input activation share its Hessian and therefore its damped inverse
  (ref algorithms/uniform_quantize/gptq.py:243-300: `tensor_qsv["activation_tensor_qsv"]["hessian"]`
  is the QSV of op.inputs[0], common_utils.py:182-216): key = ("hessian", activation name), shared
  cost = one inverse. Everything else stands alone."""
  graph_info, op, _, op_key, alg, _ = item
  alg = str(getattr(alg, "value", alg))
  if alg == "no_quantize" or op_key is None:
    return 0.0, None, 0.0
  c = COST_MODEL
  shapes = []
  for tid in op.inputs:
    if tid == -1:
      continue
    t = graph_info.subgraph_tensors[tid]
    data = graph_info.buffers[t.buffer].data
    if data is not None and len(data):
      shapes.append((tuple(int(v) for v in (t.shape if t.shape is not None else ())), len(data)))
  if not shapes:
    return 0.0, None, 0.0
  nbytes = sum(n for _, n in shapes)
  stream = c["launch_s"] + c["passes"].get(alg, 1.0) * nbytes / c["stream_bytes_per_s"]
  if alg == "GPTQ" and str(getattr(op_key, "value", op_key)) == "FULLY_CONNECTED" and len(op.inputs) > 1:
    from .utils import tfl_flatbuffer_utils
    w_shape = next((shp for shp, _ in shapes if len(shp) == 2), None)
    act = graph_info.subgraph_tensors[op.inputs[0]]
    act_name = tfl_flatbuffer_utils.get_tensor_name(act)
    has_h = model_qsvs is None or "hessian" in (model_qsvs.get(act_name) or {})
    if w_shape is not None and has_h and graph_info.buffers[act.buffer].data is None:
      rows, d = w_shape
      return stream + gptq_apply_seconds(rows, d), ("hessian", act_name), hinv_seconds(d)
  return stream, None, 0.0


def plan_op_shards(costs: Sequence[tuple[float, Optional[tuple], float]], world_size: int,
                   links: Optional[Sequence[Sequence]] = None) -> list[int]:
  """Owner rank of every op: ops with the same sharing key travel together (their shared part is
  paid once in the whole job), units are placed longest first on the least loaded rank
  (deterministic: ties -> lower op index / lower rank, so every rank derives the same plan
  without communication). `links[i]`: further keys op i shares with other ops without a cost of their own --
  shared_constant_links: ops that read one constant buffer stay on one rank, where the (buffer, config) cache
  gives both the same result object, as in one process (ref algorithms/utils/common_utils.py:48-77)."""
  parent = list(range(len(costs)))

  def find(i: int) -> int:
    while parent[i] != i:
      parent[i] = parent[parent[i]]
      i = parent[i]
    return i
  first_with: dict[Any, int] = {}
  for i, (_, key, _) in enumerate(costs):
    for k in ([] if key is None else [key]) + list(links[i] if links is not None else ()):
      j = first_with.setdefault(k, i)
      if j != i:
        a, b = find(i), find(j)
        if a != b:
          parent[max(a, b)] = min(a, b)        # (a unit is named by its first op: the tie-break below)
  units: dict[int, list] = {}
  paid: dict[int, set] = {}
  for i, (work, key, shared) in enumerate(costs):
    root = find(i)
    u = units.setdefault(root, [0.0, 0.0, []])
    u[0] += float(work)
    if key is not None and key not in paid.setdefault(root, set()):      # each distinct shared part once per unit
      paid[root].add(key)
      u[1] += float(shared)
    u[2].append(i)
  order = sorted(units.values(), key=lambda u: (-(u[0] + u[1]), u[2][0]))
  load = [0.0] * world_size
  owner = [0] * len(costs)
  for work, shared, members in order:
    r = min(range(world_size), key=lambda k: (load[k], k))
    load[r] += work + shared
    for i in members:
      owner[i] = r
  return owner


def shared_constant_links(plan: Sequence[tuple]) -> list[list]:
  """Per planned op: ("buffer", index) of every constant buffer it reads that another quantized op reads too (tied
  embedding / lm_head, weight-sharing FULLY_CONNECTED ops, tensors that share a buffer)."""
  readers: dict[int, list[int]] = {}
  for i, (graph_info, op, _, op_key, alg, _) in enumerate(plan):
    if str(getattr(alg, "value", alg)) == "no_quantize" or op_key is None:
      continue
    for tid in getattr(op, "inputs", ()):
      if tid == -1:
        continue
      b = graph_info.subgraph_tensors[tid].buffer
      if b and graph_info.buffers[b].data is not None and (not readers.get(b) or readers[b][-1] != i):
        readers.setdefault(b, []).append(i)
  out: list[list] = [[] for _ in plan]
  for b, ops_ in readers.items():
    if len(ops_) > 1:
      for i in ops_:
        out[i].append(("buffer", b))
  return out


def plan_loads(costs: Sequence[tuple[float, Optional[tuple], float]], owner: Sequence[int],
               world_size: int) -> list[float]:
  """Modelled seconds per rank under `owner` (a shared part counts once per rank that needs it)."""
  load = [0.0] * world_size
  paid: set = set()
  for (work, key, shared), r in zip(costs, owner):
    load[r] += work
    if key is not None and (r, key) not in paid:
      paid.add((r, key))
      load[r] += shared
  return load


def x2_reduce_plan(plan: Sequence[tuple], owner: Sequence[int], costs, world_size: int) -> dict:
  """What the Hessian exchange (X2) adds to a sharded GPTQ run, per the cost model: every distinct Hessian is ONE ring
  reduce of its float32 product's packed lower triangle -- d (d + 1) / 2 x 4 bytes, 0.5 GiB at d = 16384 -- to the rank
  that owns its readers (merge_hessians_across_ranks). A ring reduce of S bytes over N ranks moves S (N - 1) / N over
  every link in turn, so it takes every rank the same time whoever receives it: `seconds` is added to EVERY rank's
  modelled load, and `seconds_exposed` is what remains of it on the busiest owner once reduce k + 1 ... run beside the
  damped inverse of Hessian k (reduce_products_beside_compute): the first reduce of an owner, plus whatever exceeds the
  inverses it runs under."""
  c = COST_MODEL
  seen: dict = {}
  for item, o, (_, key, shared) in zip(plan, owner, costs):
    if key is None or key[0] != "hessian" or key in seen:
      continue
    graph_info, op = item[0], item[1]
    d = int(graph_info.subgraph_tensors[op.inputs[1]].shape[-1])
    seen[key] = (d, int(o), float(shared))
  out = {"hessians": len(seen), "bytes": 0, "seconds": 0.0, "bytes_to_owner": [0] * world_size,
         "seconds_exposed": 0.0, "assumed_link_GBps": c["xgmi_link_bytes_per_s"] / 1e9}
  if world_size <= 1 or not seen:
    return out
  inverse_s = [0.0] * world_size
  first_s = [0.0] * world_size
  for (d, o, shared) in seen.values():
    nbytes = d * (d + 1) // 2 * 4
    t = c["collective_s"] + nbytes * (world_size - 1) / world_size / c["xgmi_link_bytes_per_s"]
    out["bytes"] += nbytes
    out["seconds"] += t
    out["bytes_to_owner"][o] += nbytes
    inverse_s[o] += shared
    first_s[o] = max(first_s[o], t)
  # an owner's inverses hide every reduce but (at least) the largest one it waits for first
  out["seconds_exposed"] = max(max(first_s), out["seconds"] - min(v for v in inverse_s if v > 0.0) if any(inverse_s) else out["seconds"])
  out["seconds"] = round(out["seconds"], 5)
  out["seconds_exposed"] = round(out["seconds_exposed"], 5)
  return out


def plan_model_shards(float_model, recipe, world_size: int, calibration_result: Optional[dict] = None):
  """(ParamsGenerator, op plan, owner rank per op, modelled costs) of quantizing `float_model`
  (a path, bytes or a parsed model) on `world_size` ranks -- no arithmetic, same answer on
  every rank."""
  from . import params_generator, quantizer
  qz = float_model if isinstance(float_model, quantizer.Quantizer) else quantizer.Quantizer(float_model, recipe)
  gen = params_generator.ParamsGenerator(qz.float_model)
  plan = gen.plan_ops(qz._recipe_manager)  # pylint: disable=protected-access
  costs = [op_cost(it, calibration_result) for it in plan]
  return qz, gen, plan, plan_op_shards(costs, world_size, shared_constant_links(plan)), costs


def expected_model_bytes(float_model, plan: Sequence[tuple]) -> int:
  """How long the serialized model is expected to be once `plan` has been carried out (0: no idea): the float model's
  length minus what the planned weights lose -- a float32 weight of n elements quantized to b bits is n * b / 8 bytes
  (sub-byte types are packed, ref uniform_quantize_tensor.pack_data) -- plus the scales a channel or a block gets. An
  expectation, not a promise: it sizes the pages allocated ahead of time (LiteRTLMFile.prepare_output), the writer lays
  the file out from the sizes it finds."""
  try:
    length = os.path.getsize(float_model) if isinstance(float_model, (str, os.PathLike)) else len(memoryview(float_model).cast("B"))
  except (TypeError, ValueError, OSError):
    return 0
  seen: set = set()
  saved = 0
  hadamard_sizes: set = set()
  for graph_info, op, _, op_key, alg, cfg in plan:
    w = getattr(cfg, "weight_tensor_config", None) if cfg is not None else None
    if w is None or op_key is None or getattr(op, "inputs", None) is None:
      continue
    for index in list(op.inputs)[1:2]:          # the weight of FULLY_CONNECTED / CONV / EMBEDDING_LOOKUP-like ops
      if index is None or index < 0:
        continue
      tensor = graph_info.subgraph_tensors[index]
      shape = tensor.shape
      if tensor.buffer in seen or shape is None or not len(shape) or tensor.type != 0:      # (0: FLOAT32)
        continue
      data = graph_info.buffers[tensor.buffer].data
      have = len(data) if data is not None else 0
      numel = int(np.prod(shape, dtype=np.int64))
      if have < numel * 4:
        continue                                # (not a constant: an activation feeds this input)
      seen.add(tensor.buffer)
      bits = int(w.num_bits)
      kept = (numel * bits + 7) // 8 if bits in (2, 4) else numel * ((bits + 7) // 8)
      # one float32 scale and one int64 zero point per channel / block / tensor in the tensor's quantization table
      g = str(getattr(w.granularity, "name", w.granularity))
      block = int(g.rsplit("_", 1)[1]) if g.startswith("BLOCKWISE_") and g.rsplit("_", 1)[1].isdigit() else 0
      groups = numel // block if block else (int(shape[0]) if g == "CHANNELWISE" else 1)
      saved += numel * 4 - kept - 12 * groups
      if str(getattr(alg, "value", alg)) == "DECOMPOSED_HADAMARD_ROTATION":
        # the rotation's FULLY_CONNECTED multiplies by H_h / sqrt(h): ONE float32 constant of h x h per distinct size
        # (transformations/graph_edits.py shares it), 16 MiB at h = 2048 -- more than the 3 % below on a small model
        from .algorithms.uniform_quantize import hadamard_rotation
        hadamard_sizes.add(hadamard_rotation.hadamard_size_for(int(shape[-1]), (w.algorithm_params or {}).get("max_hadamard_size")))
  saved -= sum(4 * h * h for h in hadamard_sizes)
  expected = length - saved
  # (+ 3 % and 1 MiB: the tables of a quantized model are longer, Hadamard ops carry their own constants; pages past
  # the length the writer finds out are given back)
  return int(expected + (expected >> 5) + (1 << 20)) if 0 < expected <= length + sum(4 * h * h for h in hadamard_sizes) else 0


def hessian_owners(plan: Sequence[tuple], owner: Sequence[int], costs) -> dict[str, int]:
  """Activation tensor name -> the one rank whose ops read its GPTQ Hessian."""
  out: dict[str, int] = {}
  for (_, key, _), r in zip(costs, owner):
    if key is not None and key[0] == "hessian":
      out[key[1]] = r
  return out


def quantize_model_sharded(float_model, recipe, calibration_result: Optional[dict] = None,
                           serialize_to_path=None, group=None, planned=None, sink=None):
  """`Quantizer(float_model, recipe).quantize(...)` with the ops' weight work spread over the
  ranks of `group` (BASELINE configs 3 and 5: tensor-buffers sharded over 8 GPUs).

  Every rank maps the model file and resolves the recipe (no arithmetic); ops are assigned to
  ranks by modelled cost (op_cost: a GPTQ op costs its OBS update plus -- once per distinct
  Hessian -- the d^3 inverse; other ops cost their passes over the weight bytes) with the ops that
  share a Hessian kept on one rank, so every inverse is computed once in the whole job
  (plan_op_shards: deterministic, no communication). Each rank materializes its ops on its own
  GPU, and rank 0 gathers the per-op results, merges them in op order, applies the
  transformations and serializes. Returns the serialized model on rank 0 and None elsewhere.
  The only exchange is the final gather of the (already quantized, 4-8x smaller) results.
  """
  from . import model_modifier, requant_queue
  rank, world = _world(group)
  # (`planned`: the plan_model_shards() result a caller already derived -- calibrate_and_quantize_sharded
  # plans before calibration, when no rank's statistics can make its plan differ from the others')
  qz, gen, plan, owner, _ = planned if planned is not None else plan_model_shards(float_model, recipe, world, calibration_result)
  rm = qz._recipe_manager  # pylint: disable=protected-access
  if rm.need_calibration() and not calibration_result:
    raise RuntimeError(
        "Model quantization statistics values (QSVs) are required for the input recipe. This"
        " can be obtained by running calibration on sample dataset.")
  qsvs = calibration_result if calibration_result is not None else {}
  if world > 1:
    _require_hessians_where_read([it for it, o in zip(plan, owner) if o == rank], qsvs, rank)
  import contextlib
  from . import runtime as rt
  gen.prefetch([it for it, o in zip(plan, owner) if o == rank], qsvs)
  rt.mark("quantize: prefetched (host)")
  # (one process writing a file: the block stays open over the writer, see Quantizer.quantize)
  writes_here = world == 1 and (serialize_to_path is not None or sink is not None)
  with (requant_queue.batching() if writes_here else contextlib.nullcontext()):
    try:
      with requant_queue.batching():       # this rank's equally shaped weights leave in one launch per group
        mine = {i: gen.materialize_op(it, qsvs) for i, (it, o) in enumerate(zip(plan, owner)) if o == rank}
    finally:
      gen.release_derived(qsvs)
    rt.mark("quantize: ops walked (host)")
    if world == 1:
      params = gen.finish(mine[i] for i in range(len(plan)))
      rt.mark("quantize: params finished (host)")
      try:
        out = model_modifier.ModelModifier(qz.float_model).modify_model(params, serialize_to_path=serialize_to_path, sink=sink)
      finally:
        if writes_here:
          rt.release_upload_files()       # (gen.release_derived left them to the open block)
      rt.mark("quantize: model modified and serialized (host)")
      return out
  # The results are gathered on rank 0, which merges them, applies the transformations and lays the file out. When a file
  # is being written, the quantized payloads themselves stay in their ranks' HBM (runtime.remote_payloads) and every rank
  # writes its own to the offsets rank 0's layout gave them: no gigabyte of pickles through rank 0's host memory.
  remote = (serialize_to_path is not None or sink is not None) and torch.cuda.is_available() and not os.environ.get("MI355Q_GATHER_PAYLOADS")
  if remote:
    with rt.remote_payloads(rank):
      gathered = _gather_results(mine, group)
  else:
    gathered = _gather_results(mine, group)
  out = None
  writes, slots, failure = None, [], None
  try:
    if gathered is not None:
      params = gen.finish(gathered[i] for i in range(len(plan)))
      out = model_modifier.ModelModifier(qz.float_model).modify_model(params, serialize_to_path=serialize_to_path, sink=sink)
  except BaseException as e:  # pylint: disable=broad-exception-caught
    failure = e            # (surfaces below, after the other ranks have been told: none of them may be left waiting)
  if remote:
    if gathered is not None:
      writes, slots = rt.take_remote_writes()
    box = [(writes if failure is None else []), None if failure is None else f"{type(failure).__name__}: {failure}"]
    src = 0 if group is None else dist.get_global_rank(group, 0)
    dist.broadcast_object_list(box, src=src, group=group)
    problem = _write_remote_payloads(box[0] or [], rank, group, slots)
    if failure is None and box[1] is not None:
      failure = RuntimeError(f"the rank that lays the output file out failed: {box[1]}")
    if failure is None and problem is not None:
      failure = RuntimeError(problem)
  if failure is not None:
    raise failure
  return out


def _write_remote_payloads(writes: Sequence[tuple], rank: int, group=None, host_slots: Sequence = ()) -> Optional[str]:
  """This rank's quantized payloads, still in HBM, into the shared output file at the offsets the laying-out rank noted
  for them (runtime.RemoteBuffer.copy_into): pinned staging + pwrite() on this rank's own io ring.

  No rank is left behind: every rank reports how its writes went before anyone returns, and the first problem of any rank
  comes back on ALL ranks (the caller raises it). A payload whose place this rank cannot reach -- the file cannot be opened
  here (another node, a file system that is not shared), or the place is plain memory of the laying-out rank (a caller's
  sink) -- travels to that rank as bytes and is written there: the route every payload took before round 4."""
  from . import runtime as rt
  _, world = _world(group)
  first = 0 if group is None else dist.get_global_rank(group, 0)
  fds: dict[str, Optional[int]] = {}
  by_host: dict[tuple, bytes] = {}        # (key, path, offset) -> the payload's bytes, for the laying-out rank to place
  problem = None
  try:
    for owner, key, path, offset, nbytes in writes:
      if owner != rank:
        continue
      arr = rt._REMOTE_LOCAL.get(key)   # pylint: disable=protected-access
      if arr is None:
        problem = problem or f"rank {rank}: payload {key} is not registered here"
        continue
      t = arr.device_tensor
      if t.numel() * t.element_size() != nbytes:
        problem = problem or f"rank {rank}: payload {key} has {t.numel() * t.element_size()} bytes here, {nbytes} in the layout"
        continue
      fd = None
      if path is not None and not os.environ.get("MI355Q_REMOTE_WRITES_BY_HOST"):
        if path not in fds:
          try:
            fds[path] = os.open(path, os.O_RDWR)
          except OSError:
            fds[path] = None
        fd = fds[path]
      if fd is None:
        by_host[(key, path, offset)] = t.contiguous().reshape(-1).view(torch.uint8).cpu().numpy().tobytes()
      else:
        rt.write_to_file(t, fd, offset)
    if any(fd is not None for fd in fds.values()):
      rt.finish_downloads()
  except Exception as e:  # pylint: disable=broad-exception-caught
    problem = problem or f"rank {rank}: {type(e).__name__}: {e}"
  finally:
    for fd in fds.values():
      if fd is not None:
        os.close(fd)
    rt._REMOTE_LOCAL.clear()   # pylint: disable=protected-access
  # everybody's outcome, and the payloads that go through the laying-out rank's host
  reports = [None] * world
  dist.all_gather_object(reports, (problem, bool(by_host)), group=group)
  if any(r[1] for r in reports):
    parts = [None] * world if rank == 0 else None
    dist.gather_object(by_host, parts, dst=first, group=group)
    if parts is not None:
      try:
        opened: dict[str, int] = {}
        try:
          for part in parts:
            for (key, path, offset), data in part.items():
              if path is None:
                host_slots[offset][:] = np.frombuffer(data, np.uint8)
              else:
                if path not in opened:
                  opened[path] = os.open(path, os.O_RDWR)
                done = 0
                while done < len(data):
                  done += os.pwrite(opened[path], data[done:], offset + done)
        finally:
          for fd in opened.values():
            os.close(fd)
      except Exception as e:  # pylint: disable=broad-exception-caught
        problem = f"rank {rank} (placing payloads sent as bytes): {type(e).__name__}: {e}"
    late = [problem if parts is not None else None]
    dist.broadcast_object_list(late, src=first, group=group)
    if late[0] is not None:
      return late[0]
  return next((r[0] for r in reports if r[0] is not None), None)


def _require_hessians_where_read(items: Sequence[tuple], qsvs: dict, rank: int) -> None:
  """A sharded run keeps each GPTQ Hessian on ONE rank (hessian_owners, from op_cost's view of which
  ops read it). gptq.get_tensor_quant_params, like the reference (ref gptq.py:291-293), quantizes
  with plain min / max when its activation's QSV has no "hessian" -- so an op that reads a Hessian
  this rank gave away would be quantized WITHOUT GPTQ, silently and differently from the
  one-process run. The reader set here is the Calibrator's own (calibrator._plan: every GPTQ op
  reads the Hessian of its first input); a calibrated first input (it has "num_samples") without
  its Hessian on the rank that owns the op is refused."""
  from .utils import tfl_flatbuffer_utils
  for graph_info, op, _, op_key, alg, _ in items:
    if str(getattr(alg, "value", alg)) != "GPTQ" or op_key is None or not len(getattr(op, "inputs", ())):
      continue
    if not any(tid != -1 and graph_info.buffers[graph_info.subgraph_tensors[tid].buffer].data is not None
               for tid in op.inputs[1:]):
      continue                      # no constant operand: nothing GPTQ would update
    name = tfl_flatbuffer_utils.get_tensor_name(graph_info.subgraph_tensors[op.inputs[0]])
    qsv = qsvs.get(name)
    if isinstance(qsv, dict) and "num_samples" in qsv and "hessian" not in qsv:
      raise RuntimeError(
          f"rank {rank} owns a GPTQ op reading the Hessian of '{name}', but that Hessian was reduced to another rank:"
          " the op plan and the Hessian owners disagree (distributed.op_cost vs calibrator._plan).")


def _gather_results(mine: dict, group=None) -> Optional[dict]:
  """Per-op results of every rank -> rank 0 ({op index: results}); None on the other ranks."""
  rank, world = _world(group)
  gathered = [None] * world if rank == 0 else None
  dst = 0 if group is None else dist.get_global_rank(group, 0)
  dist.gather_object(mine, gathered, dst=dst, group=group)
  if rank != 0:
    return None
  merged: dict = {}
  for part in gathered:
    merged.update(part)
  return merged


def calibrate_and_quantize_sharded(float_model, recipe, calibration_data, serialize_to_path=None, group=None,
                                   tensor_provider=None, stats: Optional[dict] = None, sink=None):
  """BASELINE config 5 in one call: `Quantizer.calibrate` + `Quantizer.quantize` over a process group.

  The op plan comes first (a function of model, recipe and world size: the same on every rank), so
  calibration knows which rank will read which GPTQ Hessian: samples are sharded over the ranks
  (calibrate_sharded), every rank multiplies the tokens of its own samples, and each Hessian is
  reduced -- packed lower triangle, one ncclReduce -- to the one rank that owns the ops reading it
  instead of being all-reduced to all. Then every rank materializes its ops (quantize_model_sharded)
  and rank 0 writes the file. Returns the serialized model on rank 0, None elsewhere. `stats`, when
  given, receives wall seconds per phase."""
  import time
  rank, world = _world(group)
  t0 = time.perf_counter()
  from . import runtime as rt
  rt.mark("call")
  planned = plan_model_shards(float_model, recipe, world)
  rt.mark("planned")
  qz, gen, plan, owner, costs = planned
  if rank == 0 and sink is not None and hasattr(sink, "expect") and not os.environ.get("MI355Q_NO_OUTPUT_PREPARE"):
    expected = expected_model_bytes(float_model, plan)
    if expected:
      sink.expect(expected)       # (the output file's pages are allocated underneath the calibration)
  owners = hessian_owners(plan, owner, costs) if world > 1 else None
  qsvs = None
  reserved: list = []          # ops.HinvWorkspace of the large inverses, released when the call is over
  markers: list = []           # events behind the last samples' work (between_samples)
  asked_for_workspace: list = []
  if qz._recipe_manager.need_calibration():  # pylint: disable=protected-access
    mine_items = [it for it, o in zip(plan, owner) if o == rank]
    # calibration reads activations only: the weights this rank will quantize afterwards cross PCIe underneath it
    # (announced here; calibrate_sharded's sample loop starts a quarter GiB of them per sample -- their tensors are
    # fresh HBM, ~30 ms of hipMalloc per GiB on the thread that also has to keep the GPU fed)
    gen.prefetch_weights(mine_items, submit=False)

    def between_samples(walked: int) -> None:
      """Host work that costs hipMalloc time, done while the GPU has samples queued: a quarter GiB of the announced
      weight uploads per sample, and (once) the HBM the first large inverse will take."""
      from .algorithms.uniform_quantize import gptq
      from . import ops
      if torch.cuda.is_available():
        # ... but only while the GPU is behind the walk (the marker of two samples ago has not fired): until the first
        # burst of Hessian products is queued -- 32 samples of 512 tokens fill a slab -- the GPU waits for the walk, and
        # 7.5 ms of hipMalloc per sample there delayed that burst by 0.2 s
        markers.append(torch.cuda.Event())
        markers[-1].record()
        behind = len(markers) > 2 and not markers.pop(0).query()
        rt.pump_prefetch(256 << 20 if behind else 0)      # (uploads into the arena cost this thread nothing: always)
      if walked >= 2 and not reserved and torch.cuda.is_available() and not asked_for_workspace:
        asked_for_workspace.append(True)     # (once: blocks of samples count in steps of K)
        d = gptq.largest_hessian_order(mine_items)
        if d >= 4096:
          reserved.append(ops.HinvWorkspace(d).install())      # (allocated on a helper thread: ops.HinvWorkspace)

    def start_inverses(merged: dict) -> None:
      """The damped inverses this rank's ops will read, started as soon as the Hessians are final: the small ones in
      one lock-step batch, and the first large ones (each is 54 ms of GPU work; an inverse is cached on its Hessian, so
      the applies find it). The rest are inverted where they are read, as before."""
      from .algorithms.uniform_quantize import gptq
      if not torch.cuda.is_available():
        return
      by_name = {name: {"hessian": h} for name, h in merged.items() if h is not None}
      gptq.prefetch_hessian_inverses(mine_items, by_name)
      rt.mark("small inverses started (host)")
      started = 0
      for item in mine_items:
        name = gptq.hessian_name_of(item)
        h = by_name.get(name, {}).get("hessian") if name is not None else None
        if h is not None and hasattr(h, "cache") and h.shape[0] >= 4096 and ("hinv", 0.01) not in h.cache:
          gptq._device_hessian_inverse(h, 0.01)   # pylint: disable=protected-access
          started += 1
          if started == 2:
            break
  try:
    if qz._recipe_manager.need_calibration():  # pylint: disable=protected-access
      qsvs = calibrate_sharded(qz.float_model, recipe, calibration_data, tensor_provider=tensor_provider, group=group,
                               hessian_owners=owners, after_hessians=start_inverses, between_samples=between_samples)
    rt.mark("calibrated (host)", sync=True)
    if torch.cuda.is_available():
      torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = quantize_model_sharded(qz.float_model, recipe, calibration_result=qsvs, serialize_to_path=serialize_to_path,
                                 group=group, planned=planned, sink=sink)
    rt.mark("quantized and written (host)", sync=True)
  finally:
    if torch.cuda.is_available():
      torch.cuda.synchronize()
      rt.release_upload_files()        # (weights announced before a calibration that failed are waited for and dropped)
    for ws in reserved:
      ws.release()
  if stats is not None:
    stats["calibrate_s"] = stats.get("calibrate_s", 0.0) + (t1 - t0)
    stats["quantize_and_write_s"] = stats.get("quantize_and_write_s", 0.0) + (time.perf_counter() - t1)
  return out


# ------------------------------------------------ activation calibration ---
_HESSIAN_ASIDE = "GPTQ:hessian-set-aside"


def _set_hessians_aside(events: list[tuple], running: dict[str, list]) -> list[tuple]:
  """Takes the d x d Hessians out of one sample's statistics: each is merged into this rank's
  running sample-weighted mean (on the device, the same _gptq_merge_hessian chain a single
  process runs, ref utils/qsv_utils.py:71-102) and the record that travels to the other ranks
  keeps only min / max / num_samples and the Hessian's order."""
  out = []
  for name, alg, op_key, qsv in events:
    if not isinstance(qsv, dict) or "hessian" not in qsv:
      out.append((name, alg, op_key, qsv))
      continue
    small = {k: v for k, v in qsv.items() if k != "hessian"}
    h, n = qsv["hessian"], qsv["num_samples"]
    small["hessian_dim"] = int(h.shape[0])
    cur = running.get(name)
    if cur is None:
      running[name] = [h, n]
    else:
      cur[0], cur[1] = qsv_utils._gptq_merge_hessian(   # pylint: disable=protected-access
          {"hessian": cur[0], "num_samples": cur[1]}, {"hessian": h, "num_samples": n})
    out.append((name, _HESSIAN_ASIDE, op_key, small))
  return out


def _ema_and_count_update(qsv, new_qsv):
  """The min / max / num_samples part of gptq_and_moving_average_update (ref :90-102)."""
  out = qsv_utils.moving_average_update(qsv, new_qsv)
  out["num_samples"] = qsv["num_samples"] + new_qsv["num_samples"]
  for key in ("hessian", "hessian_dim"):
    if key in qsv:
      out[key] = qsv[key]
  return out


_ema_and_count_update.block_mode = "count"      # (see utils/qsv_utils.py: Calibrator.replay advances it over whole blocks)


def calibrate_sharded(float_model, recipe, calibration_data, previous_calibration_result=None,
                      tensor_provider=None, group=None, hessians: str = "consumed",
                      hessian_owners: Optional[dict[str, int]] = None,
                      after_hessians: Optional[Callable[[dict], None]] = None,
                      between_samples: Optional[Callable[[int], None]] = None,
                      samples_per_launch: Optional[int] = None) -> dict:
  """`Quantizer(float_model, recipe).calibrate(calibration_data)` with every signature's samples
  sharded contiguously over the ranks of `group` (BASELINE config 4: 512 samples over 8 GPUs;
  config 5: GPTQ Hessians).

  Each rank walks its samples on its GPU but keeps their per-tensor statistics as events
  (Calibrator.record_step); the events -- a few floats per tensor and sample -- are all-gathered
  and every rank replays all samples in dataset order through the ops' own update rules, so
  min / max (the order-dependent moving average) and OSCAR's mu2 equal the single-process result
  bit for bit. GPTQ Hessians (d x d float64 per activation) never enter the gather: every rank
  keeps the running mean over its own samples in HBM and the ranks combine them with one
  all-reduce(sum) per distinct Hessian (merge_hessians_across_ranks, X2), exact up to FP64
  rounding. Returns the model QSVs on every rank. With `hessian_owners` (tensor name -> rank, see
  calibrate_and_quantize_sharded) a Hessian is reduced to that one rank only and the other ranks'
  QSVs carry no "hessian" entry for it.
  """
  from . import calibrator, quantizer
  rank, world = _world(group)
  qz = quantizer.Quantizer(float_model, recipe)
  rm = qz._recipe_manager  # pylint: disable=protected-access
  if not rm.need_calibration():
    return {}
  from . import runtime as rt
  local = calibrator.Calibrator(qz.float_model, tensor_provider=tensor_provider, hessians=hessians)
  mine = []                                    # (signature index, sample index, events | StepBlock of K samples from there)
  running: dict[str, list] = {}                # tensor name -> [Hessian mean over my samples, count]

  def hessian_sink(name, xs, ns) -> None:
    """The samples of one block for one Hessian: straight into this rank's running statistic (what
    _set_hessians_aside does with the per-sample accumulators of the per-sample walk)."""
    from .algorithms.uniform_quantize import gptq
    cur = running.get(name)
    if cur is None:
      cur = running[name] = [gptq.HessianAccumulator(int(xs[0].shape[1])), 0]
    if hasattr(cur[0], "add_block"):
      cur[0].add_block(xs, ns)
    else:          # (a float64 array from the per-sample walk of non-float32 samples: its own merge rule)
      for x, n in zip(xs, ns):
        cur[0], _ = qsv_utils._gptq_merge_hessian(   # pylint: disable=protected-access
            {"hessian": cur[0], "num_samples": cur[1]}, {"hessian": gptq.HessianAccumulator.of(x, n), "num_samples": n})
        cur[1] = cur[1] + n
      return
    cur[1] = cur[1] + sum(ns)

  with local.plan_once():
    for sig_idx, (signature_key, dataset) in enumerate(calibration_data.items()):
      samples = dataset if hasattr(dataset, "__len__") and hasattr(dataset, "__getitem__") else list(dataset)
      shard = sample_shard(len(samples), rank, world)
      limit = local.samples_per_launch(signature_key, samples, rm, samples_per_launch)
      if limit > 1:
        # K samples per launch (calibrator.StepBlock): their statistics stay arrays until they are replayed
        walked = 0
        for k, item in local.record_blocks(signature_key, (samples[j] for j in shard), rm, limit, first=shard.start,
                                           hessian_sink=hessian_sink, hessian_tag=_HESSIAN_ASIDE):
          if not isinstance(item, calibrator.StepBlock):
            item = _set_hessians_aside(item, running)
          mine.append((sig_idx, k, item))
          walked += len(item) if isinstance(item, calibrator.StepBlock) else 1
          if between_samples is not None:
            between_samples(walked)
        continue
      steps = local.record_steps(signature_key, (samples[j] for j in shard), rm)
      for k, events in zip(shard, steps):
        mine.append((sig_idx, k, _set_hessians_aside(events, running)))
        if between_samples is not None:      # (the caller's host work that is better done while the GPU has samples queued)
          between_samples(len(mine))
      steps.close()        # (the last step's window: zip stops without resuming the generator)
  rt.mark("samples walked (host)")
  local.wait_for_statistics()
  rt.mark("statistics on the host")       # (record_steps hands the samples' min / max over while their copies are in flight)
  if world > 1:
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    mine = [step for part in parts for step in part]
  mine.sort(key=lambda step: (step[0], step[1]))
  # The Hessians first: their exchange and whatever the caller starts on them (`after_hessians`: the damped inverses,
  # calibrate_and_quantize_sharded) is GPU work that then runs underneath the host-only replay of the min / max
  # statistics below (25 000 small NumPy updates for an 18-layer model: 0.1 s during which the GPU used to idle).
  totals: dict[str, list] = {}
  for _, _, events in mine:
    if isinstance(events, calibrator.StepBlock):
      for t, d in events.hessian_dims.items():
        entry = totals.setdefault(events.slots[t][0], [d, 0])
        entry[1] += events.num_samples[:, t].sum()
      continue
    for name, alg, _, qsv in events:
      if alg == _HESSIAN_ASIDE:
        entry = totals.setdefault(name, [qsv["hessian_dim"], 0])
        entry[1] += qsv["num_samples"]
  merged = merge_hessians_across_ranks({n: (h, c) for n, (h, c) in running.items()},
                                       {n: (d, c) for n, (d, c) in totals.items()}, group, hessian_owners)
  final = calibrator.Calibrator(qz.float_model, tensor_provider=tensor_provider)
  if previous_calibration_result is not None:
    final.load_model_qsvs(previous_calibration_result)
  earlier = {name: (qsv["hessian"], qsv["num_samples"]) for name, qsv in final.get_model_qsvs().items()
             if isinstance(qsv, dict) and "hessian" in qsv}
  rt.mark("hessians merged (host)")
  if after_hessians is not None and not earlier:
    after_hessians(merged)
  rt.mark("inverses started (host)")
  final.replay((events for _, _, events in mine), update_overrides={_HESSIAN_ASIDE: _ema_and_count_update})
  rt.mark("replayed (host)")
  qsvs = final.get_model_qsvs()
  for name in totals:
    if name not in merged:          # reduced to another rank: this one never reads it
      qsvs[name].pop("hessian_dim", None)
      qsvs[name].pop("hessian", None)
  for name, h in merged.items():
    qsv = qsvs[name]
    qsv.pop("hessian_dim", None)
    if name in earlier:       # resumed calibration: the earlier result weighs in with its own count
      h, _ = qsv_utils._gptq_merge_hessian(   # pylint: disable=protected-access
          {"hessian": earlier[name][0], "num_samples": earlier[name][1]},
          {"hessian": h, "num_samples": totals[name][1]})
    qsv["hessian"] = h
  return qsvs


def sample_shard(num_samples: int, rank: int, world_size: int) -> range:
  """Contiguous, near-equal sample ranges in dataset order."""
  base, extra = divmod(num_samples, world_size)
  start = rank * base + min(rank, extra)
  return range(start, start + base + (1 if rank < extra else 0))


def local_activation_stats(samples: Sequence[dict[str, np.ndarray]], names: Sequence[str],
                           valid_range=(-3e38, 3e38)) -> np.ndarray:
  """Per-sample (min, max) of every named activation on this rank's GPU.

  One K7 launch per sample batch (mi355q_act_minmax_f32); returns
  float32 [n_local_samples, n_tensors, 2]. ref: common_quantize.py:1362-1413.
  """
  from . import ops
  from . import runtime as rt
  rt.require_gpu()
  out = np.empty((len(samples), len(names), 2), np.float32)
  for i, sample in enumerate(samples):
    dev = [rt.to_device(np.ascontiguousarray(sample[n], dtype=np.float32).reshape(-1))
           for n in names]
    out[i] = rt.to_numpy(ops.act_minmax(dev, valid_range[0], valid_range[1]))
  return out


def gather_sample_stats(local_stats: np.ndarray, group=None) -> np.ndarray:
  """All-gather of per-sample statistics -> [n_total_samples, ...] in dataset order (X1).

  Shards may have different lengths (sample_shard); ranks exchange lengths first
  and pad to the longest so that a single all-gather moves the payload.
  """
  rank, world = _world(group)
  local_stats = np.ascontiguousarray(local_stats, dtype=np.float32)
  if world == 1:
    return local_stats
  tail = local_stats.shape[1:]
  comm = rccl_comm(group)
  if comm is not None:
    from . import _ffi
    from . import runtime as rt
    L, dev = _ffi.lib(), rt.device()
    n_local = torch.tensor([float(local_stats.shape[0])], dtype=torch.float32, device=dev)
    counts_t = torch.empty((world,), dtype=torch.float32, device=dev)
    _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(n_local), 1, rt.ptr(counts_t), rt.stream_ptr()))
    counts = [int(c) for c in counts_t.cpu().tolist()]
    longest = max(counts)
    padded = np.zeros((longest,) + tail, np.float32)
    padded[: local_stats.shape[0]] = local_stats
    mine = torch.from_numpy(padded).to(dev)
    parts = torch.empty((world,) + tuple(padded.shape), dtype=torch.float32, device=dev)
    _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(mine), mine.numel(), rt.ptr(parts), rt.stream_ptr()))
    host = parts.cpu().numpy()
    return np.concatenate([host[r, :c] for r, c in enumerate(counts)], axis=0)
  dev = _comm_device(group)
  n_local = torch.tensor([local_stats.shape[0]], dtype=torch.int64, device=dev)
  counts = [torch.zeros_like(n_local) for _ in range(world)]
  dist.all_gather(counts, n_local, group=group)
  counts = [int(c.item()) for c in counts]
  longest = max(counts)
  padded = np.zeros((longest,) + tail, np.float32)
  padded[: local_stats.shape[0]] = local_stats
  mine = torch.from_numpy(padded).to(dev)
  parts = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(parts, mine, group=group)
  return np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], axis=0)


def replay_qsv_updates(stats: np.ndarray, names: Sequence[str], shapes: Sequence[tuple],
                       num_samples: Sequence[int] | None = None,
                       update_fn: Callable = qsv_utils.moving_average_update) -> dict[str, dict]:
  """Applies the QSV update rule sample by sample, exactly as one process would.

  stats: [n_samples, n_tensors, 2]; shapes[t] is the activation's shape (the QSV
  min/max have shape (1,)*ndim, ref: common_quantize.py:1380).
  """
  qsvs: dict[str, dict] = {}
  for s in range(stats.shape[0]):
    for t, name in enumerate(names):
      shp = (1,) * len(shapes[t])
      new = {"min": np.reshape(stats[s, t, 0], shp), "max": np.reshape(stats[s, t, 1], shp)}
      if num_samples is not None:
        new["num_samples"] = np.array(num_samples[t])
      qsvs[name] = update_fn(qsvs.get(name), new)
  return qsvs


def allreduce_min_max(local_stats: np.ndarray, group=None) -> np.ndarray:
  """Global [n_tensors, 2] (min, max) when the update rule is `min_max_update`.

  Associative and commutative, so one all-reduce(MIN) + one all-reduce(MAX) over
  xGMI (one RCCL group) replaces the gather + replay.
  """
  rank, world = _world(group)
  if local_stats.shape[0]:
    mn = local_stats[..., 0].min(axis=0)
    mx = local_stats[..., 1].max(axis=0)
  else:
    mn = np.full(local_stats.shape[1], np.inf, np.float32)
    mx = np.full(local_stats.shape[1], -np.inf, np.float32)
  if world > 1:
    comm = rccl_comm(group)
    dev = _comm_device(group)
    tmn = torch.from_numpy(np.ascontiguousarray(mn, dtype=np.float32)).to(dev)
    tmx = torch.from_numpy(np.ascontiguousarray(mx, dtype=np.float32)).to(dev)
    if comm is not None:
      from . import _ffi
      from . import runtime as rt
      _ffi.check(_ffi.lib().mi355q_allreduce_minmax_f32(comm, rt.ptr(tmn), rt.ptr(tmx), tmn.numel(),
                                                         rt.stream_ptr()))
    else:
      dist.all_reduce(tmn, op=dist.ReduceOp.MIN, group=group)
      dist.all_reduce(tmx, op=dist.ReduceOp.MAX, group=group)
    mn, mx = tmn.cpu().numpy(), tmx.cpu().numpy()
  return np.stack([mn, mx], axis=-1)


# ---------------------------------------------- GPTQ Hessian / OSCAR mu2 ---
def _sum_f64_across_ranks(t: torch.Tensor, group=None, comm=None) -> torch.Tensor:
  """In-place all-reduce(sum) of a float64 tensor that this function's caller owns."""
  if comm is not None:
    from . import _ffi
    from . import runtime as rt
    _ffi.check(_ffi.lib().mi355q_allreduce_sum_f64(comm, rt.ptr(t), t.numel(), rt.stream_ptr()))
  else:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
  return t


def allreduce_hessian(weighted_sum, num_samples: int, group=None):
  """Merges per-rank Hessian statistics into the global sample-weighted mean.

  weighted_sum = sum_i n_i * H_i over this rank's samples (H_i = (2/n_i) X_i^T X_i,
  ref: gptq.py:100-107), as a NumPy array or a torch tensor (HBM resident for
  nccl); num_samples = sum_i n_i. Returns (H, total_samples) with
  H = sum_all / total, which is what chaining _gptq_merge_hessian
  (ref: utils/qsv_utils.py:71-88) over all samples yields up to FP64 rounding.
  OSCAR's per-channel second moment merges by the same sample-weighted mean
  (_oscar_merge_mu2, ref: utils/qsv_utils.py:125-158): pass sum_i n_i * mu2_i and sum_i n_i.
  The caller's array is never written (the collective runs on a copy).
  """
  rank, world = _world(group)
  if hasattr(weighted_sum, "device_tensor"):      # runtime.HbmArray: reduce it where it lives
    weighted_sum = weighted_sum.device_tensor
  is_np = isinstance(weighted_sum, np.ndarray)
  t = torch.from_numpy(np.array(weighted_sum, dtype=np.float64)) if is_np else weighted_sum.double().clone()
  n = torch.tensor([float(num_samples)], dtype=torch.float64)
  if world > 1:
    comm = rccl_comm(group)
    dev = _comm_device(group)
    t, n = t.to(dev), n.to(dev)
    _sum_f64_across_ranks(t, group, comm)
    _sum_f64_across_ranks(n, group, comm)
  total = int(round(float(n.item())))
  h = t / total if total else t
  if is_np:          # the caller's dtype comes back (a float32 OSCAR mu2 stays float32, as the
    return h.cpu().numpy().astype(np.asarray(weighted_sum).dtype, copy=False), total   # one-process merge)
  return h.to(weighted_sum.dtype), total


allreduce_second_moment = allreduce_hessian   # OSCAR mu2: same sample-weighted mean


def _product_form_everywhere(names: Sequence[str], local: dict, group, comm) -> dict[str, bool]:
  """name -> every rank's statistic of it is a float32 product (or the rank saw no sample of it): what the product-form
  exchange needs. One small all-reduce(min) for all names. MI355Q_X2_F64=1 keeps the float64 exchange."""
  if os.environ.get("MI355Q_X2_F64"):
    return {}
  flags = []
  for name in names:
    h = local.get(name, (None, 0.0))[0]
    flags.append(1.0 if h is None or (hasattr(h, "product_form") and h.product_form() is not None) else 0.0)
  if not flags:
    return {}
  if comm is not None:
    from . import _ffi
    from . import runtime as rt
    lo = torch.tensor(flags, dtype=torch.float32, device=rt.device())
    hi = lo.clone()
    _ffi.check(_ffi.lib().mi355q_allreduce_minmax_f32(comm, rt.ptr(lo), rt.ptr(hi), lo.numel(), rt.stream_ptr()))
    agreed = lo.cpu().tolist()
  else:
    t = torch.tensor(flags, dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    agreed = t.tolist()
  return {name: v >= 1.0 for name, v in zip(names, agreed)}


_COMM_STREAM: list = []
ISSUED: list = []       # (tensor name, root) of every product reduce, in the order it was issued (tests read this)


def comm_stream():
  """The HIP stream the X2 reduces run on, beside the compute stream (one per process)."""
  if not _COMM_STREAM:
    _COMM_STREAM.append(torch.cuda.Stream())
  return _COMM_STREAM[0]


def reduce_products_beside_compute(comm, jobs: Sequence[tuple]) -> dict[str, Any]:
  """The float32-product reduces of `jobs` = [(name, product [d, d] float32 in HBM, d, root)], issued back to back on
  the communication stream in the order given -- the SAME order on every rank (RCCL matches collectives by order of
  issue) -- behind everything the compute stream has been given so far. Returns {name: event that fires when that
  Hessian's sum has been unpacked into `product` on the ranks that receive it}. Nothing waits here: the host goes on
  to start the damped inverses, each of which waits for ITS Hessian's event only (HessianAccumulator.ready), so the
  inverse of Hessian k runs while Hessians k + 1 ... are still crossing xGMI. The 8-rank C5 plan has 18 reduces of
  0.5 GiB (d = 16384) per job -- tens of milliseconds the owners used to spend waiting before their first inverse.
  Ref (what this exchange replaces): utils/qsv_utils.py:71-102, calibrator.py:395-421."""
  from . import _ffi
  from . import runtime as rt
  if not jobs:
    return {}
  L = _ffi.lib()
  side = comm_stream()
  side.wait_stream(torch.cuda.current_stream())          # the products' last slabs are multiplied on the compute stream
  need = max(int(L.mi355q_product_exchange_workspace_bytes(d)) for _, _, d, _ in jobs)
  scratch = rt.empty((max(need, 1),), torch.uint8)
  scratch.record_stream(side)                            # (the allocator must not hand it out again before `side` is through)
  events = {}
  for name, prod, d, root in jobs:
    prod.record_stream(side)
    _ffi.check(L.mi355q_reduce_product_f32(comm, rt.ptr(prod), d, int(root), rt.ptr(scratch), scratch.numel(),
                                           ctypes.c_void_p(side.cuda_stream)))
    ISSUED.append((name, int(root)))
    ev = torch.cuda.Event()
    ev.record(side)
    events[name] = ev
  return events


def merge_hessians_across_ranks(local: dict[str, tuple[Any, float]], totals: dict[str, tuple[int, float]],
                                group=None, owners: Optional[dict[str, int]] = None) -> dict[str, Any]:
  """X2: the sample-weighted mean of every GPTQ Hessian over all ranks.

  local[name] = (H_rank, n_rank): this rank's running mean over its own samples (the chain of
  _gptq_merge_hessian, ref utils/qsv_utils.py:71-102) and their sample count; names this rank
  saw no sample of are simply absent. totals[name] = (d, N): Hessian order and the sample count
  over all ranks (known to every rank from the gathered per-sample records, so no collective is
  spent on counts). Per distinct Hessian, in sorted-name order on every rank: the packed lower
  triangle of H_rank * n_rank / N (the matrix is symmetric: d (d + 1) / 2 float64, 1 GiB at
  d = 16384 instead of 2) goes through one collective -- mi355q_reduce_hessian_f64 over RCCL when
  ranks own GPUs; nothing d x d is ever pickled. Round 4: when every rank holds the statistic as the
  float32 product it accumulated (gptq.HessianAccumulator: the normal case on GPUs) the PRODUCTS are
  summed instead -- mi355q_reduce_product_f32: packed float32 triangle, 0.5 GiB at d = 16384, no
  float64 d x d array on any rank -- and the receiving ranks keep product form (the damped inverse
  reads it as it is). Equal to the float64 exchange within float32 summation (1e-7 relative). `owners` (tensor name -> rank, from
  hessian_owners(): the one rank whose ops read that Hessian) turns the all-reduce into a reduce to
  that rank, half the ring traffic again; without it every rank ends with every mean.
  Returns {name: H} (runtime.HbmArray when the data lives in HBM) for the Hessians this rank holds.
  """
  rank, world = _world(group)
  comm = rccl_comm(group) if world > 1 else None
  out: dict[str, Any] = {}
  on_gpu = torch.cuda.is_available()
  for h, _ in local.values():
    if hasattr(h, "finalize"):       # tokens still waiting in a slab are multiplied now
      h.finalize()
  if on_gpu:
    from . import _ffi, ops
    from . import runtime as rt
    ops.release_scratch()
  scratch = None
  names = sorted(totals)
  if comm is not None and _COMM_STREAM:
    # collectives of one communicator are kept in ONE order on the device as well: whatever an earlier call still has on
    # the communication stream is in front of everything this call issues on the compute stream (the agreement below,
    # the float64 form further down)
    torch.cuda.current_stream().wait_stream(_COMM_STREAM[0])
  in_product_form = _product_form_everywhere(names, local, group, comm) if (world > 1 and on_gpu) else {}
  # the statistics that travel in the form they are kept in -- the float32 product's packed lower triangle, summed in
  # float32 (0.5 GiB per d = 16384 Hessian); the receiving ranks keep them as products (alpha = 2 / N). Over RCCL all of
  # them are issued first, on the communication stream (reduce_products_beside_compute): same order on every rank.
  jobs, kept = [], {}
  if world > 1:
    for name in names:
      if not in_product_form.get(name):
        continue
      d, total = totals[name]
      h, _ = local.get(name, (None, 0.0))
      root = -1 if owners is None else int(owners.get(name, -1))
      prod = None if h is None else h.product_form()[0]
      if prod is None:             # this rank saw no sample of it: it still takes part (and may be the one that keeps the sum)
        prod = torch.zeros((d, d), dtype=torch.float32, device=rt.device())
      jobs.append((name, prod, d, root))
      kept[name] = prod
  ready = reduce_products_beside_compute(comm, jobs) if comm is not None else {}
  for name in names:
    d, total = totals[name]
    h, n_rank = local.get(name, (None, 0.0))
    root = -1 if owners is None else int(owners.get(name, -1))
    mine = root < 0 or root == rank
    if world == 1:
      out[name] = h
      continue
    if in_product_form.get(name):
      from .algorithms.uniform_quantize import gptq
      prod = kept[name]
      if comm is None:                                          # test transport: ranks share a GPU
        ISSUED.append((name, root))
        host = torch.tril(prod).cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        if mine:
          prod.copy_(host)
      if mine:
        acc = gptq.HessianAccumulator(d)
        acc._prod, acc._n_prod = prod, float(total)   # pylint: disable=protected-access
        acc.ready = ready.get(name)                   # (whoever reads the product first waits for its sum: flush())
        out[name] = acc
      continue
    weight = float(n_rank) / float(total) if total else 0.0
    if on_gpu:
      if h is None:
        t = torch.zeros((d, d), dtype=torch.float64, device=rt.device())
      else:                        # the rank that keeps the result works on a copy: the QSV's own Hessian stays as it is
        t = rt.on_device(h, torch.float64)
        t = t.clone() if mine else t
      if comm is not None:
        if ready:        # (this call's product reduces first: same communicator, one order on the device)
          torch.cuda.current_stream().wait_stream(comm_stream())
        L = _ffi.lib()
        need = L.mi355q_hessian_exchange_workspace_bytes(d)
        if scratch is None or scratch.numel() < need:
          scratch = rt.empty((need,), torch.uint8)
        _ffi.check(L.mi355q_reduce_hessian_f64(comm, rt.ptr(t), d, weight, root, rt.ptr(scratch), scratch.numel(),
                                               rt.stream_ptr()))
      else:                                                     # test transport: ranks share a GPU
        host = (t * weight).cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t = host.to(rt.device())
      if mine:
        out[name] = rt.HbmArray(t)
    else:
      t = torch.zeros((d, d), dtype=torch.float64) if h is None else torch.from_numpy(np.array(h, dtype=np.float64))
      t *= weight
      dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
      if mine:
        out[name] = t.numpy()
  return out
