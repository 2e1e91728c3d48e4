#!/usr/bin/env python3
"""BASELINE config 5 as ONE call chain: a Gemma-2B-shaped `.litertlm` -> calibrate -> GPTQ / Hadamard
int4 -> quantized `.litertlm`, on one GPU or sharded over the ranks of a torchrun job.

  python tools/c5_model.py [--layers 18] [--sequences 128] [--tokens 512] [--variant gptq|mixed|hadamard]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/c5_model.py ...

The model: per decoder layer q, o [2048, 2048]; k, v [256, 2048]; gate, up [16384, 2048];
down [2048, 16384]; W ~ N(0, 0.02^2), seed 5000 + layer. q / k / v read one activation, gate / up
another (what makes them share a Hessian). The calibration set is `sequences` samples of
[1, tokens, d] per FULLY_CONNECTED input, generated in HBM (seed 5000 + layer) -- the activations a
float run of the model on this GPU would have left there; the reference obtains them from the LiteRT
interpreter (calibrator.py:518-533), which is a third-party runtime. The ops' outputs are calibrated
too (min / max), from one shared buffer per tensor.

Variants (the reference has no combined "GPTQ + Hadamard" key: a recipe gives one algorithm per
scope):
  gptq      GPTQ int4 channelwise on all seven projections (the d = 16384 Hessian included)
  mixed     GPTQ on q / k / v / o / gate / up, decomposed Hadamard rotation + OCTAV int4 on down
  hadamard  decomposed Hadamard rotation + OCTAV int4 everywhere

One JSON line: seconds per phase (open, calibrate, quantize = plan + H^-1 + apply + gather, write),
GPU-busy seconds per kernel family from HIP events around every ops call, and the modelled plan.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
  if _p not in sys.path:
    sys.path.insert(0, _p)

D, DKV, DFF = 2048, 256, 16384
TEMPLATE = os.path.join(ROOT, "tests", "golden", "models", "conv_fc_mnist.litertlm")


def projections(d=D, dkv=DKV, dff=DFF):
  """(name, rows, cols, input activation) of one decoder layer's FULLY_CONNECTED ops."""
  return (("q", d, d, "attn_in"), ("k", dkv, d, "attn_in"), ("v", dkv, d, "attn_in"), ("o", d, d, "o_in"),
          ("gate", dff, d, "mlp_in"), ("up", dff, d, "mlp_in"), ("down", d, dff, "down_in"))


def build_model(layers, d=D, dkv=DKV, dff=DFF, weights=None):
  """ModelT tree of `layers` decoder layers. weights(layer, name, rows, cols) -> float32 [rows, cols]
  (default: default_rng(5000 + layer) normal * 0.02, drawn in projection order)."""
  from mi355q import qtyping as q
  model = q.ModelT(version=3, description=b"gemma-2b shaped decoder layers (synthetic, BASELINE config 5)")
  model.buffers = [q.BufferT()]
  sg = q.SubGraphT(name=b"main", tensors=[], operators=[], inputs=[], outputs=[])

  def act(name, width):
    sg.tensors.append(q.TensorT(name=name.encode(), shape=[1, width], buffer=0))
    return len(sg.tensors) - 1
  for layer in range(layers):
    rng = np.random.default_rng(5000 + layer)
    p = f"l{layer}"
    ins = {}
    for name, rows, cols, src in projections(d, dkv, dff):
      if src not in ins:
        ins[src] = act(f"{p}/{src}", cols)
        sg.inputs.append(ins[src])
      if weights == "virtual":     # shapes and byte counts only (planning): zero-stride, no memory
        model.buffers.append(q.BufferT(data=np.broadcast_to(np.zeros(1, np.uint8), (rows * cols * 4,))))
      else:
        w = (weights(layer, name, rows, cols) if weights is not None
             else rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(0.02))
        model.buffers.append(q.BufferT(data=np.ascontiguousarray(w, dtype=np.float32).reshape(-1).view(np.uint8)))
      sg.tensors.append(q.TensorT(name=f"{p}/{name}/w".encode(), shape=[rows, cols], buffer=len(model.buffers) - 1))
      wid = len(sg.tensors) - 1
      y = act(f"{p}/{name}/y", rows)
      sg.operators.append(q.OperatorT(inputs=[ins[src], wid, -1], outputs=[y], opcodeIndex=0, builtinOptionsType=8,
                                      builtinOptions=q.FullyConnectedOptionsT()))
      sg.outputs.append(y)
  model.operatorCodes = [q.OperatorCodeT(builtinCode=int(q.BuiltinOperator.FULLY_CONNECTED), deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0)]
  return model


def device_weights(torch):
  """weights() callback that draws on the GPU (an 18-layer model is 7.9 GB of normals)."""
  gens = {}

  def draw(layer, name, rows, cols):
    g = gens.get(layer)
    if g is None:
      g = gens[layer] = torch.Generator(device="cuda").manual_seed(5000 + layer)
    return (torch.randn((rows, cols), generator=g, device="cuda") * 0.02).cpu().numpy()
  return draw


def write_litertlm(model, path):
  """The model as the one TFLite section of a LiteRT-LM container (header of the reference's
  fixture, section replaced)."""
  from mi355q import model_modifier
  from mi355q.utils import litertlm_utils, tfl_flatbuffer_utils
  tfl = path + ".section.tflite"
  model_modifier.serialize_model(model, tfl)
  data = tfl_flatbuffer_utils.get_model_content(tfl)
  n = litertlm_utils.LiteRTLMFile(TEMPLATE).serialize(path, {0: data})
  del data
  os.remove(tfl)
  return n


def _fc(algorithm_key, regex=".*", bits=4, **params):
  w = dict(num_bits=bits, symmetric=True, granularity="CHANNELWISE", dtype="INT")
  if params:
    w["algorithm_params"] = params
  return dict(regex=regex, operation="FULLY_CONNECTED", algorithm_key=algorithm_key, op_config=dict(
      weight_tensor_config=w, compute_precision="INTEGER", explicit_dequantize=False, skip_checks=False,
      min_weight_elements=0))


def recipe(variant, bits=4, max_hadamard_size=2048):
  if variant == "gptq":
    return [_fc("GPTQ", bits=bits)]
  if variant == "hadamard":
    return [_fc("DECOMPOSED_HADAMARD_ROTATION", bits=bits, max_hadamard_size=max_hadamard_size)]
  if variant == "mixed":          # later entries win where both match (recipe_manager)
    return [_fc("GPTQ", bits=bits),
            _fc("DECOMPOSED_HADAMARD_ROTATION", regex=r".*/down/.*", bits=bits, max_hadamard_size=max_hadamard_size)]
  raise ValueError(variant)


def calibration_set(torch, layers, sequences, tokens, d=D, dkv=DKV, dff=DFF, batch=1, out_tokens=None):
  """`sequences / batch` samples of {tensor name: device tensor}. Inputs [batch, tokens, d] ~ N(0, 1),
  distinct per sample; outputs one shared [batch, out_tokens, rows] buffer per tensor."""
  out_tokens = tokens if out_tokens is None else out_tokens
  n = sequences // batch
  samples = [dict() for _ in range(n)]
  for layer in range(layers):
    g = torch.Generator(device="cuda").manual_seed(5000 + layer)
    p = f"l{layer}"
    seen = set()
    for name, rows, cols, src in projections(d, dkv, dff):
      if src not in seen:
        seen.add(src)
        x = torch.randn((n, batch, tokens, cols), generator=g, device="cuda")
        for k in range(n):
          samples[k][f"{p}/{src}"] = x[k]
      y = torch.randn((batch, out_tokens, rows), generator=g, device="cuda")
      for k in range(n):
        samples[k][f"{p}/{name}/y"] = y
  return samples


class GpuPhases:
  """GPU-busy milliseconds per kernel family: HIP events around every mi355q.ops call of a family
  (no synchronisation is added; the events are read after the run)."""
  FAMILIES = {"gptq_xtx": "hessian", "gptq_xtx_accum": "hessian", "gptq_xtx_finish": "hessian_finish",
              "gptq_hessian_merge": "hessian_merge", "gptq_hinv": "hinv", "gptq_hinv_batched": "hinv",
              "gptq_hinv_from_product": "hinv",
              "gptq_apply": "apply", "act_minmax": "act_minmax", "act_minmax_entries": "act_minmax", "requant_sym": "scales",
              "hadamard_rotate": "hadamard", "octav_clip": "octav", "pack_bits": "pack", "minmax": "scales"}

  def __init__(self, torch, ops):
    self.torch, self.ops, self.events, self.saved = torch, ops, [], {}

  def __enter__(self):
    for fn, family in self.FAMILIES.items():
      orig = getattr(self.ops, fn)
      self.saved[fn] = orig

      def wrapped(*a, _orig=orig, _family=family, **kw):
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _orig(*a, **kw)
        e1.record()
        self.events.append((_family, e0, e1))
        return out
      setattr(self.ops, fn, wrapped)
    return self

  def __exit__(self, *exc):
    for fn, orig in self.saved.items():
      setattr(self.ops, fn, orig)

  def totals(self):
    self.torch.cuda.synchronize()
    out = {}
    for family, e0, e1 in self.events:
      out[family] = out.get(family, 0.0) + e0.elapsed_time(e1)
    return {k: round(v / 1e3, 4) for k, v in sorted(out.items())}

  def gaps(self, least_ms=5.0):
    """Idle stretches of the compute stream between the timed calls: (start ms, length ms, family before, family
    after), longest first, and the total. Start is measured from the first timed call."""
    self.torch.cuda.synchronize()
    if not self.events:
      return []
    t0 = self.events[0][1]
    spans = sorted((t0.elapsed_time(e0), t0.elapsed_time(e1), f) for f, e0, e1 in self.events)
    out, end, last = [], spans[0][1], spans[0][2]
    for a, b, f in spans[1:]:
      if a - end >= least_ms:
        out.append((round(end, 1), round(a - end, 1), last, f))
      if b > end:
        end, last = b, f
    small = 0.0
    end = spans[0][1]
    for a, b, f in spans[1:]:
      if 0 < a - end < least_ms:
        small += a - end
      end = max(end, b)
    return {"span_ms": round(end - spans[0][0], 1), "busy_ms": round(sum(b - a for a, b, _ in spans), 1),
            "idle_ms_in_gaps_of_5ms_and_more": round(sum(g[1] for g in out), 1), "idle_ms_in_shorter_gaps": round(small, 1),
            "longest": sorted(out, key=lambda g: -g[1])[:8]}

  def spans(self, lo_ms, hi_ms):
    """(family, start ms, end ms) of the timed calls that overlap [lo_ms, hi_ms], measured from the first timed call."""
    self.torch.cuda.synchronize()
    t0 = self.events[0][1]
    out = []
    for f, e0, e1 in self.events:
      a, b = t0.elapsed_time(e0), t0.elapsed_time(e1)
      if b >= lo_ms and a <= hi_ms:
        out.append((f, round(a, 1), round(b, 1)))
    return sorted(out, key=lambda r: r[1])

  def calls(self, family):
    """Milliseconds of every call of one family, in call order."""
    self.torch.cuda.synchronize()
    return [round(e0.elapsed_time(e1), 3) for f, e0, e1 in self.events if f == family]


def scratch_dir(need_bytes, fallback=None):
  """Where the float container and the result go: a memory-backed directory when it has room (the float
  file is written right before it is read: on a disk-backed file system its 16 GB of dirty pages are still
  being written back while the run reads them, which throttles the reader, not the quantizer)."""
  import shutil
  for d in ("/dev/shm",):
    try:
      if os.path.isdir(d) and shutil.disk_usage(d).free > need_bytes:
        return d
    except OSError:
      pass
  return fallback or os.environ.get("TMPDIR", "/tmp")


def prepare(layers=18, shapes=(D, DKV, DFF), workdir="/tmp"):
  """Writes the float container (rank 0; untimed) and returns its path on every rank."""
  import torch
  import torch.distributed as dist
  from mi355q import distributed as Dm
  rank, world = Dm._world()   # pylint: disable=protected-access
  d, dkv, dff = shapes
  src = os.path.join(workdir, f"c5_{layers}l_{d}_{dff}.litertlm")
  if rank == 0 and not os.path.exists(src):
    write_litertlm(build_model(layers, d, dkv, dff, weights=device_weights(torch)), src)
  if world > 1:
    dist.barrier()
  return src


def run(layers=18, sequences=128, tokens=512, variant="gptq", batch=1, workdir="/tmp", bits=4,
        shapes=(D, DKV, DFF), keep=False, phases=True, out_tokens=None, src=None, hessian="exact"):
  """One call -- litertlm_utils.quantize_litertlm(container, recipe, out, calibration_data=...) -- timed
  from opening the container to the written file. Returns a dict (rank 0) or None. Works under
  torchrun: samples and ops are sharded over the ranks, Hessians reduced to their owners."""
  import torch
  import torch.distributed as dist
  from mi355q import distributed as Dm, ops, runtime as rt
  from mi355q.utils import litertlm_utils
  rank, world = Dm._world()   # pylint: disable=protected-access
  d, dkv, dff = shapes
  t_build = time.perf_counter()
  own_src = src is None
  if own_src:
    src = prepare(layers, shapes, workdir)
  t_build = time.perf_counter() - t_build
  dst = os.path.join(workdir, f"c5_{layers}l_{d}_{dff}_{variant}_q.litertlm")
  rcp = recipe(variant, bits)
  need_cal = variant != "hadamard"
  data = None
  if need_cal:
    mine = Dm.sample_shard(sequences // batch, rank, world)
    if world == 1:
      full = calibration_set(torch, layers, sequences, tokens, d, dkv, dff, batch, out_tokens)
    else:      # every rank generates only its own share (the shard calibrate_sharded will walk)
      full = [None] * (sequences // batch)
      part = calibration_set(torch, layers, len(mine) * batch, tokens, d, dkv, dff, batch, out_tokens)
      for k, s in zip(mine, part):
        full[k] = s
    data = {0: {"serving_default": full}}
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  if os.path.exists(dst) and rank == 0:
    os.remove(dst)
  timer = GpuPhases(torch, ops) if phases else None
  if timer:
    timer.__enter__()
  stats = {}
  issued_before = len(Dm.ISSUED)      # (the list is a process-wide log: count this call's product reduces only)
  prof = None
  if os.environ.get("MI355Q_C5_PROFILE") and rank == 0:      # where the host's time goes (slows the run down)
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
  t0 = time.perf_counter()
  with ops.hessian_product(hessian):      # "exact": the default three-way bfloat16 split; "fast": the opt-in two-way float16 one
    n_out = litertlm_utils.quantize_litertlm(src, rcp, dst, overwrite=True, calibration_data=data, stats=stats)
  rt.mark("returned (host)")
  torch.cuda.synchronize()
  rt.mark("device drained (host)")
  if prof is not None:
    import pstats
    prof.disable()
    pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(32)
  if world > 1:
    dist.barrier()
  t2 = time.perf_counter()
  busy = timer.totals() if timer else None
  trace = {f: timer.calls(f) for f in (os.environ.get("MI355Q_C5_TRACE", "").split(",")) if f} if timer else None
  if timer:
    timer.__exit__()
  del data
  # every rank's own account (a scaling run is read rank by rank: who waited for whom)
  mine_rec = dict(rank=rank, samples=len(Dm.sample_shard(sequences // batch, rank, world)) if need_cal else 0,
                  calibrate_s=round(stats.get("calibrate_s", 0.0), 3), quantize_and_write_s=round(stats.get("quantize_and_write_s", 0.0), 3),
                  gpu_busy_s=None if busy is None else round(sum(busy.values()), 3), gpu_busy_by_family_s=busy)
  per_rank = [mine_rec]
  if world > 1:
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine_rec)
  if rank != 0:
    return None
  per = projections(d, dkv, dff)
  weight_bytes = layers * 4 * sum(r * c for _, r, c, _ in per)
  _, _, plan, owner, costs = Dm.plan_model_shards(litertlm_utils.LiteRTLMFile(src).get_section_buffer(0), rcp, world)
  loads = Dm.plan_loads(costs, owner, world)
  out = dict(
      hessian_product=hessian,
      workload=f"C5 {variant}: {layers} Gemma-2B-shaped layers (d={d}, kv={dkv}, ff={dff}) in a .litertlm,"
               f" {sequences} x {tokens} calibration tokens resident in HBM, int{bits} channelwise, one"
               " quantize_litertlm(calibration_data=...) call",
      ranks=world, seconds=round(t2 - t0, 3), calibrate_s=round(stats.get("calibrate_s", 0.0), 3),
      quantize_and_write_s=round(stats.get("quantize_and_write_s", 0.0), 3), repack_s=round(stats.get("repack_s", 0.0), 3),
      s_per_layer=round((t2 - t0) / layers, 4), weight_bytes=weight_bytes,
      weight_GBps=round(weight_bytes / (t2 - t0) / 1e9, 2), out_bytes=n_out, build_s=round(t_build, 2),
      section_bytes=stats.get("section_bytes"), expected_section_bytes=stats.get("expected_section_bytes"),
      gpu_busy_s_rank0=busy, gpu_busy_total_s=None if busy is None else round(sum(busy.values()), 3),
      gpu_busy_frac=None if busy is None else round(sum(busy.values()) / (t2 - t0), 3),
      trace=trace or None, per_rank=per_rank,
      hbm=dict(device_allocations=torch.cuda.memory_stats().get("num_device_alloc"), peak_reserved_GiB=round(torch.cuda.max_memory_reserved() / 2**30, 1),
               peak_allocated_GiB=round(torch.cuda.max_memory_allocated() / 2**30, 1)),
      idle_gaps=(timer.gaps() if timer and os.environ.get("MI355Q_C5_GAPS") else None),
      spans=(timer.spans(*[float(v) for v in os.environ["MI355Q_C5_SPANS"].split(",")]) if timer and os.environ.get("MI355Q_C5_SPANS") else None),
      plan=dict(modelled_s_per_rank=[round(v, 4) for v in loads],
                makespan_over_mean=round(max(loads) / (sum(loads) / world), 3) if sum(loads) else None,
                x2=Dm.x2_reduce_plan(plan, owner, costs, world),
                x2_issued=len(Dm.ISSUED) - issued_before))
  from mi355q import runtime as rt
  if rt.TIMELINE:      # MI355Q_TIMELINE=1: (label, ms since the call began when the host got there, ms when the GPU had drained if waited for, hipMallocs so far)
    out["timeline"] = [(label, round((a - t0) * 1e3, 1), round((b - t0) * 1e3, 1), n) for label, a, b, n in rt.TIMELINE]
    del rt.TIMELINE[:]
  if os.path.exists(dst) and not keep:
    os.remove(dst)
  if own_src and not keep and os.path.exists(src):
    os.remove(src)
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--layers", type=int, default=18)
  ap.add_argument("--sequences", type=int, default=128)
  ap.add_argument("--tokens", type=int, default=512)
  ap.add_argument("--batch", type=int, default=1, help="sequences per calibration sample (the sample's leading dim)")
  ap.add_argument("--variant", default="gptq", choices=("gptq", "mixed", "hadamard"))
  ap.add_argument("--bits", type=int, default=4)
  ap.add_argument("--dir", default=None, help="default: /dev/shm when it has room, else $TMPDIR")
  ap.add_argument("--keep", action="store_true")
  ap.add_argument("--no-phases", action="store_true")
  ap.add_argument("--hessian", default="exact", choices=("exact", "fast"), help="Hessian product: three-way bf16 split / two-way f16 split")
  a = ap.parse_args()
  import __graft_entry__ as g
  g.build()
  from mi355q import distributed as Dm
  rank, world = Dm.init()
  workdir = a.dir or scratch_dir(a.layers * 1000 * (1 << 20))
  res = run(a.layers, a.sequences, a.tokens, a.variant, a.batch, workdir, a.bits, keep=a.keep, phases=not a.no_phases, hessian=a.hessian)
  if rank == 0:
    print(json.dumps(res), flush=True)
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
