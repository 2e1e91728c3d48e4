#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: weight-bytes quantized / second and
% of the HBM roofline for per-channel int8 symmetric requantization of
4096x4096 FP32 weight buffers (BASELINE config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: ONE launch of
mi355q_requant_sym_f32_batched over POOL distinct, HBM-resident 4096x4096 FP32
buffers (POOL x 64 MiB = 1 GiB > the 256 MiB Infinity Cache, so no step can be
served from cache). Multi-GPU shards whole tensor-buffers: every rank owns its
own pool (weak scaling, no collective on the data path).

One JSON line is printed by rank 0 (contract in the task statement) with
`roofline` (dominant kernel, HIP-event timed) and `cpu_baseline` (the NumPy
oracle timed on the host, baseline only).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
# PyTorch alone makes 64 streams at its first torch.cuda.Stream(): streams that share a queue run in turn. The
# library's look-ahead stream and the lanes of the batched Hessian inverse want queues of their own
# (include/mi355q.h, mi355q_prepare_device); it must be set before the runtime is loaded (import torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)

ONE_GPU_HOSTS = os.environ.get("MI355Q_BENCH_ONE_GPU_HOSTS", "") == "1"
ROWS = COLS = 4096
POOL = 16
PCT_LAUNCHES = 200     # launches the per-launch percentiles are taken over (>= SURVEY 8d's 100)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
# Algorithmic bytes per 4096x4096 buffer: one FP32 read, int8 write,
# f32 scale write, int8 zero-point write (zero points are implicit here but the
# byte count follows the survey's definition).
ALG_BYTES = ROWS * COLS * 4 + ROWS * COLS + ROWS * 4 + ROWS


def parse():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=None,
                 help="ranks = GPUs of this node; a plain `python bench.py --gpus N` starts the N ranks itself")
  p.add_argument("--steps", type=int, default=200)
  p.add_argument("--warmup", type=int, default=20)
  p.add_argument("--cpu-seconds", type=float, default=10.0,
                 help="target CPU time for the cpu_baseline sample (0 disables)")
  p.add_argument("--extras", type=int, default=1, help="also time configs 3 / int4 (untimed region)")
  return p.parse_args()


def cpu_baseline(seconds: float):
  """Oracle (NumPy restatement, kind='port') on the host: same workload, bounded sample."""
  if seconds <= 0:
    return None
  import numpy as np
  from oracle import aeq_oracle as O
  w = np.random.default_rng(1234).standard_normal((ROWS, COLS), dtype=np.float32)
  O.min_max_quant_params(w[:256], 8, True, "CHANNELWISE")  # warm caches / imports
  reps, t0 = 0, time.perf_counter()
  while True:
    O.min_max_quant_params(w, 8, True, "CHANNELWISE")
    reps += 1
    dt = time.perf_counter() - t0
    if dt >= seconds or reps >= 200:
      break
  return {"value": round(reps * ROWS * COLS * 4 / dt / 1e9, 4), "unit": "GB/s", "cores": 1,
          "kind": "port",
          "sample": f"{reps} x get_tensor_quant_params(4096x4096 f32, int8 sym CHANNELWISE) via"
                    f" oracle/aeq_oracle.py (NumPy {np.__version__}, single thread,"
                    f" host has {os.cpu_count()} cpus) in {dt:.2f}s"}


def per_launch_ms(fn, iters: int):
  """Duration of every single fn() launch (one HIP event pair each, on the launch stream):
  sorted list of milliseconds. SURVEY 8d asks for median / p10 / p90, not only the mean."""
  import torch
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in evs:
    a.record()
    fn()
    b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a, b in evs)


def pct(sorted_ms, q: float) -> float:
  return sorted_ms[min(len(sorted_ms) - 1, int(q * len(sorted_ms)))]


def event_time_ms(fn, iters: int) -> float:
  """Average duration of fn() measured with HIP events on the launch stream."""
  import torch
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  start.record()
  for _ in range(iters):
    fn()
  end.record()
  end.synchronize()
  return start.elapsed_time(end) / iters


MFMA_F32_PEAK_TF, MFMA_F64_PEAK_TF = 157.3, 78.6   # dense MFMA peaks (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0


class ClockSampler:
  """Shader clock (MHz) and socket power (W) of GPU `index` while a block runs, from the amdgpu hwmon files
  (freq1_input, power1_average / power1_input), sampled every 20 ms by a helper thread. The Hessian products run at the
  socket's power limit: their time moves with the clock a box sustains, and this is what makes a box-to-box spread
  attributable. All fields None when the files are not there."""

  def __init__(self, index: int = 0, expect_s: float = 0.0):
    import glob
    self.expect_s = expect_s      # > 0: also count shader clocks from inside (mi355q_clock_probe) for about that long
    self.probe = None
    self.freq = self.power = None
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
      mons = sorted(glob.glob(card + "/hwmon/hwmon*"))
      if not mons or not os.path.exists(mons[0] + "/freq1_input"):
        continue
      if index == 0:
        self.freq = mons[0] + "/freq1_input"
        self.power = next((mons[0] + "/" + n for n in ("power1_average", "power1_input") if os.path.exists(mons[0] + "/" + n)), None)
        break
      index -= 1
    self.mhz, self.watts, self._stop, self._thread = [], [], False, None

  @staticmethod
  def _read(path):
    try:
      with open(path) as fh:
        return float(fh.read().strip())
    except (OSError, ValueError):
      return None

  def __enter__(self):
    if self.expect_s > 0:
      # the hwmon clock is a constant on these boxes: one wave on a stream of its own counts shader clocks against the
      # constant 100 MHz counter while the timed kernels run (include/mi355q.h: mi355q_clock_probe)
      import ctypes
      import torch
      from mi355q import _ffi
      each = min(0.03, self.expect_s)
      n = max(1, int(0.8 * self.expect_s / each))
      side = torch.cuda.Stream()
      out = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
      torch.cuda.synchronize()
      for i in range(n):
        _ffi.check(_ffi.lib().mi355q_clock_probe(each, ctypes.c_void_p(out[i].data_ptr()), ctypes.c_void_p(side.cuda_stream)))
      self.probe = (side, out)
    if self.freq is not None:
      import threading

      def loop():
        while not self._stop:
          f = self._read(self.freq)
          if f:
            self.mhz.append(f / 1e6)
          w = self._read(self.power) if self.power else None
          if w:
            self.watts.append(w / 1e6)
          time.sleep(0.02)
      self._thread = threading.Thread(target=loop, daemon=True)
      self._thread.start()
    return self

  def __exit__(self, *exc):
    self._stop = True
    if self._thread is not None:
      self._thread.join(1.0)

  def summary(self) -> dict:
    inside = {}
    if self.probe is not None:
      side, out = self.probe
      side.synchronize()
      mhz = [100.0 * c / t for c, t in out.cpu().tolist() if t > 0]
      inside = {"shader_MHz_counted_in_kernel_mean": round(sum(mhz) / len(mhz)) if mhz else None,
                "shader_MHz_counted_in_kernel_min": round(min(mhz)) if mhz else None, "probes": len(mhz)}
    if not self.mhz:
      return {"sclk_MHz_mean": None, "sclk_MHz_min": None, "socket_W_mean": None, "samples": 0, **inside}
    return {"sclk_MHz_mean": round(sum(self.mhz) / len(self.mhz)), "sclk_MHz_min": round(min(self.mhz)),
            "socket_W_mean": round(sum(self.watts) / len(self.watts)) if self.watts else None, "samples": len(self.mhz), **inside}


def hinv_roofline(d: int, ms: float) -> dict:
  """The damped inverse against the matrix cores it runs on: the Cholesky factorization (d^3 / 3) is FP64 MFMA at every
  order; the triangular inverse and L^-T L^-1 (2 d^3 / 3, single precision in the reference) are FP64 MFMA below d = 4096 and
  six bf16 MFMA products per float32 product from there on. frac = the time those flops take at the dense peaks / the time
  measured (pricing all of d^3 at the FP64 peak, as rounds 2 - 3 did, gives > 1 once two thirds of it left the FP64 pipe)."""
  chol, rest = d ** 3 / 3.0, 2.0 * d ** 3 / 3.0
  if d >= 4096:
    ideal_ms = chol / (MFMA_F64_PEAK_TF * 1e9) + 6.0 * rest / (MFMA_BF16_PEAK_TF * 1e9)
    how = "Cholesky d^3 / 3 at the FP64 MFMA peak + triangular inverse and L^-T L^-1 (2 d^3 / 3) as 6 bf16 MFMA products each at the bf16 peak"
  else:
    ideal_ms = (chol + rest) / (MFMA_F64_PEAK_TF * 1e9)
    how = "d^3 at the FP64 MFMA peak (everything in FP64 below d = 4096)"
  return {"bound": "mfma", "achieved": round(d ** 3 / ms / 1e9, 1), "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s (d^3 / time)",
          "frac": round(ideal_ms / ms, 4), "ideal_ms_at_peaks": round(ideal_ms, 3), "flops": how}


def timed_ms(torch, fn, reps: int, warm: int = 2) -> float:
  for _ in range(warm):
    fn()
  return event_time_ms(fn, reps)


def more_extras(torch, ops, gen, xs) -> dict:
  """The other BASELINE configurations at their own rooflines, the product path through the
  public interface, and file in -> file out. All outside the timed region; a few seconds."""
  import numpy as np
  out = {}
  # ---- C2 / C3 through the public interface: get_tensor_quant_params inside the batching
  # context ParamsGenerator runs in (weights resident in HBM)
  from mi355q import qtyping as q, requant_queue, runtime as rt
  from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as mm
  api = {}
  for label, pool, bits, gran, alg in (
      ("c2_int8_channelwise_4096x4096", xs, 8, "CHANNELWISE", ALG_BYTES),
      ("c3_int4_blockwise128_4096x11008",
       [torch.randn((4096, 11008), generator=gen, device="cuda") * 0.02 for _ in range(8)], 4, "BLOCKWISE_128",
       4096 * 11008 * 4 + 4096 * 11008 // 2 + 4096 * 11008 // 128 * 2)):
    cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
    info = q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                    op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg))
    res = [rt.HbmArray(t) for t in pool] * (64 // len(pool))
    best = None
    for _ in range(5):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      with requant_queue.batching() as queue:
        for r in res:
          mm.get_tensor_quant_params(info, cfg, r)
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / len(res)
      best = dt if best is None else min(best, dt)
    api[label] = {"us_per_tensor": round(best * 1e6, 2), "launches": queue.stats["launches"],
                  "tensors": queue.stats["tensors"], "weight_GBps": round(pool[0].numel() * 4 / best / 1e9, 1),
                  "hbm_frac": round(alg / best / 1e9 / HBM_PEAK_GBS, 4)}
    del res, pool
  out["api_resident"] = dict(api, note="wall time per tensor of get_tensor_quant_params(resident weight) inside"
                                       " requant_queue.batching(): Python + one batched launch per 16 tensors of a shape")
  # ---- file in -> file out (PCIe, page cache and the flatbuffer writer included)
  try:
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import file_bench
    from mi355q import quantizer, recipe
    with tempfile.TemporaryDirectory() as d:
      src, dst = os.path.join(d, "in.tflite"), os.path.join(d, "out.tflite")
      layers, rows, cols = 8, 4096, 11008
      file_bench.build_model(src, layers, rows, cols)
      best = None
      for _ in range(3):
        if os.path.exists(dst):
          os.remove(dst)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        quantizer.Quantizer(src, recipe.dynamic_wi4b128_afp32()).quantize(serialize_to_path=dst)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
      wbytes = layers * rows * cols * 4
      alg = layers * (rows * cols * 4 + rows * cols // 2 + rows * cols // 128 * 2)
      out["file_to_file"] = {"workload": f"{layers} x FC {rows}x{cols} fp32 .tflite -> int4 blockwise-128 .tflite",
                             "seconds": round(best, 4), "weight_GBps": round(wbytes / best / 1e9, 2),
                             "hbm_frac": round(alg / best / 1e9 / HBM_PEAK_GBS, 5),
                             "note": "host-bound: mmap'd file -> PCIe -> HBM -> PCIe -> mmap'd file"}
  except Exception as e:  # noqa: BLE001 - an extra must never cost the headline
    out["file_to_file"] = {"error": repr(e)[:200]}
  # ---- host <-> device staging of one C2 buffer (reported separately, never in `value`)
  hostw = np.random.default_rng(0).standard_normal((ROWS, COLS), dtype=np.float32)
  pinned = torch.from_numpy(hostw).pin_memory()
  devw, q8 = torch.empty((ROWS, COLS), device="cuda"), torch.empty((ROWS, COLS), dtype=torch.int8, device="cuda")
  hq = torch.empty((ROWS, COLS), dtype=torch.int8).pin_memory()

  def wall(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
  out["pcie"] = {
      "h2d_pageable_GBps": round(hostw.nbytes / wall(lambda: devw.copy_(torch.from_numpy(hostw))) / 1e9, 1),
      "h2d_pinned_GBps": round(hostw.nbytes / wall(lambda: devw.copy_(pinned, non_blocking=True)) / 1e9, 1),
      "d2h_pinned_int8_GBps": round(q8.numel() / wall(lambda: hq.copy_(q8, non_blocking=True)) / 1e9, 1),
      "note": "one 4096x4096 buffer: 64 MiB FP32 up, 16 MiB int8 down"}
  del pinned, devw, q8, hq
  # ---- C4: activation min/max, 128 tensors of [1, 256, 4096] per launch
  acts = [torch.randn((1, 256, 4096), generator=gen, device="cuda") * (1 + i / 8) for i in range(128)]
  amm = ops.ActMinMaxBatch([a.reshape(-1) for a in acts])
  ms = timed_ms(torch, amm.run, 50, 10)
  nbytes = sum(a.numel() for a in acts) * 4
  out["c4_act_minmax"] = {"ms_per_launch": round(ms, 5), "tensors_per_launch": len(acts),
                          "roofline": {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                       "alg_bytes_per_launch": nbytes}}
  del acts, amm
  # ---- OCTAV / Hadamard on the C5 rotation shape (4096 x 4096, int4 channelwise)
  w = xs[0] * 0.02
  ms = timed_ms(torch, lambda: ops.octav_clip(w.view(-1), 4096, 4096, 4, 10, 3.0, True, True), 20, 3)
  out["octav_clip_4096x4096_int4"] = {"ms": round(ms, 4), "hbm_frac_of_one_read": round(ROWS * COLS * 4 / ms / 1e6 / HBM_PEAK_GBS, 4),
                                      "note": "bit-exact NumPy-order masked sums, 10 Newton iterations, sigma = 0.02"}
  ms = timed_ms(torch, lambda: ops.octav_clip(w.view(-1), 4096 * 32, 128, 4, 10, 3.0, True, True), 20, 3)
  out["octav_clip_4096x4096_int4_blockwise128"] = {"ms": round(ms, 4),
                                                   "hbm_frac_of_one_read": round(ROWS * COLS * 4 / ms / 1e6 / HBM_PEAK_GBS, 4),
                                                   "note": "bit-exact, a lane per block (octav_unit_lanes_kernel; round 5: octav_groups_kernel 0.217 ms)"}
  ms = timed_ms(torch, lambda: ops.octav_clip(w.view(-1), 4096 * 128, 32, 4, 10, 3.0, True, True), 20, 3)
  out["octav_clip_4096x4096_int4_blockwise32"] = {"ms": round(ms, 4),
                                                  "hbm_frac_of_one_read": round(ROWS * COLS * 4 / ms / 1e6 / HBM_PEAK_GBS, 4)}
  # ... and the opt-in one-read kernel (ops.octav_mode("fast"), tolerance class T2): the row stays in registers
  with ops.octav_mode("fast"):
    ms = timed_ms(torch, lambda: ops.octav_clip(w.view(-1), 4096, 4096, 4, 10, 3.0, True, True), 20, 3)
    out["octav_clip_4096x4096_int4_fast"] = {"ms": round(ms, 4), "hbm_frac_of_one_read": round(ROWS * COLS * 4 / ms / 1e6 / HBM_PEAK_GBS, 4),
                                             "note": "opt-in (T2: scales within 1e-6, not bit-exact): one pass over HBM, float32 tree sums"}
    w2 = (torch.randn((2048, 16384), generator=gen, device="cuda") * 0.02).contiguous()
    ms = timed_ms(torch, lambda: ops.octav_clip(w2.view(-1), 2048, 16384, 4, 10, 3.0, True, True), 20, 3)
    out["octav_clip_2048x16384_int4_fast"] = {"ms": round(ms, 4), "hbm_frac_of_one_read": round(w2.numel() * 4 / ms / 1e6 / HBM_PEAK_GBS, 4)}
    del w2
  ms = timed_ms(torch, lambda: ops.hadamard_rotate(w, 4096), 50, 5)
  out["hadamard_4096x4096"] = {"ms": round(ms, 5), "roofline": {"bound": "hbm", "achieved": round(2 * ROWS * COLS * 4 / ms / 1e6, 1),
                                                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                               "frac": round(2 * ROWS * COLS * 4 / ms / 1e6 / HBM_PEAK_GBS, 4)}}
  # ---- C5: GPTQ at the Gemma-2B shapes (MFMA-bound; utilisation counters: profiles/r0*_gptq_mfma_util.txt)
  c5 = {}
  for d, tokens in ((2048, 65536), (16384, 16384)):
    x = torch.randn((tokens, d), generator=gen, device="cuda")
    reps = 10 if d == 2048 else 3
    # (the d = 16384 product is repeated for ~0.3 s so that the clock sampler sees the sustained state, not the ramp)
    reps_h = reps if d == 2048 else 12
    expect = (reps_h + 1) * (0.0023 if d == 2048 else 0.023)
    with ClockSampler(expect_s=expect) as clk:
      ms_h = timed_ms(torch, lambda: ops.gptq_xtx(x, 2.0 / 128), reps_h, 1)
    with ops.hessian_product("fast"), ClockSampler(expect_s=0.6 * expect) as clk_fast:
      ms_h_fast = timed_ms(torch, lambda: ops.gptq_xtx(x, 2.0 / 128), reps_h, 1)
    h = ops.gptq_xtx(x, 2.0 / 128)
    del x
    ms_i = timed_ms(torch, lambda: ops.gptq_hinv(h, 0.01), reps, 1)
    hinv, _ = ops.gptq_hinv(h, 0.01)
    rows = 2048
    wq = torch.randn((rows, d), generator=gen, device="cuda") * 0.02
    sc = (wq.abs().amax(dim=1) / 7).contiguous()
    ms_a = timed_ms(torch, lambda: ops.gptq_apply(wq, hinv, sc, None, 1, 0, 4, False, False, 8), reps, 1)
    # the triangular product computes half of 2 n d^2; the inverse is d^3 (Cholesky + triangular inverse + product)
    c5[f"d{d}"] = {
        "hessian": {"ms": round(ms_h, 3), "tokens": tokens,
                    "roofline": {"bound": "mfma", "achieved": round(6 * tokens * d * d / ms_h / 1e9, 1), "peak": MFMA_BF16_PEAK_TF,
                                 "unit": "TFLOP/s", "frac": round(6 * tokens * d * d / ms_h / 1e9 / MFMA_BF16_PEAK_TF, 4),
                                 "flops": "6 n d^2: lower triangle of X^T X, every float32 product as six bf16 MFMA products (the exact"
                                          " three-way bfloat16 split of both operands, xtx_bf16x3.hip: the default since round 4 -- with"
                                          " it the d = 16384 GPTQ chain reproduces the oracle's integers, profiles/r04_parity_rates.txt)",
                                 "float32_product_TFLOPs": round(tokens * d * d / ms_h / 1e9, 1),
                                 "mfma_f32_peak": MFMA_F32_PEAK_TF},
                    "clock_while_timed": clk.summary(),
                    "fast_f16x2": {"ms": round(ms_h_fast, 3), "how": "ops.hessian_product('fast') / MI355Q_XTX_F16X2=1",
                                   "clock_while_timed": clk_fast.summary(),
                                   "roofline_frac": round(3 * tokens * d * d / ms_h_fast / 1e9 / MFMA_BF16_PEAK_TF, 4),
                                   "note": "two-way float16 split, three f16 MFMA products per float32 product, 22-23 of 24 mantissa"
                                           " bits: 1.2e-3 of the d = 16384 integers differ from the oracle's (its own re-ordering"
                                           " floor: 1.5e-3); opt-in"}},
        "hinv": {"ms": round(ms_i, 3),
                 "roofline": hinv_roofline(d, ms_i)},
        "apply_2048_rows_int4": {"ms": round(ms_a, 3),
                                 "roofline": ({"bound": "mfma", "achieved": round(2 * rows * d * d / ms_a / 1e9, 1),
                                               "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                               "frac": round(2 * rows * d * d / ms_a / 1e9 / MFMA_F32_PEAK_TF, 4),
                                               "note": "FP32 MFMA; latency-bound on the column-serial quantize -> divide -> update chain"}
                                              if d < 4096 else
                                              {"bound": "mfma", "achieved": round(6 * 2 * rows * d * d / ms_a / 1e9, 1),
                                               "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                               "frac": round(6 * 2 * rows * d * d / ms_a / 1e9 / MFMA_BF16_PEAK_TF, 4),
                                               "float32_product_TFLOPs": round(2 * rows * d * d / ms_a / 1e9, 1),
                                               "note": "the update behind a group of columns runs as six bf16 MFMA products per float32"
                                                       " product (xtx_bf16x3.hip): priced against the bf16 pipe it uses; the column-serial"
                                                       " chain between the updates is latency-bound"})}}
    del h, hinv, wq, sc
  out["c5_gptq"] = c5
  return out


def roofline_summary(extras: dict) -> dict:
  """The figures a reader of the driver's record needs beside the 16-buffers-per-launch headline, small enough to
  live inside `roofline` (the driver keeps `roofline` whole and reduces `extras` to its keys): BASELINE config 2
  taken literally -- ONE 4096 x 4096 buffer per launch -- and one fraction of the bounding roofline per other
  configuration / algorithm, each computed from the entry of the same name under `extras`."""
  if not extras:
    return {}
  def get(*path):
    cur = extras
    for k in path:
      if not isinstance(cur, dict) or k not in cur:
        return None
      cur = cur[k]
    return cur
  out = {}
  one = get("single_buffer_launch")
  if one:
    out["single_buffer_launch"] = {"ms": one["ms"], "frac": one["hbm_frac"],
                                   "what": "BASELINE config 2 as worded: one 4096 x 4096 buffer per launch"}
  configs = {
      "c2_public_call": get("api_resident", "c2_int8_channelwise_4096x4096", "hbm_frac"),
      "c3": get("c3_blockwise128_int4_packed", "hbm_frac"),
      "c3_public_call": get("api_resident", "c3_int4_blockwise128_4096x11008", "hbm_frac"),
      "c4": get("c4_act_minmax", "roofline", "frac"),
      "hadamard": get("hadamard_4096x4096", "roofline", "frac"),
      "c5_hessian_d2048": get("c5_gptq", "d2048", "hessian", "roofline", "frac"),
      "c5_hessian_d16384": get("c5_gptq", "d16384", "hessian", "roofline", "frac"),
      "c5_hinv_d2048": get("c5_gptq", "d2048", "hinv", "roofline", "frac"),
      "c5_hinv_d16384": get("c5_gptq", "d16384", "hinv", "roofline", "frac"),
      "c5_apply_d2048": get("c5_gptq", "d2048", "apply_2048_rows_int4", "roofline", "frac"),
      "c5_apply_d16384": get("c5_gptq", "d16384", "apply_2048_rows_int4", "roofline", "frac"),
      "octav_exact_one_read": get("octav_clip_4096x4096_int4", "hbm_frac_of_one_read"),
      "octav_exact_b128_one_read": get("octav_clip_4096x4096_int4_blockwise128", "hbm_frac_of_one_read"),
      "octav_exact_b32_one_read": get("octav_clip_4096x4096_int4_blockwise32", "hbm_frac_of_one_read"),
      "octav_fast_one_read": get("octav_clip_4096x4096_int4_fast", "hbm_frac_of_one_read"),
      "mse_scale_one_read": get("mse_4096x4096_int4", "hbm_frac_of_one_read"),
      "mse_scale_and_quantize": get("mse_4096x4096_int4", "scale_and_quantize_hbm_frac"),
      "mse_public_call": get("mse_4096x4096_int4", "public_call_hbm_frac"),
      "oscar_clip_channelwise_one_read": get("oscar_4096x4096_int4_channelwise", "clip_bounds", "hbm_frac_of_one_read"),
      "oscar_clip_b128_one_read": get("oscar_4096x4096_int4_b128", "clip_bounds", "hbm_frac_of_one_read"),
      "oscar_clip_2048x16384_one_read": get("oscar_clip_bounds_2048x16384_channelwise", "hbm_frac_of_one_read"),
      "minmax_f32": get("minmax_f32_4096x4096_channelwise", "hbm_frac_of_one_read"),
      "quantize_f32": get("quantize_f32_4096x4096_int8_asymmetric", "hbm_frac"),
      "dequantize_f32": get("dequantize_f32_4096x4096_int8", "hbm_frac"),
  }
  out["configs"] = {k: v for k, v in configs.items() if v is not None}
  out["configs_note"] = ("fraction of the bounding roofline per configuration (HBM 8 TB/s; c5_hessian / c5_apply_d16384: bf16 MFMA"
                         " 2.5 PF; c5_hinv: FP64 / mixed pipes; c5_apply_d2048: f32 MFMA 157 TF); details under extras.<name>")
  return out


def round6_extras(torch, ops, gen, xs) -> dict:
  """The kernels of SURVEY 8 that had no timing since round 1 (VERDICT r05 missing #3): MSE (a14), OSCAR (f4) stage by
  stage and as the whole public call, and the non-fused route's three kernels (a1 min/max, a3 quantize with given
  parameters -- what TENSORWISE / asymmetric weights take --, f3 dequantize). Each against ONE read of the 4096 x 4096
  float32 buffer (64 MiB) -- or its own algorithmic bytes where it writes as much as it reads -- at the 8 TB/s HBM peak."""
  import numpy as np
  from mi355q import qtyping as q, runtime as rt
  from mi355q.algorithms.uniform_quantize import mse, oscar
  out = {}
  n = d = 4096
  w = (xs[1] * 0.02).contiguous()
  one_read = n * d * 4

  def frac(nbytes, ms):
    return round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)

  def info_cfg(bits, gran):
    cfg = q.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q.QuantGranularity[gran])
    return q.OpInfo(op=q.OperatorT(), op_name=q.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                    op_quant_config=q.OpQuantizationConfig(weight_tensor_config=cfg)), cfg

  def wall(fn, reps=5):
    best = None
    for _ in range(reps):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      fn()
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      best = dt if best is None else min(best, dt)
    return best
  # ---- a1 / a3 / f3 as separate launches (the route of TENSORWISE and asymmetric weights, and of dequantize)
  ms = timed_ms(torch, lambda: ops.minmax(w, 1, n, d), 50, 5)
  out["minmax_f32_4096x4096_channelwise"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms)}
  ms = timed_ms(torch, lambda: ops.minmax(w, 1, 1, n * d), 50, 5)
  out["minmax_f32_4096x4096_tensorwise"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms)}
  mn, mx = ops.minmax(w, 1, n, d)
  scale = ((mx - mn) / 255.0).contiguous()
  zp = torch.round(-128 - mn / scale).to(torch.int32)
  ms = timed_ms(torch, lambda: ops.quantize(w, 1, n, d, scale, zp, 8, False), 50, 5)
  out["quantize_f32_4096x4096_int8_asymmetric"] = {"ms": round(ms, 5), "alg_bytes": n * d * 5,
                                                   "hbm_frac": frac(n * d * 5, ms), "hbm_frac_of_one_read": frac(one_read, ms)}
  q8 = ops.quantize(w, 1, n, d, scale, zp, 8, False)
  ms = timed_ms(torch, lambda: ops.dequantize(q8, 1, n, d, scale, zp, 8), 50, 5)
  out["dequantize_f32_4096x4096_int8"] = {"ms": round(ms, 5), "alg_bytes": n * d * 5, "hbm_frac": frac(n * d * 5, ms)}
  del q8, mn, mx, zp
  # ---- a14 MSE: the order-exact row reduction, then the whole public call on a resident weight
  ms = timed_ms(torch, lambda: ops.mse_scale(w.view(-1), n, d, 0.37755), 50, 5)
  info, cfg = info_cfg(4, "CHANNELWISE")
  res = rt.HbmArray(w)
  mse.get_tensor_quant_params(info, cfg, res, None)
  sec = wall(lambda: mse.get_tensor_quant_params(info, cfg, res, None))
  ms_fused = timed_ms(torch, lambda: ops.mse_requant(w.view(-1), n, d, 0.37755, 4, False), 50, 5)
  out["mse_4096x4096_int4"] = {"scale_kernel_ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms),
                               "scale_and_quantize_kernel_ms": round(ms_fused, 5),
                               "scale_and_quantize_hbm_frac": frac(one_read + n * d + n * 4, ms_fused),
                               "public_call_ms": round(sec * 1e3, 4),
                               "public_call_hbm_frac": frac(one_read + n * d + n * 4, sec * 1e3),
                               "note": "scale = 0.37755 * sqrt(mean(x^2)) in NumPy's pairwise order (bit-exact) and clip(rint(x / scale)):"
                                       " ONE launch (mi355q_mse_requant_f32: the wave that summed a row quantizes it out of the L2),"
                                       " one read + one int8 write per element (round 5: two kernels, two reads)"}
  # ---- f4 OSCAR: stage by stage, then the whole public call (3 fixed-point iterations, clip search, quantize)
  rng = np.random.default_rng(1)
  mu2 = np.exp(rng.normal(size=d) * 1.5)
  s = ops._f64_dev(np.exp(rng.normal(size=d) * 0.2))   # pylint: disable=protected-access
  m = ops._f64_dev(mu2)   # pylint: disable=protected-access
  ms = timed_ms(torch, lambda: ops.oscar_col_sumsq(w, False), 20, 3)
  out["oscar_col_sumsq_4096x4096"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms)}
  for label, gran, g in (("channelwise", "CHANNELWISE", d), ("b128", "BLOCKWISE_128", 128)):
    rec = {}
    ms = timed_ms(torch, lambda: ops.oscar_group_terms(w, s, g), 20, 3)
    rec["group_terms"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms)}
    _, winner, wsq = ops.oscar_group_terms(w, s, g)
    ms = timed_ms(torch, lambda: ops.oscar_winner_energy(winner, wsq, d, g), 20, 3)
    rec["winner_energy"] = {"ms": round(ms, 5)}
    groups = d // g
    u = ops._f64_dev(np.full(groups, 0.01))   # pylint: disable=protected-access
    noise = ops._f64_dev(np.full(groups, 0.005))   # pylint: disable=protected-access
    ms = timed_ms(torch, lambda: ops.oscar_clip_bounds(w, s, m, g, u, noise, 7, g != d, True, True), 10, 2)
    rec["clip_bounds"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(one_read, ms),
                          "note": "stable descending sort of every group's |w| * s in LDS + the sequential FP64 running sums"
                                  " of the breakpoint scan (ref oscar.py:62-104)"}
    sc = ops._f64_dev(np.full(n * groups, 0.05))   # pylint: disable=protected-access
    ms = timed_ms(torch, lambda: ops.oscar_quantize(w, s, sc, g, -8, 7), 20, 3)
    rec["quantize"] = {"ms": round(ms, 5), "hbm_frac": frac(n * d * 5, ms)}
    info, cfg = info_cfg(4, gran)
    oscar.get_tensor_quant_params(info, cfg, res, {"mu2": mu2})
    sec = wall(lambda: oscar.get_tensor_quant_params(info, cfg, res, {"mu2": mu2}), 4)
    rec["public_call_ms"] = round(sec * 1e3, 4)
    rec["public_call_hbm_frac_of_one_read"] = frac(one_read, sec * 1e3)
    out[f"oscar_4096x4096_int4_{label}"] = rec
    del winner, wsq
  # the rotated down_proj shape's rows (16384 columns: the longest sort + scan a Gemma-2B layer has)
  w2 = (torch.randn((2048, 16384), generator=gen, device="cuda") * 0.02).contiguous()
  s2 = ops._f64_dev(np.exp(rng.normal(size=16384) * 0.2))   # pylint: disable=protected-access
  m2 = ops._f64_dev(np.exp(rng.normal(size=16384) * 1.5))   # pylint: disable=protected-access
  u1, n1 = ops._f64_dev(np.full(1, 0.01)), ops._f64_dev(np.full(1, 0.005))   # pylint: disable=protected-access
  ms = timed_ms(torch, lambda: ops.oscar_clip_bounds(w2, s2, m2, 16384, u1, n1, 7, False, True, True), 5, 2)
  out["oscar_clip_bounds_2048x16384_channelwise"] = {"ms": round(ms, 5), "hbm_frac_of_one_read": frac(w2.numel() * 4, ms)}
  return out


def sharded_configs(torch, dist, rank, world, workdir):
  """The BASELINE configurations that shard over ranks AND exchange something, through the public
  calls, at this run's N (after the timed region; N = 1 gives the first point of each curve):
    C3  32 x (4096 x 11008) blockwise-128 int4, ops sharded by cost, results gathered to rank 0, file written
    C4  512 samples x 32 activation tensors [1, 256, 4096], samples sharded, per-sample statistics
        all-gathered and replayed in dataset order
    C5  18 Gemma-2B-shaped layers in a .litertlm, GPTQ int4: plan -> samples sharded -> every Hessian
        reduced (packed lower triangle) to the rank that owns its ops -> applies -> gather -> file."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import c4_bench
  import c5_model
  import file_bench
  from mi355q import distributed as D, recipe

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()

  def timed(fn):
    """(seconds between the barriers = the job's time, every rank's own seconds before the closing barrier)."""
    barrier()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    per_rank = [round(own, 4)]
    if world > 1:
      per_rank = [None] * world
      dist.all_gather_object(per_rank, round(own, 4))
    return dt, per_rank
  out = {}
  # ---- C3
  layers, rows, cols = 32, 4096, 11008
  src, dst = os.path.join(workdir, "bench_c3_in.tflite"), os.path.join(workdir, "bench_c3_out.tflite")
  if rank == 0:
    file_bench.build_model(src, layers, rows, cols)
  barrier()
  rcp = recipe.dynamic_wi4b128_afp32()
  D.quantize_model_sharded(src, rcp)                       # page cache + clocks warm
  dt, per_rank = min((timed(lambda: D.quantize_model_sharded(src, rcp, serialize_to_path=dst)) for _ in range(2)), key=lambda r: r[0])
  _, _, _, owner, costs = D.plan_model_shards(src, rcp, world)
  loads = D.plan_loads(costs, owner, world)
  out["c3_32x4096x11008_int4_b128"] = {"seconds": round(dt, 4), "weight_GBps": round(layers * rows * cols * 4 / dt / 1e9, 2),
                                       "seconds_per_rank": per_rank,
                                       "plan": {"ops_per_rank": [owner.count(r) for r in range(world)],
                                                "makespan_over_mean": round(max(loads) / (sum(loads) / world), 3) if sum(loads) else None},
                                       "note": "file in (mmap, H2D) -> ops sharded by cost -> gather to rank 0 -> file out"}
  barrier()
  if rank == 0:
    for f in (src, dst):
      if os.path.exists(f):
        os.remove(f)
  # ---- C4
  model = c4_bench.build_model(32)
  gen = torch.Generator(device="cuda").manual_seed(40 + rank)
  pool = [{f"act{i}": torch.randn((1, 256, 4096), generator=gen, device="cuda") * (1 + i / 8) for i in range(32)}
          for _ in range(4)]
  data = {"serving_default": [pool[k % 4] for k in range(512)]}
  static = recipe.static_wi8_ai8()
  D.calibrate_sharded(model, static, {"serving_default": data["serving_default"][:2 * world]})
  dt, per_rank = timed(lambda: D.calibrate_sharded(model, static, data))
  out["c4_512_samples_static_wi8_ai8"] = {"seconds": round(dt, 4), "samples_per_s": round(512 / dt, 1),
                                          "seconds_per_rank": per_rank, "samples_per_rank": [len(D.sample_shard(512, r, world)) for r in range(world)],
                                          "activation_GBps": round(512 * 32 * 256 * 4096 * 4 / dt / 1e9, 1),
                                          "note": "samples resident in HBM; statistics all-gathered, replayed in dataset order"}
  del pool, data, model
  # ---- C5
  c5dir = c5_model.scratch_dir(20 << 30, workdir)
  layers = int(os.environ.get("MI355Q_BENCH_C5_LAYERS", 18))       # (a control-path run on a shared GPU takes 2)
  c5src = c5_model.prepare(layers, workdir=c5dir)
  # A quantizer process makes ONE whole-model call: `seconds` (and every other top-level field) of c5_gptq is the FIRST call of
  # this process -- it pays what no later one does (code objects of kernels only this path uses, 20 GiB of Hessian accumulators
  # and 8 GB of weight tensors as FRESH device memory, the page-locking of the io rings) -- and the same call made again is kept
  # beside it as `second_call_of_the_process`.
  first = c5_model.run(layers, 128, 512, "gptq", workdir=c5dir, src=c5src, phases=True, hessian="exact")
  for variant, hessian in (("gptq", "exact"), ("gptq", "fast"), ("mixed", "exact")):
    res = c5_model.run(layers, 128, 512, variant, workdir=c5dir, src=c5src, phases=True, hessian=hessian)
    if rank == 0:
      res.pop("trace", None)
      if variant == "gptq" and hessian == "exact":
        first.pop("trace", None)
        first["second_call_of_the_process"] = {k: res[k] for k in ("seconds", "calibrate_s", "quantize_and_write_s", "gpu_busy_total_s", "gpu_busy_frac")}
        res = first
      out[f"c5_{variant}" + ("_fast_hessian" if hessian == "fast" else "")] = res
  barrier()
  if rank == 0 and os.path.exists(c5src):
    os.remove(c5src)
  return out


def collective_probe(torch, dist, rank, world, backend):
  """N > 1 only, after the timed region: one all-gather of C4's per-sample statistics (128 KiB
  per rank set) and one 16 MiB all-reduce (a d = 2048 Hessian in float32 terms) through
  libmi355q's RCCL entry points, so a scaling run exercises xGMI and reports RCCL's own rank count."""
  if backend != "nccl":
    return {"transport": backend, "note": "RCCL needs one GPU per rank"}
  import ctypes
  from mi355q import _ffi, distributed as D, runtime as rt
  L = _ffi.lib()
  comm = D.rccl_comm()
  nr, rk = ctypes.c_int32(0), ctypes.c_int32(0)
  _ffi.check(L.mi355q_comm_info(comm, ctypes.byref(nr), ctypes.byref(rk)))
  n_local = 512 // world * 32 * 2                      # C4: 512 samples x 32 tensors x (min, max) over the ranks
  loc = torch.full((n_local,), float(rank), device="cuda")
  allv = torch.empty((world * n_local,), device="cuda")
  big = torch.ones((4 << 20,), device="cuda")

  def gather():
    _ffi.check(L.mi355q_allgather_minmax(comm, rt.ptr(loc), n_local, rt.ptr(allv), rt.stream_ptr()))

  def reduce():
    _ffi.check(L.mi355q_allreduce_sum_f32(comm, rt.ptr(big), big.numel(), rt.stream_ptr()))
  res = {}
  for name, fn, nbytes in (("allgather_minmax", gather, world * n_local * 4), ("allreduce_sum_f32_16MiB", reduce, 16 << 20)):
    fn()
    torch.cuda.synchronize()
    dist.barrier()
    ms = event_time_ms(fn, 10)
    res[name] = {"ms": round(ms, 4), "bytes": nbytes}
  ok = bool(torch.equal(allv.view(world, n_local)[:, 0].cpu(), torch.arange(world, dtype=torch.float32)))
  res.update(transport="rccl via libmi355q", rccl_ranks=nr.value, rccl_rank0=rk.value, allgather_correct=ok)
  return res


def free_port() -> int:
  import socket
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def launch_command(gpus: int, argv, port: int):
  """The command line a plain `python bench.py --gpus N` turns itself into: N ranks of this
  node, one per GPU, rendezvous on the loopback address (the container's hostname may not resolve)."""
  return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args) -> int:
  """`--gpus N` (N > 1) without a launcher's environment: start the N ranks here and hand back
  their exit status. Rank 0's JSON line goes to this process's stdout (inherited)."""
  import subprocess
  backend = os.environ.get("MI355Q_BENCH_BACKEND", "nccl")
  if backend == "nccl" and not ONE_GPU_HOSTS:
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
      raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (one rank per GPU over RCCL;"
                       " MI355Q_BENCH_ONE_GPU_HOSTS=1 runs the ranks as separate hosts on cuda:0 over RCCL's socket"
                       " transport, MI355Q_BENCH_BACKEND=gloo shares cuda:0 for a control-path check)")
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  env.setdefault("OMP_NUM_THREADS", "8")
  cmd = launch_command(args.gpus, sys.argv[1:], free_port())
  print("bench.py: starting " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
  return subprocess.run(cmd, env=env).returncode


def main():
  args = parse()
  if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
    sys.exit(spawn_ranks(args))
  import torch
  import torch.distributed as dist

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if args.gpus is not None and args.gpus != world:
    raise SystemExit(f"bench.py --gpus {args.gpus} inside a launcher with WORLD_SIZE={world}: they must agree")
  if not torch.cuda.is_available():
    raise SystemExit(f"bench.py needs a GPU (no CPU fallback on the product path); rank {rank} of {world}")
  # MI355Q_BENCH_BACKEND=gloo lets the N > 1 control path (barriers, max-reduce, rank-0 line) be
  # exercised on a one-GPU box: all ranks then share cuda:0 (RCCL refuses duplicate devices)
  backend = os.environ.get("MI355Q_BENCH_BACKEND", "nccl")
  if backend != "nccl":
    local = local % torch.cuda.device_count()
  if ONE_GPU_HOSTS and backend == "nccl" and world > 1:
    # RCCL with N > 1 on a one-GPU box: every rank names a host of its own (no "duplicate GPU"), the peers meet over RCCL's
    # socket transport on the loopback interface, all of them on cuda:0. Not xGMI and not a scaling figure (N ranks share
    # one GPU's HBM and CUs): it is the N > 1 code path -- collective probe, C3 / C4 / C5 sharded -- with real RCCL peers.
    from mi355q import distributed as D_
    os.environ.update(D_.one_gpu_ranks_env(rank))
    local = 0
  torch.cuda.set_device(local)
  # MI355Q_BENCH_FORCE_PROBE=1: run the N > 1 collective probe on a world of one as well (the only
  # way to exercise it on a one-GPU box)
  force_probe = os.environ.get("MI355Q_BENCH_FORCE_PROBE", "") == "1"
  if world > 1 or force_probe:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend == "nccl":
      dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
      dist.init_process_group(backend)

  import __graft_entry__ as g
  g.build()
  from mi355q import ops

  gen = torch.Generator(device="cuda")
  gen.manual_seed(1234 + rank)
  xs = [torch.randn((ROWS, COLS), generator=gen, device="cuda", dtype=torch.float32)
        for _ in range(POOL)]
  batch = ops.RequantBatch(xs, block=0, bits=8, want_q=True)

  def barrier():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  # Clock / power-state pre-warm: the first tens of milliseconds after idle run
  # 15-25 % slower on MI355X (measured with tools/kbench), so spin the same kernel
  # for ~0.3 s before the W contract warm-up steps. Untimed.
  t_pre = time.perf_counter()
  while time.perf_counter() - t_pre < 0.3:
    for _ in range(10):
      batch.run()
    torch.cuda.synchronize()
  for _ in range(args.warmup):
    batch.run()
  # ONE loop gives both figures: `ms_per_step` is the wall clock between the barriers, `roofline.launch_ms` the mean
  # of the HIP event pairs recorded around the SAME K launches on the launch stream -- the kernel time can then never
  # exceed the step time (two separate loops differed by 0.2 % the wrong way round in round 4)
  pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
  barrier()
  t0 = time.perf_counter()
  for a, b in pairs:
    a.record()
    batch.run()
    b.record()
  barrier()
  elapsed = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # --- per-launch duration of the dominant kernel: the event pairs of the timed launches themselves
  launch_sorted = sorted(a.elapsed_time(b) for a, b in pairs)
  kern_ms = sum(launch_sorted) / len(launch_sorted)
  achieved = POOL * ALG_BYTES / (kern_ms * 1e-3) / 1e9
  # SURVEY 8d asks for the percentiles of >= 100 launches; the driver's --steps may be 20. The step's work is never
  # changed: when K < PCT_LAUNCHES the same launch is repeated AFTER the timed region until that many event pairs
  # exist, and the percentiles are taken over all of them (`launch_ms` / `achieved` stay the mean of the K timed ones).
  all_sorted = launch_sorted
  if args.steps < PCT_LAUNCHES:
    all_sorted = sorted(launch_sorted + per_launch_ms(batch.run, PCT_LAUNCHES - args.steps))

  extras = {}
  if args.extras and rank == 0 and world == 1:
    # one 4096x4096 buffer per launch, rotating over the pool (launch-gap inclusive)
    outs = [ops.requant_sym(x, 0, 8) for x in xs[:2]]  # warm
    del outs
    single = [ops.RequantBatch([x], 0, 8) for x in xs]
    state = {"i": 0}

    def one():
      single[state["i"] % POOL].run()
      state["i"] += 1
    for _ in range(POOL):
      one()
    ms = event_time_ms(one, 320)
    extras["single_buffer_launch"] = {"ms": round(ms, 5),
                                      "weight_GBps": round(ROWS * COLS * 4 / ms / 1e6, 1),
                                      "hbm_frac": round(ALG_BYTES / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del single
    # config 3 shape: blockwise-128 int4 with fused packing, 4096x11008, pool of 6 (1 GiB)
    r3, c3 = 4096, 11008
    x3 = [torch.randn((r3, c3), generator=gen, device="cuda", dtype=torch.float32) * 0.02
          for _ in range(6)]
    b3 = ops.RequantBatch(x3, block=128, bits=4, want_q=False, want_packed=True, want_scale_f16=True)
    for _ in range(200):
      b3.run()
    ms3 = event_time_ms(b3.run, 100)
    alg3 = r3 * c3 * 4 + r3 * c3 // 2 + (r3 * c3 // 128) * 2
    extras["c3_blockwise128_int4_packed"] = {
        "ms_per_layer": round(ms3 / 6, 5),
        "weight_GBps": round(6 * r3 * c3 * 4 / ms3 / 1e6, 1),
        "hbm_frac": round(6 * alg3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "alg_bytes_per_layer": alg3}
    del b3, x3
    # int4 channelwise with fused packing on the C2 shape
    b4 = ops.RequantBatch(xs, block=0, bits=4, want_q=False, want_packed=True)
    for _ in range(100):
      b4.run()
    ms4 = event_time_ms(b4.run, 100)
    extras["c2_int4_packed"] = {"ms": round(ms4 / POOL, 5),
                                "weight_GBps": round(POOL * ROWS * COLS * 4 / ms4 / 1e6, 1)}
    del b4

  if args.extras and rank == 0 and world == 1:
    try:
      extras.update(more_extras(torch, ops, gen, xs))
    except Exception as e:  # noqa: BLE001 - an extra must never cost the headline
      import traceback
      extras["more_extras_error"] = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}
    try:
      extras.update(round6_extras(torch, ops, gen, xs))
    except Exception as e:  # noqa: BLE001 - an extra must never cost the headline
      import traceback
      extras["round6_extras_error"] = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}
  collectives = None
  probe_hung = False
  if world > 1 or force_probe:
    # The probe must never cost the headline: it runs in a helper thread with a deadline; a rank
    # whose probe does not come back reports that, skips the final barrier and leaves.
    import threading
    box = {}

    def run_probe():
      try:
        torch.cuda.set_device(local)
        box["result"] = collective_probe(torch, dist, rank, world, backend)
      except Exception as e:  # noqa: BLE001
        box["result"] = {"error": repr(e)[:300]}
    th = threading.Thread(target=run_probe, daemon=True)
    th.start()
    th.join(float(os.environ.get("MI355Q_BENCH_PROBE_SECONDS", "90")))
    probe_hung = th.is_alive()
    collectives = {"error": "collective probe did not finish in time"} if probe_hung else box.get("result")
  sharded = None
  if args.extras and not probe_hung:
    # same rule: a sharded configuration that wedges (first contact with N > 1 RCCL happens on the
    # driver's node) must not cost the headline -- helper thread, deadline, error string
    import threading
    sbox = {}

    def run_sharded():
      try:
        torch.cuda.set_device(local)
        sbox["result"] = sharded_configs(torch, dist, rank, world, os.environ.get("TMPDIR", "/tmp"))
      except Exception as e:  # noqa: BLE001
        import traceback
        sbox["result"] = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}
    ts = threading.Thread(target=run_sharded, daemon=True)
    ts.start()
    ts.join(float(os.environ.get("MI355Q_BENCH_SHARDED_SECONDS", "420")))
    if ts.is_alive():
      probe_hung = True
      sharded = {"error": "sharded configurations did not finish in time"}
    else:
      sharded = sbox.get("result")

  rccl_failed = False
  if rank == 0:
    total_bytes = world * args.steps * POOL * ROWS * COLS * 4
    traffic = traffic_source = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
      try:
        rec = json.load(open(pmc))
        traffic, traffic_source = rec.get("hbm_bytes_per_launch"), "recorded: " + rec.get("source", "profiles/")
      except Exception:  # noqa: BLE001
        traffic = traffic_source = None
    line = {
        "metric": "weight-bytes quantized/sec (GB/s), 4096x4096 per-channel int8",
        "value": round(total_bytes / elapsed / 1e9, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "per-channel int8 symmetric requant of 4096x4096 FP32 weight"
                               " buffers (BASELINE config 2)",
                   "buffers_per_step": POOL, "bytes_in_per_step": POOL * ROWS * COLS * 4,
                   "sharding": f"tensor-buffers x{world} (no collective)",
                   **({"ranks_share_one_gpu": "MI355Q_BENCH_ONE_GPU_HOSTS=1: N ranks as separate hosts on cuda:0 over RCCL's"
                                              " socket transport -- the N > 1 code path, not a scaling figure"}
                      if ONE_GPU_HOSTS and world > 1 else {})},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     # HBM bytes per launch from the rocprofv3 PMC passes recorded under profiles/
                     # (FETCH_SIZE / WRITE_SIZE with the gfx950 corrections): a RECORDED figure of the
                     # same kernel and launch shape, not a counter read during this run
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "requant_rows_kernel<8,256,4,ieee-div,batched,nt>",
                     "entry_point": "mi355q_requant_sym_f32_batched (what ParamsGenerator's batched"
                                    " loop issues, mi355q/requant_queue.py)",
                     "alg_bytes_per_launch": POOL * ALG_BYTES,
                     "launch_ms": round(kern_ms, 5),
                     "launch_ms_p10": round(pct(all_sorted, 0.10), 5),
                     "launch_ms_p50": round(pct(all_sorted, 0.50), 5),
                     "launch_ms_p90": round(pct(all_sorted, 0.90), 5),
                     "percentiles_over_launches": len(all_sorted),
                     **roofline_summary(extras)},
        "cpu_baseline": cpu_baseline(args.cpu_seconds) if world == 1 else None,
        "extras": extras,
        "collectives": collectives,
        "sharded": sharded,
    }
    # A scaling run whose ranks did not meet over RCCL is not a scaling run: say so in the line and fail the process
    # (the line is still printed -- the per-rank figures in it are real -- but nobody can take it for an N-GPU result)
    if world > 1 and backend == "nccl":
      seen = (collectives or {}).get("rccl_ranks")
      if seen != world or not (collectives or {}).get("allgather_correct", False):
        line["error"] = (f"RCCL reports {seen} rank(s) for --gpus {world} (mi355q_comm_info) or the all-gather check failed:"
                         f" {json.dumps(collectives)[:300]}")
        rccl_failed = True
    try:        # whatever native libraries (RCCL's version banner ...) left in C stdio goes out first:
      import ctypes   # the JSON line has to be the last line of stdout
      ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
      pass
    print(json.dumps(line), flush=True)
  if world > 1 or force_probe:
    sys.stdout.flush()
    if probe_hung:
      os._exit(3 if rccl_failed else 0)          # a wedged collective cannot be cancelled; the line is out
    # (no closing barrier: a rank whose peer left through the branch above must not wait for it)
    try:
      dist.destroy_process_group()
    except Exception:  # noqa: BLE001
      pass
  if rccl_failed:
    sys.exit(3)


if __name__ == "__main__":
  main()
