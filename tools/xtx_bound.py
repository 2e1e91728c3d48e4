#!/usr/bin/env python3
"""What bounds the GPTQ Hessian product (xtx_bf16x3_kernel, d = 16384 x 16384 tokens): ONE measurement with every
clock and power reading side by side, to be run twice on the same box -- plainly and under a rocprofv3 PMC pass:

  python tools/xtx_bound.py [seconds=3]
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d DIR -o p -- python tools/xtx_bound.py

One JSON line: milliseconds and bf16 TFLOP/s per product (HIP events, every product of the run), the shader clock
counted from INSIDE a kernel that runs beside the products (mi355q_clock_probe: s_memtime ticks per 100 MHz tick), and
what amd-smi / rocm-smi report for clock and socket power while they run (sampled by a thread).
"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def smi_sample():
  out = {}
  for cmd, key in ((["amd-smi", "metric", "-g", "0", "-p", "-c", "--json"], "amd-smi"),
                   (["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"], "rocm-smi")):
    try:
      r = subprocess.run(cmd, capture_output=True, text=True, timeout=5)
      out[key] = json.loads(r.stdout) if r.stdout.strip().startswith(("{", "[")) else r.stdout.strip()[:300]
    except Exception as e:  # pylint: disable=broad-except
      out[key] = {"error": str(e)[:120]}
  return out


def flatten(obj, prefix=""):
  if isinstance(obj, dict):
    for k, v in obj.items():
      yield from flatten(v, f"{prefix}{k}.")
  elif isinstance(obj, list):
    for i, v in enumerate(obj):
      yield from flatten(v, f"{prefix}{i}.")
  else:
    yield prefix[:-1], obj


def main():
  secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import _ffi, ops
  d = tokens = 16384
  x = torch.randn((tokens, d), device="cuda")
  for _ in range(2):
    ops.gptq_xtx(x, 1.0)
  torch.cuda.synchronize()
  idle = smi_sample()
  samples, stop = [], [False]

  def sampler():
    while not stop[0]:
      samples.append(smi_sample())
      time.sleep(0.05)
  th = threading.Thread(target=sampler, daemon=True)
  side = torch.cuda.Stream()
  n_probe = int(secs / 0.05)
  probes = torch.zeros((n_probe, 2), dtype=torch.int64, device="cuda")
  for i in range(n_probe):
    _ffi.check(_ffi.lib().mi355q_clock_probe(0.04, ctypes.c_void_p(probes[i].data_ptr()), ctypes.c_void_p(side.cuda_stream)))
  th.start()
  t0 = time.perf_counter()
  events = []
  while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gptq_xtx(x, 1.0)
    e1.record()
    e1.synchronize()
    events.append(e0.elapsed_time(e1))
  stop[0] = True
  th.join(6.0)
  side.synchronize()
  mhz = [100.0 * c / t for c, t in probes.cpu().tolist() if t > 0]
  ms = sorted(events)
  flops = 6.0 * tokens * d * d
  # every numeric smi field that looks like a clock or a power reading: mean over the samples taken under load
  fields = {}
  for s in samples:
    for k, v in flatten(s):
      lk = k.lower()
      if isinstance(v, (int, float)) and any(w in lk for w in ("power", "clk", "clock", "sclk", "gfx")) and "limit" not in lk and "max" not in lk and "min" not in lk:
        fields.setdefault(k, []).append(float(v))
      elif isinstance(v, str):
        head = v.strip().split(" ")[0].replace("Mhz", "").replace("MHz", "").strip("()")
        try:
          if any(w in lk for w in ("power", "clk", "clock", "sclk")):
            fields.setdefault(k, []).append(float(head))
        except ValueError:
          pass
  under_load = {k: round(sum(v) / len(v), 1) for k, v in fields.items() if v}
  print(json.dumps({
      "workload": "mi355q_gptq_xtx_f32, d = 16384, 16384 tokens, back to back for %.1f s" % secs,
      "products": len(ms), "ms_median": round(ms[len(ms) // 2], 3), "ms_p10": round(ms[len(ms) // 10], 3), "ms_p90": round(ms[(9 * len(ms)) // 10], 3),
      "bf16_TFLOPs_median": round(flops / ms[len(ms) // 2] / 1e9, 1),
      "mfma_busy_frac_if_clock_is_the_probe_s": round(flops / (ms[len(ms) // 2] * 1e-3) / (1.048576e6 * (sum(mhz) / len(mhz)) * 1e6), 3) if mhz else None,
      "shader_MHz_counted_in_kernel": {"mean": round(sum(mhz) / len(mhz)) if mhz else None, "min": round(min(mhz)) if mhz else None,
                                       "max": round(max(mhz)) if mhz else None, "probes": len(mhz)},
      "smi_under_load_mean": under_load, "smi_samples": len(samples),
      "smi_idle_before": {k: v for k, v in flatten(idle) if isinstance(v, (int, float, str)) and any(w in k.lower() for w in ("power", "clk", "sclk"))},
      "profiled": bool(os.environ.get("ROCPROFILER_REGISTER_ROOT") or os.environ.get("ROCP_TOOL_LIBRARIES") or any("rocprof" in v for v in os.environ.values() if isinstance(v, str)))}), flush=True)


if __name__ == "__main__":
  main()
