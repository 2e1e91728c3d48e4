"""Shader clock and power while the FP32 / FP64 MFMA GEMMs run back to back (rocm-smi sampled from
a thread).   python tools/clock_under_load.py [seconds=3]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
from mi355q import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
samples = []
stop = False


def sampler():
  while not stop:
    try:
      out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
      samples.append((time.perf_counter(), json.loads(out)))
    except Exception as e:  # pylint: disable=broad-except
      samples.append((time.perf_counter(), {"error": str(e)}))
    time.sleep(0.05)


def pick(js):
  card = js.get("card0", {})
  return {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower()}


for name, fn, flops in (
    ("xtx f32 d=16384 n=16384", lambda x=torch.randn((16384, 16384), device="cuda"): ops.gptq_xtx(x, 1.0), 16384.0 ** 3),
    ("gemm f64 8192^3", lambda a=torch.randn((8192, 8192), device="cuda", dtype=torch.float64): ops.gemm(a, a), 2.0 * 8192.0 ** 3)):
  fn(); torch.cuda.synchronize()
  time.sleep(1.0)
  samples.clear(); stop = False
  th = threading.Thread(target=sampler); th.start()
  time.sleep(0.3)
  t0 = time.perf_counter(); n = 0; marks = []
  while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    marks.append((time.perf_counter() - t0, e0.elapsed_time(e1)))
    n += 1
  t1 = time.perf_counter()
  stop = True; th.join()
  print(name)
  for t, ms in marks[:: max(1, len(marks) // 12)]:
    print(f"  t={t:5.2f}s  {ms:8.2f} ms  {flops / ms / 1e9:6.1f} TFLOP/s")
  for t, js in samples[:: max(1, len(samples) // 12)]:
    print(f"  t={t - t0:5.2f}s ", pick(js) if "error" not in js else js)
