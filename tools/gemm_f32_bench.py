"""FP32 MFMA GEMM timing (HIP events) for the Hessian's shapes: X^T X as a full product and as its
lower triangle.   python tools/gemm_f32_bench.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build()
from mi355q import _ffi
lib = _ffi.lib()
f = lib.mi355q_gemm_f32
st = torch.cuda.current_stream().cuda_stream
for (d, n) in ((16384, 16384), (8192, 16384), (2048, 65536)):
  X = torch.randn((n, d), dtype=torch.float32, device="cuda")
  C = torch.zeros((d, d), dtype=torch.float32, device="cuda")
  for lower in (0, 1):
    ts = []
    for it in range(4):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      # C(i,j) = sum_k X[k][i] X[k][j]
      rc = f(X.data_ptr(), 1, d, X.data_ptr(), d, 1, C.data_ptr(), d, 1, d, d, n, 1.0, 0.0, lower, st)
      e1.record(); e1.synchronize()
      assert rc == 0, rc
      ts.append(e0.elapsed_time(e1))
    ms = sorted(ts[1:])[1]
    flops = 2.0 * d * d * n * (0.5 if lower else 1.0)
    print(json.dumps(dict(d=d, tokens=n, lower=lower, ms=round(ms, 3), tflops_executed=round(flops / ms / 1e9, 1))))
  del X, C
