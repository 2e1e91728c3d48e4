"""X^T X: the FP32-MFMA product against the bf16 three-way split (MI355Q_XTX_FP32_MFMA=1), error vs float64 and time.
usage: python tools/xtx_split_probe.py [d=2048] [tokens=16384]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd")); sys.path.insert(0, ROOT)
import __graft_entry__ as g; g.build()
from mi355q import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
torch.manual_seed(1)
x = torch.randn((n, d), device="cuda") * (1.0 + torch.rand((1, d), device="cuda") * 3.0) + 0.3
h = ops.gptq_xtx(x, 2.0 / n)
torch.cuda.synchronize()
ts = []
for _ in range(5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); h = ops.gptq_xtx(x, 2.0 / n); e1.record(); e1.synchronize()
  ts.append(e0.elapsed_time(e1))
if d <= 4096:
  ref = (x.double().T @ x.double()) * (2.0 / n)
  err = float((h - ref).abs().max() / ref.abs().max())
  sym = float((h - h.T).abs().max())
else:   # a 512-column strip
  ref = (x.double().T @ x[:, :512].double()) * (2.0 / n)
  err = float((h[:, :512] - ref).abs().max() / ref.abs().max())
  sym = float((h[:512, :512] - h[:512, :512].T).abs().max())
print(f"d={d} tokens={n} split={os.environ.get('MI355Q_XTX_FP32_MFMA', '0')}  ms {min(ts):.3f} (median {sorted(ts)[2]:.3f})  "
      f"TFLOP/s on the triangle {n * d * d / min(ts) / 1e9:.1f}  max err / max {err:.3e}  asym {sym:.1e}")
