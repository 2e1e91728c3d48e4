"""Damped Hessian inverse: time and error against the exact FP64 inverse, with the single-precision
steps (triangular inverse merges, L^-T L^-1) on the bf16 split or in FP64 (MI355Q_HINV_FP64=1 in
the environment selects the latter for the whole process).
    python tools/hinv_accuracy.py [d ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def main():
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  dims = [int(v) for v in sys.argv[1:]] or [4096, 16384]
  for d in dims:
    gen = torch.Generator(device="cuda").manual_seed(5000)
    tokens = 4 * d if d <= 8192 else 65536
    h = None
    for k0 in range(0, tokens, 16384):
      x = torch.randn((min(16384, tokens - k0), d), generator=gen, device="cuda")
      x[:, 5:21] *= 6.0
      x[:, 77] = 0.0
      part = ops.gptq_xtx(x, 2.0 / 128)
      h = part if h is None else h + part
      del x, part
    for _ in range(2):
      hinv, info = ops.gptq_hinv(h, 0.01)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
      hinv, info = ops.gptq_hinv(h, 0.01)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dg = torch.diagonal(h).clone()
    dg = torch.where(dg == 0, torch.ones_like(dg), dg)
    damped = h.clone()
    damped.diagonal().copy_(dg + 0.01 * dg.mean())
    exact = torch.linalg.inv(damped)
    err = float((hinv.double() - exact).abs().max() / exact.abs().max())
    resid = hinv.double() @ damped
    resid.diagonal().sub_(1.0)
    print(json.dumps(dict(d=d, tokens=tokens, ms=round(ms, 3), info=int(info.item()), max_rel_error_vs_exact=err,
                          residual=float(resid.abs().max()), symmetric=bool(torch.equal(hinv, hinv.T)),
                          mode="fp64" if os.environ.get("MI355Q_HINV_FP64") else "bf16 split")), flush=True)
    del exact, resid, damped, hinv, h


if __name__ == "__main__":
  main()
