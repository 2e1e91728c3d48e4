#!/usr/bin/env python3
"""GPTQ at d = 16384, full chain, on several INSTANCES: is "0 differing integers" with the exact Hessian product a property
of the kernels or of one lucky seed?   python tools/gptq_parity_instances.py [instances=3] [rows=32]

Per instance (seed): activations [64 x 512 tokens, 16384] (unit normal, sixteen loud channels, one dead), a weight
[rows, 16384] ~ N(0, 0.02^2), int4 channelwise. The oracle's own chain (NumPy sgemm Hessian, FP64 Cholesky, float32 strtri, float32
product: oracle/aeq_oracle.py) gives the reference integers, and -- with the Hessian summed as two half products -- the reference's
own re-ordering floor. The GPU chain runs twice: Hessian by the exact three-way bfloat16 split (the default) and by the two-way
float16 split (ops.hessian_product("fast")). One JSON line per instance. This is a CHECKER tool (it imports oracle/): not part
of the product path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)


def main():
  instances = int(sys.argv[1]) if len(sys.argv) > 1 else 3
  rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import ops
  from oracle import aeq_oracle as O
  d, seqs, tokens = 16384, 64, 512
  for inst in range(instances):
    t0 = time.time()
    gen = torch.Generator(device="cuda").manual_seed(7000 + inst)
    x = torch.randn((seqs, tokens, d), generator=gen, device="cuda")
    x[..., 5:21] *= 6.0
    x[..., 77] = 0.0
    w = torch.randn((rows, d), generator=gen, device="cuda") * 0.02
    scale = (torch.clamp(w.abs().amax(dim=1), min=1e-9) / 7.0).contiguous()
    got = {}
    for mode in ("exact", "fast"):
      with ops.hessian_product(mode):
        h = ops.gptq_xtx(x.reshape(-1, d), 2.0 / seqs)
      hinv, info = ops.gptq_hinv(h, 0.01)
      assert int(info.item()) == 0
      got[mode] = ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8).cpu().numpy()
      del h, hinv
    xh = x.cpu().numpy()
    del x
    hess = O.gptq_hessian(xh)
    x2 = xh.reshape(-1, d)
    half = x2.shape[0] // 2
    hess_b = (2.0 / np.array(seqs)) * (x2[:half].T.dot(x2[:half]) + x2[half:].T.dot(x2[half:]))
    del xh, x2
    hinv = O.gptq_hessian_inverse(hess, product="matmul")
    hinv_b = O.gptq_hessian_inverse(hess_b, product="matmul")
    del hess, hess_b
    wh, sh = w.cpu().numpy(), scale.cpu().numpy().reshape(-1, 1)
    zp = np.zeros((rows, 1), np.int8)
    ref = O.gptq_apply(wh, sh, zp, 4, True, None, "CHANNELWISE", hinv=hinv)
    ref_b = O.gptq_apply(wh, sh, zp, 4, True, None, "CHANNELWISE", hinv=hinv_b)
    n = ref.size
    rec = dict(instance=inst, seed=7000 + inst, shape=[rows, d], elements=int(n),
               gpu_exact_vs_oracle=int((got["exact"] != ref).sum()), gpu_fast_vs_oracle=int((got["fast"] != ref).sum()),
               gpu_exact_vs_reordered=int((got["exact"] != ref_b).sum()), gpu_fast_vs_reordered=int((got["fast"] != ref_b).sum()),
               oracle_vs_reordered_oracle=int((ref != ref_b).sum()),
               max_step=int(max(np.abs(got["exact"].astype(int) - ref).max(), np.abs(got["fast"].astype(int) - ref).max())),
               seconds=round(time.time() - t0, 1))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
  main()
