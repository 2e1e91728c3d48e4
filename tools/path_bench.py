#!/usr/bin/env python3
"""Timings of the non-headline rows of the hot path on one MI355X (HIP events on
the launch stream, HBM-resident inputs): activation min/max (C4 shape), OCTAV,
Hadamard (+OCTAV), and the GPTQ pieces at Gemma-2B shapes (C5).

    python tools/path_bench.py [--big]      # --big adds d=16384 (1 GiB Hessians)
Prints one JSON object per line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

g.build()
from mi355q import ops  # noqa: E402


def timed(fn, iters, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters):
    fn()
  b.record()
  b.synchronize()
  return a.elapsed_time(b) / iters


def emit(**kw):
  print(json.dumps(kw), flush=True)


def prewarm():
  x = torch.randn((4096, 4096), device="cuda")
  t = time.perf_counter()
  while time.perf_counter() - t < 0.3:
    ops.requant_sym(x, 0, 8)
  torch.cuda.synchronize()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--big", action="store_true")
  args = ap.parse_args()
  gen = torch.Generator(device="cuda").manual_seed(0)
  prewarm()

  # C4: one calibration sample = 32 activation tensors of [1,256,4096] (4 MiB each)
  acts = [torch.randn((256 * 4096,), generator=gen, device="cuda") * (1 + i / 8) for i in range(32)]
  acts += [torch.randn((256 * 4096,), generator=gen, device="cuda") for _ in range(96)]  # 512 MiB pool
  amm = ops.ActMinMaxBatch(acts)
  ms = timed(amm.run, 50, warm=10)
  nbytes = sum(a.numel() for a in acts) * 4
  emit(op="act_minmax", tensors=len(acts), bytes=nbytes, ms=round(ms, 4),
       GBps=round(nbytes / ms / 1e6, 1), hbm_frac=round(nbytes / ms / 1e6 / 8000, 4))
  del acts

  w = torch.randn((4096, 4096), generator=gen, device="cuda")
  flat = w.view(-1)
  for bits in (4, 8):
    ms = timed(lambda: ops.octav_clip(flat, 4096, 4096, bits), 10)
    emit(op="octav_clip", shape=[4096, 4096], bits=bits, granularity="CHANNELWISE", ms=round(ms, 4))
  ms = timed(lambda: ops.octav_clip(flat, 4096 * 32, 128, 4), 5)
  emit(op="octav_clip", shape=[4096, 4096], bits=4, granularity="BLOCKWISE_128", ms=round(ms, 4))
  ms = timed(lambda: ops.hadamard_rotate(flat, 4096), 20)
  emit(op="hadamard_rotate", shape=[4096, 4096], h=4096, ms=round(ms, 4),
       GBps_rw=round(2 * w.numel() * 4 / ms / 1e6, 1))

  def had_octav():
    r = ops.hadamard_rotate(flat, 4096)
    clip, _ = ops.octav_clip(r, 4096, 4096, 4)
    return ops.requant_sym(r.view(4096, 4096), 0, 4, clip=clip, want_q=False, want_packed=True)
  ms = timed(had_octav, 10)
  emit(op="hadamard+octav+requant int4 (reference CPU: 3.51 s)", shape=[4096, 4096], ms=round(ms, 4))

  # PCIe-inclusive rates (host NumPy in -> host NumPy out); never the headline value
  import numpy as np
  from mi355q import qtyping
  from mi355q.algorithms.uniform_quantize import naive_min_max_quantize as mm
  host = [np.random.default_rng(i).standard_normal((4096, 4096), dtype=np.float32) for i in range(16)]
  cfg = qtyping.TensorQuantizationConfig(num_bits=8, granularity=qtyping.QuantGranularity.CHANNELWISE)
  info = qtyping.OpInfo(op=qtyping.OperatorT(), op_name=qtyping.TFLOperationName.FULLY_CONNECTED,
                        subgraph_op_index=0,
                        op_quant_config=qtyping.OpQuantizationConfig(weight_tensor_config=cfg))
  mm.get_tensor_quant_params(info, cfg, host[0])
  t0 = time.perf_counter()
  for hw in host:
    np.asarray(mm.get_tensor_quant_params(info, cfg, hw).quantized_data)   # host array out (D2H included)
  dt = time.perf_counter() - t0
  emit(op="get_tensor_quant_params host->host, one tensor at a time", tensors=len(host),
       GBps=round(len(host) * 64 * 2**20 / dt / 1e9, 2), ms_per_tensor=round(dt / len(host) * 1e3, 2))
  del host

  # C5 (Gemma-2B shapes): d = 2048, calibration 128 x 512 tokens
  dims = [2048] + ([16384] if args.big else [])
  for d in dims:
    ntok = 65536 if d == 2048 else 16384
    x = torch.randn((ntok, d), generator=gen, device="cuda")
    ms = timed(lambda: ops.gptq_xtx(x, 2.0 / 128), 5 if d == 2048 else 2, warm=1)
    flops = 2.0 * ntok * d * d            # of the full product; only the tiles on / below the
    nt = d // 128                          # diagonal are computed (symmetric result, mirrored)
    done = flops * (nt + 1) / (2 * nt)
    emit(op="gptq_hessian xtx (bf16 MFMA x 6 on the exact three-way split, lower triangle + mirror)", d=d, tokens=ntok,
         ms=round(ms, 3), float32_product_TFLOPs_executed=round(done / ms / 1e9, 2),
         bf16_mfma_TFLOPs_executed=round(6 * done / ms / 1e9, 1), mfma_bf16_peak=2500.0, mfma_f32_peak=157.3)
    h = ops.gptq_xtx(x, 2.0 / 128)
    del x
    ms = timed(lambda: ops.gptq_hinv(h), 3 if d == 2048 else 1, warm=1)
    emit(op="gptq_hinv (f64 cholesky+trtri+product)", d=d, ms=round(ms, 3),
         TFLOPs_f64=round((d ** 3) * (1 / 3 + 1 / 3 + 1 / 3) / ms / 1e9, 2))
    hinv, info = ops.gptq_hinv(h)
    assert int(info.item()) == 0
    rows = 2048
    wq = torch.randn((rows, d), generator=gen, device="cuda") * 0.02
    scale = (wq.abs().amax(dim=1) / 7).contiguous()
    ms = timed(lambda: ops.gptq_apply(wq, hinv, scale, None, 1, 0, 4, False, False, 8),
               3 if d == 2048 else 1, warm=1)
    emit(op="gptq_apply int4 channelwise", rows=rows, d=d, ms=round(ms, 3),
         TFLOPs=round(1.0 * rows * d * d / ms / 1e9, 2))
    del h, hinv, wq


if __name__ == "__main__":
  main()
