"""Launches each HBM-bound kernel of the path a few times on fresh data so that two rocprofv3 PMC
passes (FETCH_SIZE, WRITE_SIZE) can be held against the algorithmic bytes of SURVEY 8d.
    python tools/traffic_driver.py            (prints "<kernel substring> <algorithmic bytes per launch>" lines)"""
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
  gen = torch.Generator(device="cuda").manual_seed(3)
  out = []

  def rand(*shape, scale=0.02):
    return torch.randn(shape, generator=gen, device="cuda") * scale
  # C3: blockwise-128 int4, fused pack (6 layers per launch)
  xs = [rand(4096, 11008) for _ in range(6)]
  b3 = ops.RequantBatch(xs, block=128, bits=4, want_q=False, want_packed=True, want_scale_f16=True)
  for _ in range(3):
    b3.run()
  out.append(("requant_groups_kernel", 6 * (4096 * 11008 * 4 + 4096 * 11008 // 2 + 4096 * 11008 // 128 * 2 + 4096 * 11008 // 128 * 4)))
  del b3, xs
  # C4: activation min / max, 128 tensors of 4 MiB
  acts = [rand(1, 256, 4096, scale=1.0).reshape(-1) for _ in range(128)]
  amm = ops.ActMinMaxBatch(acts)
  for _ in range(3):
    amm.run()
  out.append(("act_minmax_kernel", 128 * 256 * 4096 * 4))
  del amm, acts
  # Hadamard rotations
  for rows, h, tag in ((4096, 4096, "fwht_tile_kernel<12"), (4096, 8192, "fwht_tile_kernel<13"), (2048, 16384, "fwht_tile_kernel<14")):
    w = rand(rows, h)
    for _ in range(3):
      ops.hadamard_rotate(w, h)
    out.append((tag, 2 * rows * h * 4))
    del w
  # OCTAV: rows kernel (4096 x 4096 and 2048 x 16384), groups kernel (blocks of 128)
  w = rand(4096, 4096)
  for _ in range(3):
    ops.octav_clip(w.view(-1), 4096, 4096, 4, 10, 3.0, True, True)
  out.append(("octav_rows_kernel<1, 256>", 4096 * 4096 * 4))
  for _ in range(3):
    ops.octav_clip(w.view(-1), 4096 * 32, 128, 4, 10, 3.0, True, True)
  out.append(("octav_unit_lanes_kernel", 4096 * 4096 * 4))
  w2 = rand(2048, 16384)
  for _ in range(3):
    ops.octav_clip(w2.view(-1), 2048, 16384, 4, 10, 3.0, True, True)
  out.append(("octav_rows_kernel<1, 1024>", 2048 * 16384 * 4))
  # ... and the opt-in one-read kernel (ops.octav_mode("fast")) on the same two shapes
  with ops.octav_mode("fast"):
    for _ in range(3):
      ops.octav_clip(w.view(-1), 4096, 4096, 4, 10, 3.0, True, True)
    out.append(("octav_fast_kernel<64, 16>", 4096 * 4096 * 4))
    for _ in range(3):
      ops.octav_clip(w2.view(-1), 2048, 16384, 4, 10, 3.0, True, True)
    out.append(("octav_fast_kernel<1024, 16>", 2048 * 16384 * 4))
  del w, w2
  # GPTQ apply, 2048 x 2048 int4: the column-serial kernel reads W once and writes int8 once per group of 4 blocks
  d = 2048
  x = rand(8192, d, scale=1.0)
  hinv, _ = ops.gptq_hinv(ops.gptq_xtx(x, 2.0 / 16), 0.01)
  w = rand(2048, d)
  sc = (w.abs().amax(dim=1) / 7).contiguous()
  for _ in range(3):
    ops.gptq_apply(w, hinv, sc, None, 1, 0, 4, False, False, 8)
  out.append(("gptq_rows_kernel", (2048 * 256 * 4 * 2 + 2048 * 256) * 1))     # per launch: one group of 256 columns read + written back (float32) + int8
  # ---- round 6: MSE scale (one read), OSCAR clip search by prefix (one read of w + s), the non-fused kernels
  import numpy as np
  w = rand(4096, 4096)
  for _ in range(3):
    ops.mse_scale(w.view(-1), 4096, 4096, 0.37755)
  out.append(("mse_scale_balanced_kernel", 4096 * 4096 * 4))
  rng = np.random.default_rng(1)
  s_ = ops._f64_dev(np.exp(rng.normal(size=4096) * 0.2))   # pylint: disable=protected-access
  m_ = ops._f64_dev(np.exp(rng.normal(size=4096) * 1.5))   # pylint: disable=protected-access
  u_, n_ = ops._f64_dev(np.full(1, 0.01)), ops._f64_dev(np.full(1, 0.005))   # pylint: disable=protected-access
  for _ in range(3):
    ops.oscar_clip_bounds(w, s_, m_, 4096, u_, n_, 7, False, True, True)
  out.append(("clip_prefix_kernel<64, 1>", 4096 * 4096 * 4))
  for _ in range(3):
    ops.minmax(w, 1, 4096, 4096)
  out.append(("minmax_runs_kernel", 4096 * 4096 * 4))
  mn, mx = ops.minmax(w, 1, 4096, 4096)
  scale = ((mx - mn) / 255.0).contiguous()
  zp = torch.round(-128 - mn / scale).to(torch.int32)
  for _ in range(3):
    q8 = ops.quantize(w, 1, 4096, 4096, scale, zp, 8, False)
  out.append(("quantize_rows_vec4_kernel", 4096 * 4096 * 5))
  for _ in range(3):
    ops.dequantize(q8, 1, 4096, 4096, scale, zp, 8)
  out.append(("dequantize_rows_vec4_kernel", 4096 * 4096 * 5))
  torch.cuda.synchronize()
  for k, v in out:
    print(f"ALG {k}\t{v}")


if __name__ == "__main__":
  main()
