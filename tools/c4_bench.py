"""BASELINE config 4 through the public API on one GPU: static_wi8_ai8 calibration of a chain
of FullyConnected ops with 32 activation tensors of [1, 256, 4096] FP32 (4 MiB each) per sample.

  python tools/c4_bench.py [--samples 32]
One JSON line: samples/s and activation GB/s, host arrays -> QSVs (PCIe included).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)


def build_model(tensors=32, width=4096, seq=256):
  """A chain of FullyConnected ops with `tensors` activation tensors of [1, seq, width]."""
  from mi355q import qtyping as q
  rng = np.random.default_rng(4)
  model = q.ModelT(version=3)
  model.buffers = [q.BufferT()]
  sg = q.SubGraphT(name=b"main", tensors=[], operators=[], inputs=[0], outputs=[tensors - 1])
  w = rng.standard_normal((width, width), dtype=np.float32) * np.float32(0.02)
  for i in range(tensors):
    sg.tensors.append(q.TensorT(name=f"act{i}".encode(), shape=[1, seq, width], buffer=0))
  for i in range(tensors - 1):
    model.buffers.append(q.BufferT(data=w.reshape(-1).view(np.uint8)))
    sg.tensors.append(q.TensorT(name=f"w{i}".encode(), shape=[width, width], buffer=len(model.buffers) - 1))
    sg.operators.append(q.OperatorT(inputs=[i, len(sg.tensors) - 1, -1], outputs=[i + 1], opcodeIndex=0,
                                    builtinOptionsType=8, builtinOptions=q.FullyConnectedOptionsT(keepNumDims=True)))
  model.operatorCodes = [q.OperatorCodeT(builtinCode=int(q.BuiltinOperator.FULLY_CONNECTED), deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0)]
  return model


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--samples", type=int, default=32)
  ap.add_argument("--tensors", type=int, default=32)
  ap.add_argument("--resident", action="store_true", help="samples are device tensors")
  ap.add_argument("--profile", action="store_true", help="cProfile the timed loop (slower)")
  a = ap.parse_args()
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import calibrator, qtyping as q, recipe, recipe_manager
  rng = np.random.default_rng(4)
  width, seq = 4096, 256
  model = build_model(a.tensors, width, seq)
  rm = recipe_manager.RecipeManager()
  rm.load_quantization_recipe(recipe.static_wi8_ai8())
  pool = [{f"act{i}": rng.standard_normal((1, seq, width), dtype=np.float32) * np.float32(1 + i / 8)
           for i in range(a.tensors)} for _ in range(4)]          # 4 distinct samples, reused
  if a.resident:
    pool = [{k: torch.from_numpy(v).cuda() for k, v in m.items()} for m in pool]
  cal = calibrator.Calibrator(model)
  cal.calibrate({"serving_default": pool[:2]}, rm)                # warm-up
  cal.reset_model_qsvs()
  torch.cuda.synchronize()
  if a.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
  t0 = time.perf_counter()
  cal.calibrate({"serving_default": (pool[i % 4] for i in range(a.samples))}, rm)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if a.profile:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
  nbytes = a.samples * a.tensors * seq * width * 4
  print(json.dumps(dict(workload=f"C4: static_wi8_ai8 calibration, {a.samples} samples x {a.tensors} x [1,{seq},{width}] f32"
                                 + (" (resident in HBM)" if a.resident else ""),
                        seconds=round(dt, 3), samples_per_s=round(a.samples / dt, 1),
                        activation_GBps=round(nbytes / dt / 1e9, 2), tensors_calibrated=len(cal.get_model_qsvs()))))


if __name__ == "__main__":
  main()
