"""File in -> file out timing: a synthetic `.tflite` of N FullyConnected layers (FP32 weights,
external-buffer layout) through Quantizer(path).quantize(serialize_to_path=...).

usage: python tools/file_bench.py [--layers 8] [--rows 4096] [--cols 11008] [--recipe wi4b128|wi8]
Prints one JSON line with the end-to-end rate (FP32 weight bytes / wall time, PCIe and file
system included) and a breakdown. This is NOT bench.py's `value` (inputs resident in HBM).
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


def build_model(path, layers, rows, cols, same=()):
  """`same`: layer indices that get layer 0's weights (equal constants: the writer's buffer sharing)."""
  from mi355q import qtyping as q
  from mi355q import model_modifier
  rng = np.random.default_rng(0)
  model = q.ModelT(version=3, description=b"mi355q file bench")
  model.buffers = [q.BufferT()]
  sg = q.SubGraphT(name=b"main", tensors=[], operators=[])
  sg.tensors.append(q.TensorT(name=b"x0", shape=[1, cols], buffer=0))
  base = rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(0.02)
  prev = 0
  for i in range(layers):
    w = base if i == 0 or i in same else np.roll(base, i, axis=1) + np.float32(i * 1e-4)   # distinct content, cheap to make
    model.buffers.append(q.BufferT(data=w.reshape(-1).view(np.uint8)))
    sg.tensors.append(q.TensorT(name=f"w{i}".encode(), shape=[rows, cols], buffer=len(model.buffers) - 1))
    wid = len(sg.tensors) - 1
    sg.tensors.append(q.TensorT(name=f"y{i}".encode(), shape=[1, rows], buffer=0))
    yid = len(sg.tensors) - 1
    sg.operators.append(q.OperatorT(inputs=[prev, wid, -1], outputs=[yid], opcodeIndex=0, builtinOptionsType=8,
                                    builtinOptions=q.FullyConnectedOptionsT()))
    # next layer consumes a fresh input so shapes stay [1, cols]
    sg.tensors.append(q.TensorT(name=f"x{i + 1}".encode(), shape=[1, cols], buffer=0))
    prev = len(sg.tensors) - 1
  sg.inputs = [0]
  sg.outputs = [len(sg.tensors) - 2]
  model.operatorCodes = [q.OperatorCodeT(builtinCode=int(q.BuiltinOperator.FULLY_CONNECTED), deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  model_modifier.serialize_model(model, path)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--layers", type=int, default=8)
  ap.add_argument("--rows", type=int, default=4096)
  ap.add_argument("--cols", type=int, default=11008)
  ap.add_argument("--recipe", default="wi4b128")
  ap.add_argument("--dir", default="/tmp")
  ap.add_argument("--repeat", type=int, default=2)
  a = ap.parse_args()
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import quantizer, recipe
  src = os.path.join(a.dir, "file_bench_in.tflite")
  dst = os.path.join(a.dir, "file_bench_out.tflite")
  t0 = time.perf_counter()
  build_model(src, a.layers, a.rows, a.cols)
  t_build = time.perf_counter() - t0
  rcp = recipe.dynamic_wi4b128_afp32() if a.recipe == "wi4b128" else recipe.dynamic_wi8_afp32()
  weight_bytes = a.layers * a.rows * a.cols * 4
  best = None
  res = qz = None
  for _ in range(a.repeat):
    res = qz = None              # (the last result maps the output file: its pages go when the mapping does, not inside the timing)
    if os.path.exists(dst):
      os.remove(dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    qz = quantizer.Quantizer(src, rcp)
    t_open = time.perf_counter() - t0
    res = qz.quantize(serialize_to_path=dst)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    best = min(best, t_all) if best else t_all
    from mi355q import runtime as rt
    if rt.TIMELINE:      # MI355Q_TIMELINE=1
      print(json.dumps({"seconds": round(t_all, 4), "timeline": [(label, round((a - t0) * 1e3, 2)) for label, a, _, _ in rt.TIMELINE]}))
      del rt.TIMELINE[:]
  print(json.dumps(dict(workload=f"{a.layers} x FC {a.rows}x{a.cols} fp32 .tflite -> {a.recipe}",
                        weight_bytes=weight_bytes, in_file=os.path.getsize(src), out_file=os.path.getsize(dst),
                        seconds=round(best, 4), gbps=round(weight_bytes / best / 1e9, 2),
                        open_parse_s=round(t_open, 4), build_input_s=round(t_build, 2))))
  for p in (src, dst):
    os.remove(p)


if __name__ == "__main__":
  main()
