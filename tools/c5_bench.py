"""BASELINE config 5 on one GPU through the public API: GPTQ int4 on the seven FullyConnected
weights of Gemma-2B-shaped decoder layers (q, o: [2048,2048]; k, v: [256,2048]; gate, up:
[16384,2048]; down: [2048,16384]) with Hessians calibrated from synthetic activations
(`Quantizer.calibrate` over supplied tensors - the LiteRT interpreter's role).

  python tools/c5_bench.py [--layers 1] [--tokens 65536] [--samples 2]
One JSON line: seconds per layer split into calibrate / quantize / serialize.
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

D, DKV, DFF = 2048, 256, 16384


def build_layer_model(path, layers, rng):
  from mi355q import qtyping as q
  from mi355q import model_modifier
  model = q.ModelT(version=3, description=b"gemma-2b shaped decoder layers (synthetic)")
  model.buffers = [q.BufferT()]
  sg = q.SubGraphT(name=b"main", tensors=[], operators=[], inputs=[], outputs=[])

  def act(name, width):
    sg.tensors.append(q.TensorT(name=name.encode(), shape=[1, width], buffer=0))
    return len(sg.tensors) - 1

  def fc(name, x_id, rows, cols):
    w = (rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(0.02))
    model.buffers.append(q.BufferT(data=w.reshape(-1).view(np.uint8)))
    sg.tensors.append(q.TensorT(name=(name + "/w").encode(), shape=[rows, cols], buffer=len(model.buffers) - 1))
    wid = len(sg.tensors) - 1
    y = act(name + "/y", rows)
    sg.operators.append(q.OperatorT(inputs=[x_id, wid, -1], outputs=[y], opcodeIndex=0, builtinOptionsType=8,
                                    builtinOptions=q.FullyConnectedOptionsT()))
    return y
  inputs = {}
  for layer in range(layers):
    p = f"l{layer}"
    x_attn, x_o, x_mlp, x_down = (act(f"{p}/{n}", w) for n, w in
                                  (("attn_in", D), ("o_in", D), ("mlp_in", D), ("down_in", DFF)))
    sg.inputs += [x_attn, x_o, x_mlp, x_down]
    inputs[p] = dict(attn_in=(D, True), o_in=(D, True), mlp_in=(D, True), down_in=(DFF, True))
    # the calibration functions also want the ops' outputs (one token is enough here; the
    # reference collects min/max and a Hessian for them as well)
    for n, rows in (("q", D), ("k", DKV), ("v", DKV), ("o", D), ("gate", DFF), ("up", DFF), ("down", D)):
      inputs[p][f"{n}/y"] = (rows, False)
    outs = [fc(f"{p}/q", x_attn, D, D), fc(f"{p}/k", x_attn, DKV, D), fc(f"{p}/v", x_attn, DKV, D),
            fc(f"{p}/o", x_o, D, D), fc(f"{p}/gate", x_mlp, DFF, D), fc(f"{p}/up", x_mlp, DFF, D),
            fc(f"{p}/down", x_down, D, DFF)]
    sg.outputs += outs
  model.operatorCodes = [q.OperatorCodeT(builtinCode=int(q.BuiltinOperator.FULLY_CONNECTED), deprecatedBuiltinCode=9)]
  model.subgraphs = [sg]
  model.signatureDefs = [q.SignatureDefT(signatureKey=b"serving_default", subgraphIndex=0)]
  model_modifier.serialize_model(model, path)
  return inputs


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--layers", type=int, default=1)
  ap.add_argument("--tokens", type=int, default=65536)
  ap.add_argument("--samples", type=int, default=2, help="calibration samples (tokens are split over them)")
  ap.add_argument("--dir", default="/tmp")
  ap.add_argument("--resident", action="store_true",
                  help="samples are device tensors (activations produced on this GPU never visit the host)")
  ap.add_argument("--profile", action="store_true", help="cProfile of the quantize + write phase")
  ap.add_argument("--profile-calibrate", action="store_true", help="cProfile of the calibration phase")
  a = ap.parse_args()
  import __graft_entry__ as g
  g.build()
  import torch
  from mi355q import quantizer, recipe
  rng = np.random.default_rng(5000)
  src, dst = os.path.join(a.dir, "c5_in.tflite"), os.path.join(a.dir, "c5_out.tflite")
  inputs = build_layer_model(src, a.layers, rng)
  per = a.tokens // a.samples

  def samples():
    for s in range(a.samples):
      m = {}
      for p, widths in inputs.items():
        for name, (width, full) in widths.items():
          m[f"{p}/{name}"] = rng.standard_normal((1, per if full else 1, width), dtype=np.float32)
      yield m
  qz = quantizer.Quantizer(src, recipe.dynamic_wi4_afp32(algorithm_key="GPTQ"))
  data = list(samples())            # synthetic activations are generated outside the timed region
  if a.resident:
    data = [{k: torch.from_numpy(v).cuda() for k, v in m.items()} for m in data]
  torch.cuda.synchronize()
  if a.profile_calibrate:
    import cProfile
    import pstats
    prc = cProfile.Profile()
    prc.enable()
  t0 = time.perf_counter()
  qsvs = qz.calibrate({"serving_default": data})
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  if a.profile_calibrate:
    prc.disable()
    pstats.Stats(prc).sort_stats("tottime").print_stats(18)
  if a.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
  qz.quantize(calibration_result=qsvs, serialize_to_path=dst)
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  if a.profile:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
  weight_bytes = a.layers * 4 * (2 * D * D + 2 * DKV * D + 3 * DFF * D)
  print(json.dumps(dict(workload=f"C5: GPTQ int4, {a.layers} Gemma-2B-shaped layer(s), {a.tokens} calibration tokens"
                                 + (" (samples resident in HBM)" if a.resident else ""),
                        calibrate_s=round(t1 - t0, 3), quantize_and_write_s=round(t2 - t1, 3),
                        s_per_layer=round((t2 - t0) / a.layers, 3), weight_bytes=weight_bytes,
                        out_file=os.path.getsize(dst))))
  os.remove(src)
  os.remove(dst)


if __name__ == "__main__":
  main()
