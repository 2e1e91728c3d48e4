#!/usr/bin/env python3
"""Quantize one model file on N GPUs of a node, one process per GPU (RCCL over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/quantize_sharded.py model.tflite recipe.json model_q.tflite [calibration.npz]

The ops' weight buffers are spread over the ranks by bytes (no collective on the data path), rank 0
gathers the results and writes the file. For recipes that need calibration, `calibration.npz`
holds the samples as arrays named "<sample index>/<tensor name>" (what an interpreter run of the
float model yields); the samples are sharded over the ranks and the statistics replayed in order,
so the file equals the single-GPU one byte for byte. Works unchanged with N = 1.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ai-edge-quantizer_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main() -> int:
  if len(sys.argv) < 4:
    print(__doc__)
    return 2
  model_path, recipe_path, out_path = sys.argv[1:4]
  import __graft_entry__ as g
  g.build()
  from mi355q import distributed as D
  rank, world = D.init()
  with open(recipe_path, "r", encoding="utf-8") as f:
    recipe = json.load(f)
  qsvs = None
  if len(sys.argv) > 4:
    z = np.load(sys.argv[4])
    samples: dict[int, dict] = {}
    for key in z.files:
      idx, name = key.split("/", 1)
      samples.setdefault(int(idx), {})[name] = z[key]
    qsvs = D.calibrate_sharded(model_path, recipe, {None: [samples[i] for i in sorted(samples)]})
  result = D.quantize_model_sharded(model_path, recipe, calibration_result=qsvs, serialize_to_path=out_path)
  if rank == 0:
    print(json.dumps(dict(ranks=world, out=out_path, bytes=len(result))))
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
  return 0


if __name__ == "__main__":
  sys.exit(main())
