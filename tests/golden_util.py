"""Helpers shared by the oracle and GPU parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a) -> str:
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_names(algo: str):
  with open(os.path.join(GOLDEN, "ref_cases.json")) as f:
    return [c["name"] for c in json.load(f)["cases"] if c["algo"] == algo]


def num(v, val=None):
  """JSON scalars: 'inf' / '-inf' / 'val' / '-val' placeholders."""
  if isinstance(v, str):
    table = {"inf": np.inf, "-inf": -np.inf}
    if val is not None:
      table.update({"val": val, "-val": -val})
    return table[v]
  return v


def gen_c2(seed=1234, shape=(4096, 4096)):
  return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def gen_c3(layer: int, shape=(4096, 11008)):
  return np.random.default_rng(1000 + layer).standard_normal(
      shape, dtype=np.float32) * np.float32(0.02)


def describe_model(model):
  """Payload-level description of a ModelT tree: every tensor's type / shape / buffer /
  quantization record and the SHA-256 of every buffer. Same structure as the `result` entries
  of tests/golden/ref_model_cases.json (written by tests/golden/gen/make_model_golden.py)."""
  import hashlib

  def digest(a):
    return hashlib.sha256(bytes(a)).hexdigest()

  out = dict(n_buffers=len(model.buffers), subgraphs=[], buffers=[])
  for b in model.buffers:
    if b.data is None:
      out["buffers"].append(None)
    else:
      raw = np.ravel(np.asarray(b.data)).view(np.uint8)
      out["buffers"].append(dict(nbytes=int(raw.nbytes), sha256=digest(raw)))
  for sg in model.subgraphs:
    tensors = []
    for t in sg.tensors:
      name = t.name.decode() if isinstance(t.name, (bytes, bytearray)) else str(t.name)
      rec = dict(name=name, type=int(t.type), shape=None if t.shape is None else [int(s) for s in t.shape],
                 buffer=int(t.buffer))
      q = t.quantization
      if q is not None:
        qr = dict(quantized_dimension=int(q.quantizedDimension), details_type=int(q.detailsType))
        for k, want in (("scale", np.float32), ("zeroPoint", np.int64), ("min", np.float32), ("max", np.float32)):
          a = getattr(q, k, None)
          if a is not None:
            a = np.asarray(a).astype(want)
            qr[k] = dict(n=int(a.size), sha256=digest(a.tobytes()), head=[float(x) for x in a.ravel()[:4]])
        if q.details is not None and int(q.detailsType) == 2:
          qr["blockwise"] = dict(scales=int(q.details.scales), zero_points=int(q.details.zeroPoints),
                                 block_size=int(q.details.blockSize))
        rec["quantization"] = qr
      tensors.append(rec)
    ops = [[int(model.operatorCodes[op.opcodeIndex].builtinCode), [int(i) for i in op.inputs],
            [int(i) for i in op.outputs]] for op in (sg.operators or [])]
    out["subgraphs"].append(dict(tensors=tensors, n_operators=len(sg.operators or []), operators=ops,
                                 inputs=[int(i) for i in sg.inputs], outputs=[int(i) for i in sg.outputs]))
  out["signatures"] = [[int(sig.subgraphIndex), [int(i.tensorIndex) for i in (sig.inputs or [])],
                        [int(i.tensorIndex) for i in (sig.outputs or [])]] for sig in (model.signatureDefs or [])]
  return out
