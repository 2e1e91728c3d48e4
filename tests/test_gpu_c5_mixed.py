"""BASELINE config 5 AS bench.py TIMES IT (`c5_mixed`): the `mixed` recipe -- GPTQ int4 on q / k / v / o / gate / up,
decomposed Hadamard rotation + OCTAV int4 on down -- through ONE
`quantize_litertlm(container, recipe, out, calibration_data=...)` call on a 2-layer Gemma-2B-shaped container (the
full shapes: q, o [2048, 2048]; k, v [256, 2048]; gate, up [16384, 2048]; down [2048, 16384]).

  (a) every GPTQ projection of the written container against the oracle's OWN chain (Hessian from the calibration
      tokens by sgemm, FP64 Cholesky, strtri, the reference's einsum product, the column loop) on row slices, under
      the default path's fixed bound -- ref gptq.py:100-128, 131-216, 219-300;
  (b) down against the oracle's rotate -> OCTAV -> quantize (T2: scales 1e-6, integers one step on <= 1e-5), and the
      three ops the transformation inserts in front of every down FULLY_CONNECTED -- ref hadamard_rotation.py:137-203,
      transformations/insert_decomposed_hadamard_rotation.py:82-265;
  (c) the container's bytes equal the two-step route's (calibrate_litertlm -> quantize_litertlm(calibration_results=))
      and the route with the overlap machinery off (MI355Q_NO_PREFETCH / MI355Q_NO_OUTPUT_PREPARE);
  (d) ... and do not depend on how many calibration samples share a launch (1 = the per-sample walk, 8, 64).
The overlapped call is what a user gets and what the bench reports; (c) and (d) tie it to the routes the other tests
check piece by piece. Ref: aeq.py:61-181 (the container loop)."""
import hashlib
import os
import sys
import tempfile
import warnings

import numpy as np
import pytest

import parity_rates
from oracle import aeq_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

LAYERS, SEQUENCES, TOKENS = 2, 64, 512


def _sha(path: str) -> str:
  h = hashlib.sha256()
  with open(path, "rb") as f:
    for chunk in iter(lambda: f.read(1 << 24), b""):
      h.update(chunk)
  return h.hexdigest()


def _unpack_int4(packed: np.ndarray, n: int) -> np.ndarray:
  b = np.asarray(packed, dtype=np.uint8)
  out = np.empty(b.size * 2, np.int8)
  out[0::2], out[1::2] = (b & 0xF).astype(np.int8), (b >> 4).astype(np.int8)
  return np.where(out > 7, out - 16, out).astype(np.int8)[:n]


@pytest.fixture(scope="module")
def c5m():
  """The bench's call, once: container in, container out."""
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import c5_model as C
  from mi355q import calibrator, ops
  from mi355q.utils import litertlm_utils
  tmp = tempfile.mkdtemp(prefix="mi355q_c5m_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
  src, dst = os.path.join(tmp, "in.litertlm"), os.path.join(tmp, "out.litertlm")
  model = C.build_model(LAYERS)
  weights = {}
  for t in model.subgraphs[0].tensors:
    name = t.name.decode()
    if name.endswith("/w"):
      weights[name] = np.asarray(model.buffers[t.buffer].data).view(np.float32).reshape(t.shape).copy()
  n_float_ops = len(model.subgraphs[0].operators)
  C.write_litertlm(model, src)
  del model
  samples = C.calibration_set(torch, LAYERS, SEQUENCES, TOKENS)
  rcp = C.recipe("mixed")
  data = {0: {"serving_default": samples}}
  launches = []
  real = calibrator.Calibrator._gather_block   # pylint: disable=protected-access

  def counted(self, slots, samples_, limit, *a, **kw):
    out = real(self, slots, samples_, limit, *a, **kw)
    launches.append((limit, out[1]))
    return out
  calibrator.Calibrator._gather_block = counted   # pylint: disable=protected-access
  try:
    stats = {}
    n = litertlm_utils.quantize_litertlm(src, rcp, dst, calibration_data=data, stats=stats)     # bench.py's c5_mixed call
  finally:
    calibrator.Calibrator._gather_block = real   # pylint: disable=protected-access
  torch.cuda.synchronize()
  qmodel = litertlm_utils.LiteRTLMFile(dst).read_model(0)
  yield dict(C=C, torch=torch, ops=ops, lm=litertlm_utils, cal=calibrator, weights=weights, samples=samples, data=data,
             rcp=rcp, src=src, dst=dst, tmp=tmp, out_bytes=n, qmodel=qmodel, sha=_sha(dst), launches=launches,
             n_float_ops=n_float_ops, stats=stats)
  for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
  os.rmdir(tmp)


def _tensor(qmodel, name):
  return next(t for t in qmodel.subgraphs[0].tensors if t.name.decode() == name)


def _quantized(qmodel, name):
  t = _tensor(qmodel, name)
  rows, d = (int(v) for v in t.shape)
  q = _unpack_int4(np.asarray(qmodel.buffers[t.buffer].data), rows * d).reshape(rows, d)
  return q, np.asarray(t.quantization.scale, dtype=np.float32), t


def _tokens(c5m, layer, src):
  x = c5m["torch"].stack([s[f"l{layer}/{src}"] for s in c5m["samples"]])     # [sequences, 1, tokens, d]
  return x.reshape(SEQUENCES, TOKENS, -1).cpu().numpy()


def test_the_call_is_the_overlapped_one_and_wrote_an_int4_container(c5m):
  from mi355q import qtyping
  per_layer = sum(r * c for _, r, c, _ in c5m["C"].projections())
  assert c5m["out_bytes"] == os.path.getsize(c5m["dst"])
  assert c5m["out_bytes"] > LAYERS * per_layer // 2
  # + one float32 Hadamard matrix of order 2048 (shared by the rotations) and metadata
  assert c5m["out_bytes"] < LAYERS * per_layer // 2 + 2048 * 2048 * 4 + (8 << 20)
  # calibration took whole blocks (64 samples: one launch of 64), not the per-sample walk
  assert c5m["launches"] and c5m["launches"][0] == (64, 64), c5m["launches"]
  # the output file was laid out before the model was quantized (the machinery (c) switches off)
  assert c5m["stats"].get("expected_section_bytes", 0) >= c5m["stats"]["section_bytes"] > 0
  for t in c5m["qmodel"].subgraphs[0].tensors:
    if t.name.decode().endswith("/w"):
      assert t.type == qtyping.TensorType.INT4, t.name


@pytest.mark.parametrize("layer,src", [(0, "attn_in"), (0, "o_in"), (0, "mlp_in"), (1, "attn_in"), (1, "mlp_in")])
def test_gptq_projections_against_the_oracles_own_chain(c5m, layer, src):
  """(a) nothing of the GPU's enters the reference side but the float weights."""
  x = _tokens(c5m, layer, src)
  hess = O.gptq_hessian(x)
  hinv = O.gptq_hessian_inverse(hess)                      # FP64 Cholesky, strtri, the reference's einsum
  x2 = x.reshape(-1, x.shape[-1])
  half = x2.shape[0] // 2
  # the reference's own reproducibility, recorded beside the observation: the Hessian's float32 sums in another order
  hinv_b = O.gptq_hessian_inverse((2.0 / np.array(x.shape[0])) * (x2[:half].T.dot(x2[:half]) + x2[half:].T.dot(x2[half:])))
  del x, x2, hess
  for name, rows, d, reads in c5m["C"].projections():
    if reads != src:
      continue
    w = c5m["weights"][f"l{layer}/{name}/w"]
    q, scale, _ = _quantized(c5m["qmodel"], f"l{layer}/{name}/w")
    ref_scale = O.min_max_quant_params(w, 4, True, "CHANNELWISE")["scale"]
    assert np.array_equal(scale, ref_scale.reshape(-1)), name            # a1 + a2: bit-exact scales
    sel = np.r_[0:32, rows - 32:rows]
    zp = np.zeros((len(sel), 1), np.int8)
    ref = O.gptq_apply(w[sel], ref_scale[sel], zp, 4, True, None, "CHANNELWISE", hinv=hinv)
    ref_b = O.gptq_apply(w[sel], ref_scale[sel], zp, 4, True, None, "CHANNELWISE", hinv=hinv_b)
    parity_rates.check_default_path(
        f"C5 MIXED one call: l{layer}/{name} [{rows},{d}] int4, 64 rows vs oracle FULL CHAIN (einsum)", q[sel], ref, ref_b)
    assert q.min() >= -8 and q.max() <= 7 and (q != 0).mean() > 0.5


@pytest.mark.parametrize("layer", range(LAYERS))
def test_down_rows_against_the_oracles_rotation_and_octav(c5m, layer):
  """(b) down [2048, 16384], max_hadamard_size 2048: eight rotations of order 2048 per row, OCTAV clip, int4."""
  w = c5m["weights"][f"l{layer}/down/w"]
  q, scale, t = _quantized(c5m["qmodel"], f"l{layer}/down/w")
  sel = np.r_[0:32, 1000:1016, 2048 - 16:2048]
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.hadamard_quant_params(w[sel], 4, "CHANNELWISE", max_size=2048)
  assert ref["hadamard_size"] == 2048
  np.testing.assert_allclose(scale[sel], ref["scale"].reshape(-1), rtol=1e-6)
  parity_rates.check_rel(f"C5 MIXED one call: l{layer}/down scales, 64 rows vs oracle rotate + OCTAV", scale[sel],
                         ref["scale"].reshape(-1), 1e-6)
  parity_rates.check(f"C5 MIXED one call: l{layer}/down [2048,16384] hadamard(2048)+octav int4, 64 rows vs oracle",
                     q[sel], ref["quantized_data"], parity_rates.T2)
  assert t.quantization.quantizedDimension == 0 and not np.any(np.asarray(t.quantization.zeroPoint))


def test_every_down_reads_a_rotated_input(c5m):
  """(b) x -> RESHAPE(-1, 2048) -> FULLY_CONNECTED(H / sqrt(2048)) -> RESHAPE(x.shape) -> down."""
  from mi355q import qtyping
  sg, model = c5m["qmodel"].subgraphs[0], c5m["qmodel"]
  code = lambda op: model.operatorCodes[op.opcodeIndex].builtinCode
  name = lambda tid: sg.tensors[tid].name.decode()
  assert len(sg.operators) == c5m["n_float_ops"] + 3 * LAYERS
  matrices = set()
  for layer in range(LAYERS):
    w_id = next(i for i, t in enumerate(sg.tensors) if t.name.decode() == f"l{layer}/down/w")
    at = next(i for i, op in enumerate(sg.operators) if len(op.inputs) > 1 and op.inputs[1] == w_id)
    pre, fc, post, down = sg.operators[at - 3:at + 1]
    assert [code(o) for o in (pre, fc, post, down)] == [qtyping.BuiltinOperator.RESHAPE, qtyping.BuiltinOperator.FULLY_CONNECTED,
                                                        qtyping.BuiltinOperator.RESHAPE, qtyping.BuiltinOperator.FULLY_CONNECTED]
    assert name(pre.inputs[0]) == f"l{layer}/down_in"
    assert fc.inputs[0] == pre.outputs[0] and post.inputs[0] == fc.outputs[0] and down.inputs[0] == post.outputs[0]
    assert list(sg.tensors[pre.outputs[0]].shape) == [16384 // 2048, 2048]
    assert list(sg.tensors[post.outputs[0]].shape) == [1, 16384]
    mt = sg.tensors[fc.inputs[1]]
    assert list(mt.shape) == [2048, 2048] and mt.type == qtyping.TensorType.FLOAT32
    matrices.add(mt.buffer)
    h = np.asarray(model.buffers[mt.buffer].data).view(np.float32).reshape(2048, 2048)
    assert np.array_equal(h, O.hadamard_matrix(2048).astype(np.float32))
  assert len(matrices) == 1                   # one constant, shared by the rotations (allow_tensor_sharing)
  # the GPTQ projections' inputs are untouched
  for layer in range(LAYERS):
    for proj, _, _, src in c5m["C"].projections():
      if proj == "down":
        continue
      w_id = next(i for i, t in enumerate(sg.tensors) if t.name.decode() == f"l{layer}/{proj}/w")
      op = next(op for op in sg.operators if len(op.inputs) > 1 and op.inputs[1] == w_id)
      assert name(op.inputs[0]) == f"l{layer}/{src}"


def test_two_step_route_writes_the_same_bytes(c5m):
  """(c) calibrate_litertlm -> quantize_litertlm(calibration_results=...): what tests/test_gpu_c5_model.py checks."""
  lm = c5m["lm"]
  out = os.path.join(c5m["tmp"], "two_step.litertlm")
  qsvs = lm.calibrate_litertlm(c5m["src"], c5m["rcp"], c5m["data"])
  for name, qsv in qsvs[0].items():            # a Hessian exactly where a GPTQ op reads one: not for down_in, not for outputs
    want = name.endswith(("attn_in", "o_in", "mlp_in"))
    assert ("hessian" in qsv) == want, name
  n = lm.quantize_litertlm(c5m["src"], c5m["rcp"], out, calibration_results=qsvs)
  c5m["torch"].cuda.synchronize()
  assert n == c5m["out_bytes"] == os.path.getsize(out)
  assert _sha(out) == c5m["sha"]
  os.remove(out)


def test_overlap_machinery_off_writes_the_same_bytes(c5m, monkeypatch):
  """(c) no upload thread, no output file laid out under the calibration, payloads and scales read on the spot."""
  lm, rt = c5m["lm"], sys.modules["mi355q.runtime"]
  out = os.path.join(c5m["tmp"], "plain.litertlm")
  monkeypatch.setenv("MI355Q_NO_PREFETCH", "1")
  monkeypatch.setenv("MI355Q_NO_OUTPUT_PREPARE", "1")
  monkeypatch.setattr(rt, "late_vector", lambda values, dtype: np.ravel(values).astype(dtype, copy=False))
  monkeypatch.setattr(rt, "late_constants_allowed", lambda: False)
  monkeypatch.setattr(rt, "download_into_file", lambda t, dst, *gate: False)
  stats = {}
  n = lm.quantize_litertlm(c5m["src"], c5m["rcp"], out, calibration_data=c5m["data"], stats=stats)
  c5m["torch"].cuda.synchronize()
  assert "expected_section_bytes" not in stats
  assert n == c5m["out_bytes"] == os.path.getsize(out)
  assert _sha(out) == c5m["sha"]
  os.remove(out)


@pytest.mark.parametrize("k", [1, 8])
def test_samples_per_launch_does_not_change_the_bytes(c5m, k, monkeypatch):
  """(d) the fixture ran 64 per launch; 1 is the per-sample walk (ref calibrator.py:312-331), 8 closes blocks inside
  Hessian products."""
  cal = c5m["cal"]
  monkeypatch.setattr(cal.Calibrator, "BLOCK_SAMPLES", k)
  launches = []
  real = cal.Calibrator._gather_block   # pylint: disable=protected-access

  def counted(self, slots, samples_, limit, *a, **kw):
    out = real(self, slots, samples_, limit, *a, **kw)
    launches.append((limit, out[1]))
    return out
  monkeypatch.setattr(cal.Calibrator, "_gather_block", counted)
  out = os.path.join(c5m["tmp"], f"k{k}.litertlm")
  n = c5m["lm"].quantize_litertlm(c5m["src"], c5m["rcp"], out, calibration_data=c5m["data"])
  c5m["torch"].cuda.synchronize()
  assert launches == ([] if k == 1 else [(8, 8)] * (SEQUENCES // 8)), launches
  assert n == c5m["out_bytes"]
  assert _sha(out) == c5m["sha"]
  os.remove(out)


def _worker_c5_mixed_two_ranks(rank, world, port, out, src="", dst=""):
  """One rank of the bench's C5 call over RCCL: samples sharded, every Hessian's float32 product reduced to the rank that
  owns the ops reading it (mi355q_reduce_product_f32 on the communication stream), ops sharded, rank 0 writes."""
  os.environ["MI355Q_TEST_DIST_BACKEND"] = "nccl"
  import test_gpu_distributed as T
  dist = T._setup(rank, world, port)   # pylint: disable=protected-access
  import torch
  import c5_model as C
  from mi355q import distributed as D
  from mi355q.utils import litertlm_utils
  samples = C.calibration_set(torch, LAYERS, SEQUENCES, TOKENS)          # the fixture's samples: every rank walks ITS shard of them
  del D.ISSUED[:]
  n = litertlm_utils.quantize_litertlm(src, C.recipe("mixed"), dst, calibration_data={0: {"serving_default": samples}})
  torch.cuda.synchronize()
  out.put((rank, n, T._rccl_ranks(), list(D.ISSUED), len(D.sample_shard(SEQUENCES, rank, world))))   # pylint: disable=protected-access
  dist.barrier()
  D.destroy_rccl_comms()
  dist.destroy_process_group()


def test_two_ranks_over_rccl_write_the_container_of_one(c5m):
  """SURVEY 8e at the headline's shapes: the SAME call on two ranks over RCCL (one rank per GPU where two are visible, else two
  "hosts" on cuda:0). Everything that does not depend on a Hessian -- scales, zero points, the rotated down rows, the graph -- is
  byte for byte the single-process container's; a GPTQ projection's integers may differ only where the float32 product summed as
  two partial products instead of one tips a rounding: under the default path's fixed bound, one step."""
  import functools
  from test_distributed_gloo import _run
  dst = os.path.join(c5m["tmp"], "two_ranks.litertlm")
  worker = functools.partial(_worker_c5_mixed_two_ranks, src=c5m["src"], dst=dst)
  results = _run(worker, world=2, timeout=900)
  (_, n0, comm0, issued0, share0), (_, n1, comm1, issued1, share1) = results
  assert comm0 == (2, 0) and comm1 == (2, 1)
  assert n0 == c5m["out_bytes"] == os.path.getsize(dst) and n1 is None
  assert share0 + share1 == SEQUENCES and share0 > 0 and share1 > 0
  # every rank issued the same product reduces in the same order (RCCL matches collectives by order of issue): one per Hessian
  # (attn_in, o_in, mlp_in of every layer), each to ONE owner
  assert issued0 == issued1 and len(issued0) == 3 * LAYERS
  assert sorted(name for name, _ in issued0) == sorted(f"l{l}/{s}" for l in range(LAYERS) for s in ("attn_in", "o_in", "mlp_in"))
  assert all(root in (0, 1) for _, root in issued0)
  one, two = c5m["qmodel"], c5m["lm"].LiteRTLMFile(dst).read_model(0)
  sg1, sg2 = one.subgraphs[0], two.subgraphs[0]
  assert len(sg1.tensors) == len(sg2.tensors) and len(sg1.operators) == len(sg2.operators)
  ints_one, ints_two = [], []
  for t1, t2 in zip(sg1.tensors, sg2.tensors):
    assert t1.name == t2.name and list(t1.shape) == list(t2.shape) and t1.type == t2.type
    q1, q2 = t1.quantization, t2.quantization
    assert (q1 is None) == (q2 is None)
    if q1 is not None:
      assert np.array_equal(np.asarray(q1.scale), np.asarray(q2.scale)) and np.array_equal(np.asarray(q1.zeroPoint), np.asarray(q2.zeroPoint))
    b1, b2 = np.asarray(one.buffers[t1.buffer].data), np.asarray(two.buffers[t2.buffer].data)
    name = t1.name.decode()
    gptq_weight = name.endswith("/w") and "/down/" not in name
    if not gptq_weight:
      assert np.array_equal(b1, b2), name
      continue
    count = int(np.prod(t1.shape))
    ints_one.append(_unpack_int4(b1, count))
    ints_two.append(_unpack_int4(b2, count))
  assert len(ints_one) == 6 * LAYERS
  parity_rates.check("C5 MIXED one call, two ranks over RCCL vs one process: GPTQ int4 of q / k / v / o / gate / up, every row",
                     np.concatenate(ints_two), np.concatenate(ints_one), 5e-5)
  os.remove(dst)
